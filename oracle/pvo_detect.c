/*
 * pvo_detect.c -- ORACLE (test infrastructure): dlib frontal face detector,
 *   object_detector<scan_fhog_pyramid<pyramid_down<6>>> called as detector(rgb, 1)
 *   reference: pyannote/video/face/face.py:54,66 ; adapter face/tracking.py:36-42 ; caller tracking.py:426
 * PARITY UNPINNED ([EXT] restatement of dlib/image_processing/{scan_fhog_pyramid,object_detector}.h and
 * the python binding's run_detector_with_upscale).
 *
 * Score of filter f at feature position (r,c) of one pyramid level (the HIP kernel keeps this exact order):
 *   acc = 0; for m in rows, n in cols, p in 0..30:  acc = fmaf(F[r-fr/2+m][c-fc/2+n][p], W[f][m][n][p], acc)
 * Candidate order before NMS: score desc, then (filter, level, r, c) asc  (dlib sorts by score only;
 * ties are unspecified there).
 */
#include "pvo.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

static inline long iround(double v) { return (long)floor(v + 0.5); }

/* pyramid_down<6>: point_down(p) = (p-0.3)*5/6+0.3 ; point_up(p) = (p-0.3)*6/5+0.3 ; rectangle corners rounded */
static void rect_down6(long r[4])
{
    const double ratio = (6 - 1.0) / 6;
    for (int i = 0; i < 4; ++i) r[i] = iround((r[i] - 0.3) * ratio + 0.3);
}
static void rect_up6(long r[4])
{
    const double ratio = 6 / (6 - 1.0);
    for (int i = 0; i < 4; ++i) r[i] = iround((r[i] - 0.3) * ratio + 0.3);
}
/* pyramid_down<2>::rect_down : p/2 - (1.25,0.75), rounded */
static void rect_down2i(long r[4])
{
    r[0] = iround(r[0] / 2.0 - 1.25); r[1] = iround(r[1] / 2.0 - 0.75);
    r[2] = iround(r[2] / 2.0 - 1.25); r[3] = iround(r[3] / 2.0 - 0.75);
}

int pvo_detector_levels(int h, int w, const pvo_detector* m)
{
    /* [EXT create_fhog_pyramid]: count rect_down steps while the layer still holds min_layer */
    long r[4] = {0, 0, w - 1, h - 1};
    int levels = 0;
    do {
        rect_down6(r);
        ++levels;
    } while ((r[2] - r[0] + 1) >= m->min_layer_w && (r[3] - r[1] + 1) >= m->min_layer_h && levels < m->max_levels);
    return levels;
}

/* build the image the scanner sees (after `upsample` pyramid_up steps), caller frees */
static uint8_t* upsampled(const uint8_t* rgb, int h, int w, int upsample, int* oh, int* ow)
{
    uint8_t* cur = (uint8_t*)malloc((size_t)h * w * 3);
    memcpy(cur, rgb, (size_t)h * w * 3);
    int ch = h, cw = w;
    for (int u = 0; u < upsample; ++u) {
        int nh, nw;
        pvo_pyramid_up_dims(ch, cw, &nh, &nw);
        uint8_t* nxt = (uint8_t*)malloc((size_t)nh * nw * 3);
        pvo_resize_bilinear_rgb(cur, ch, cw, nxt, nh, nw);
        free(cur);
        cur = nxt; ch = nh; cw = nw;
    }
    *oh = ch; *ow = cw;
    return cur;
}

int pvo_pyramid_level(const uint8_t* rgb, int h, int w, int upsample, int level, uint8_t* out, int* oh, int* ow)
{
    int ch, cw;
    uint8_t* cur = upsampled(rgb, h, w, upsample, &ch, &cw);
    for (int l = 0; l < level; ++l) {
        int nh, nw;
        pvo_pyramid_down6_dims(ch, cw, &nh, &nw);
        uint8_t* nxt = (uint8_t*)malloc((size_t)nh * nw * 3);
        pvo_resize_bilinear_rgb(cur, ch, cw, nxt, nh, nw);
        free(cur);
        cur = nxt; ch = nh; cw = nw;
    }
    *oh = ch; *ow = cw;
    if (out) memcpy(out, cur, (size_t)ch * cw * 3);
    free(cur);
    return 0;
}

void pvo_score_level(const float* feat, int fh, int fw, const pvo_detector* m, int filter, float* out)
{
    const int fr = m->frows, fc = m->fcols;
    const float* W = m->w + (size_t)filter * fr * fc * PVO_FHOG_STRIDE;
    memset(out, 0, (size_t)fh * fw * sizeof(float));
    /* [EXT spatially_filter_image]: only the non-border area is computed; window top-left = (r-fr/2, c-fc/2) */
    const int r0 = fr / 2, c0 = fc / 2;
    const int r1 = fh - (fr - fr / 2 - 1), c1 = fw - (fc - fc / 2 - 1);
    #pragma omp parallel for schedule(static)
    for (int r = r0; r < r1; ++r)
        for (int c = c0; c < c1; ++c) {
            float acc = 0.0f;
            for (int mm = 0; mm < fr; ++mm)
                for (int n = 0; n < fc; ++n) {
                    const float* f = feat + ((size_t)(r - r0 + mm) * fw + (c - c0 + n)) * PVO_FHOG_STRIDE;
                    const float* wv = W + ((size_t)mm * fc + n) * PVO_FHOG_STRIDE;
                    for (int p = 0; p < PVO_FHOG_PLANES; ++p) acc = fmaf(f[p], wv[p], acc);
                }
            out[(size_t)r * fw + c] = acc;
        }
}

static int det_cmp(const void* a, const void* b)
{
    const pvo_det* x = (const pvo_det*)a;
    const pvo_det* y = (const pvo_det*)b;
    if (x->score > y->score) return -1;
    if (x->score < y->score) return 1;
    if (x->filter != y->filter) return x->filter < y->filter ? -1 : 1;
    if (x->level != y->level) return x->level < y->level ? -1 : 1;
    if (x->r != y->r) return x->r < y->r ? -1 : 1;
    if (x->c != y->c) return x->c < y->c ? -1 : 1;
    return 0;
}

/* [EXT fhog_to_image]: feature-space point -> image pixel at that pyramid level */
static void fhog_to_image(long px, long py, int cell, int pad_r, int pad_c, long* ox, long* oy)
{
    long x = (px + 1 - (pad_c - 1) / 2) * cell + 1;
    long y = (py + 1 - (pad_r - 1) / 2) * cell + 1;
    x += (x >= 0) ? cell / 2 : -(cell / 2);
    y += (y >= 0) ? cell / 2 : -(cell / 2);
    *ox = x; *oy = y;
}

int pvo_detect_raw(const uint8_t* rgb, int h, int w, int upsample, const pvo_detector* m, double adjust,
                   pvo_det* out, int cap)
{
    int ch, cw;
    uint8_t* cur = upsampled(rgb, h, w, upsample, &ch, &cw);
    const int levels = pvo_detector_levels(ch, cw, m);
    const int bw = m->fcols - 2 * m->padding, bh = m->frows - 2 * m->padding; /* window in cells w/o padding */
    int n = 0;
    for (int l = 0; l < levels; ++l) {
        if (l > 0) {
            int nh, nw;
            pvo_pyramid_down6_dims(ch, cw, &nh, &nw);
            uint8_t* nxt = (uint8_t*)malloc((size_t)nh * nw * 3);
            pvo_resize_bilinear_rgb(cur, ch, cw, nxt, nh, nw);
            free(cur);
            cur = nxt; ch = nh; cw = nw;
        }
        int fh, fw;
        pvo_fhog_dims(ch, cw, m->cell, m->frows, m->fcols, &fh, &fw);
        if (fh < m->frows || fw < m->fcols) continue;
        float* feat = (float*)malloc((size_t)fh * fw * PVO_FHOG_STRIDE * sizeof(float));
        float* sal = (float*)malloc((size_t)fh * fw * sizeof(float));
        pvo_fhog(cur, ch, cw, m->cell, m->frows, m->fcols, feat);
        const int r0 = m->frows / 2, c0 = m->fcols / 2;
        const int r1 = fh - (m->frows - m->frows / 2 - 1), c1 = fw - (m->fcols - m->fcols / 2 - 1);
        for (int f = 0; f < m->n_filters; ++f) {
            pvo_score_level(feat, fh, fw, m, f, sal);
            const float thresh = (float)((double)m->thresh[f] + adjust);
            for (int r = r0; r < r1; ++r)
                for (int c = c0; c < c1; ++c) {
                    const float s = sal[(size_t)r * fw + c];
                    if (!(s >= thresh)) continue;
                    if (n >= cap) continue;
                    /* centered_rect(point(c,r), bw, bh) -> fhog_to_image corners -> rect_up l times */
                    long rect[4];
                    const long cl = c - bw / 2, ct = r - bh / 2;
                    fhog_to_image(cl, ct, m->cell, m->frows, m->fcols, &rect[0], &rect[1]);
                    fhog_to_image(cl + bw - 1, ct + bh - 1, m->cell, m->frows, m->fcols, &rect[2], &rect[3]);
                    for (int k = 0; k < l; ++k) rect_up6(rect);
                    for (int u = 0; u < upsample; ++u) rect_down2i(rect);
                    pvo_det* d = &out[n++];
                    d->score = s - thresh; /* dlib reports confidence relative to the threshold */
                    d->filter = f; d->level = l; d->r = r; d->c = c;
                    d->l = (int32_t)rect[0]; d->t = (int32_t)rect[1]; d->rr = (int32_t)rect[2]; d->b = (int32_t)rect[3];
                }
        }
        free(feat); free(sal);
    }
    free(cur);
    qsort(out, (size_t)n, sizeof(pvo_det), det_cmp);
    return n;
}

/* [EXT test_box_overlap]: inner = |a & b| ; outer = |bounding box of a and b| */
static int boxes_overlap(const pvo_det* a, const pvo_det* b, double iou, double covered)
{
    const long il = a->l > b->l ? a->l : b->l, it = a->t > b->t ? a->t : b->t;
    const long ir = a->rr < b->rr ? a->rr : b->rr, ib = a->b < b->b ? a->b : b->b;
    if (il > ir || it > ib) return 0;
    const double inner = (double)(ir - il + 1) * (double)(ib - it + 1);
    const long ol = a->l < b->l ? a->l : b->l, ot = a->t < b->t ? a->t : b->t;
    const long orr = a->rr > b->rr ? a->rr : b->rr, ob = a->b > b->b ? a->b : b->b;
    const double outer = (double)(orr - ol + 1) * (double)(ob - ot + 1);
    const double aa = (double)(a->rr - a->l + 1) * (double)(a->b - a->t + 1);
    const double ab = (double)(b->rr - b->l + 1) * (double)(b->b - b->t + 1);
    return (inner / outer > iou || inner / aa > covered || inner / ab > covered);
}

int pvo_nms(const pvo_det* cands, int n, double iou, double covered, pvo_det* out, int cap)
{
    int k = 0;
    for (int i = 0; i < n; ++i) {
        int hit = 0;
        for (int j = 0; j < k && !hit; ++j) hit = boxes_overlap(&out[j], &cands[i], iou, covered);
        if (hit) continue;
        if (k < cap) out[k++] = cands[i];
    }
    return k;
}

int pvo_detect(const uint8_t* rgb, int h, int w, int upsample, const pvo_detector* m, double adjust,
               pvo_det* out, int cap)
{
    const int rawcap = 1 << 16;
    pvo_det* raw = (pvo_det*)malloc(sizeof(pvo_det) * rawcap);
    const int n = pvo_detect_raw(rgb, h, w, upsample, m, adjust, raw, rawcap);
    const int k = pvo_nms(raw, n, m->nms_iou, m->nms_covered, out, cap);
    free(raw);
    return k;
}

void pvo_set_threads(int n) { omp_set_num_threads(n > 0 ? n : 1); }
int pvo_get_max_threads(void) { return omp_get_max_threads(); }
