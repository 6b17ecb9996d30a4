/*
 * pvo_image.c -- ORACLE (test infrastructure): image resampling used by the dlib routines the
 * reference calls.  PARITY UNPINNED (dlib 19.12 not available; restated from the published code).
 *
 *   dlib.get_frontal_face_detector()(rgb, 1)          reference face.py:54,66   -> pyramid_up, pyramid_down<6>
 *   face_recognition_model_v1.compute_face_descriptor reference face.py:74-75   -> extract_image_chips (pyramid_down<2>)
 *   dlib.correlation_tracker.start_track/update       reference tracking.py:203,250-251 -> extract_image_chip, transform_image
 */
#include "pvo.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* source coordinates of a resize stage: i * scale (the restatement), or carried from index to index when PVO_RESIZE_COORDS=accumulate
 * (v = -scale; v += scale per step): the two ways dlib's loop may generate them -- oracle/EXT_REGISTER.md E1.  The HIP path has the
 * same switch (PVF_RESIZE_COORDS, csrc/detect.hip: resize_coords); tests/test_gpu_parity.py runs the pyramid under both. */
static double* resize_coords(int n_in, int n_out)
{
    const double scale = (n_in - 1) / (double)imax(n_out - 1, 1);
    const char* mode = getenv("PVO_RESIZE_COORDS");
    const int accumulate = mode && strcmp(mode, "accumulate") == 0;
    double* v = (double*)malloc(sizeof(double) * (size_t)imax(n_out, 1));
    double a = -scale;
    for (int i = 0; i < n_out; ++i) { a += scale; v[i] = accumulate ? a : i * scale; }
    return v;
}

/* [EXT dlib/image_transforms/interpolation.h resize_image(in,out,interpolate_bilinear)], RGB branch.
 * Coordinates and blend in double; result = (uint8)(v + 0.5). */
void pvo_resize_bilinear_rgb(const uint8_t* in, int ih, int iw, uint8_t* out, int oh, int ow)
{
    double* ys = resize_coords(ih, oh);
    double* xs = resize_coords(iw, ow);
    /* rows are independent: OpenMP over rows for the all-core CPU baseline (pvo_set_threads); same bytes for any thread count */
    #pragma omp parallel for schedule(static) if (oh * ow > 65536)
    for (int r = 0; r < oh; ++r) {
        const double y = ys[r];
        const int top = (int)floor(y);
        const int bottom = imin(top + 1, ih - 1);
        const double tb = y - top;
        for (int c = 0; c < ow; ++c) {
            const double x = xs[c];
            const int left = (int)floor(x);
            const int right = imin(left + 1, iw - 1);
            const double lr = x - left;
            const uint8_t* ptl = in + ((size_t)top * iw + left) * 3;
            const uint8_t* ptr = in + ((size_t)top * iw + right) * 3;
            const uint8_t* pbl = in + ((size_t)bottom * iw + left) * 3;
            const uint8_t* pbr = in + ((size_t)bottom * iw + right) * 3;
            uint8_t* o = out + ((size_t)r * ow + c) * 3;
            for (int k = 0; k < 3; ++k) {
                const double tl = ptl[k], tr = ptr[k], bl = pbl[k], br = pbr[k];
                const double v = (1 - tb) * ((1 - lr) * tl + lr * tr) + tb * ((1 - lr) * bl + lr * br);
                o[k] = (uint8_t)(v + 0.5);
            }
        }
    }
    free(ys); free(xs);
}

/* [EXT OpenCV resize.cpp, INTER_LINEAR, 8-bit, 3 channels] -- cv2.resize(frame, (ow, oh)) as the reference's Video applies it to
 * down-scaled detection frames (video.py:402-403, tracking.py:389-400).  PARITY UNPINNED (OpenCV is not installed here).
 * Pixel-centre mapping, clamped source index, 11-bit coefficients cvRound(f * 2048) (round half to even), horizontal pass in int,
 * vertical pass (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2. */
static void cv_linear_coeffs(int in, int out, int d, int* idx, int* c0, int* c1)
{
    const double scale = (double)in / out;
    float f = (float)((d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { f = 0; s = 0; }
    if (s >= in - 1) { f = 0; s = in - 1; }
    *idx = s;
    *c0 = (int)(short)nearbyintf((1.f - f) * 2048.f);
    *c1 = (int)(short)nearbyintf(f * 2048.f);
}

void pvo_cv_resize_linear_rgb(const uint8_t* in, int ih, int iw, uint8_t* out, int oh, int ow)
{
    for (int y = 0; y < oh; ++y) {
        int sy, b0, b1;
        cv_linear_coeffs(ih, oh, y, &sy, &b0, &b1);
        const int sy1 = imin(sy + 1, ih - 1);
        for (int x = 0; x < ow; ++x) {
            int sx, a0, a1;
            cv_linear_coeffs(iw, ow, x, &sx, &a0, &a1);
            const int sx1 = imin(sx + 1, iw - 1);
            for (int k = 0; k < 3; ++k) {
                const int S0 = in[((size_t)sy * iw + sx) * 3 + k] * a0 + in[((size_t)sy * iw + sx1) * 3 + k] * a1;
                const int S1 = in[((size_t)sy1 * iw + sx) * 3 + k] * a0 + in[((size_t)sy1 * iw + sx1) * 3 + k] * a1;
                out[((size_t)y * ow + x) * 3 + k] = (uint8_t)((((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2);
            }
        }
    }
}

/* [EXT dlib pyramid_up(in,out,pyramid_down<2>)]: out size = rect_up(get_rect(in)) bottom/right + 1 with
 * pyramid_down<2>::point_up(p) = (p + (1.25,0.75))*2 and point rounding floor(v+0.5). */
void pvo_pyramid_up_dims(int ih, int iw, int* oh, int* ow)
{
    const double right = ((iw - 1) + 1.25) * 2.0;
    const double bottom = ((ih - 1) + 0.75) * 2.0;
    *ow = (int)floor(right + 0.5) + 1;
    *oh = (int)floor(bottom + 0.5) + 1;
}

/* [EXT dlib pyramid_down<N>::operator()]: size ((N-1)*nr)/N, then resize_image bilinear. N = 6. */
void pvo_pyramid_down6_dims(int ih, int iw, int* oh, int* ow)
{
    *oh = (5 * ih) / 6;
    *ow = (5 * iw) / 6;
}

/* [EXT dlib pyramid_down<2>::operator()], RGB: separable 1-4-6-4-1 filter in integers, /256 truncating. */
void pvo_pyr_down2_dims(int ih, int iw, int* oh, int* ow)
{
    if (ih <= 8 || iw <= 8) { *oh = 0; *ow = 0; return; }
    *oh = (ih - 3) / 2;
    *ow = (iw - 3) / 2;
}

void pvo_pyr_down2_rgb(const uint8_t* in, int ih, int iw, uint8_t* out)
{
    int oh, ow;
    pvo_pyr_down2_dims(ih, iw, &oh, &ow);
    if (oh == 0 || ow == 0) return;
    static const int k[5] = {1, 4, 6, 4, 1};
    for (int r = 0; r < oh; ++r)
        for (int c = 0; c < ow; ++c)
            for (int ch = 0; ch < 3; ++ch) {
                int acc = 0;
                for (int i = 0; i < 5; ++i) {
                    int row = 0;
                    const uint8_t* p = in + ((size_t)(2 * r + i) * iw + 2 * c) * 3 + ch;
                    for (int j = 0; j < 5; ++j) row += k[j] * p[j * 3];
                    acc += k[i] * row;
                }
                out[((size_t)r * ow + c) * 3 + ch] = (uint8_t)(acc / 256);
            }
}

/* [EXT dlib transform_image(in,out,interpolate_bilinear(),map_point)] with black background;
 * interpolate_bilinear: floor corner, false when the 2x2 footprint leaves the image; assign_pixel truncates. */
static void transform_sub(const uint8_t* img, int stride_w, int x0, int y0, int sw, int sh,
                          const double m[4], const double b[2], uint8_t* out, int oh, int ow)
{
    for (int r = 0; r < oh; ++r)
        for (int c = 0; c < ow; ++c) {
            const double px = m[0] * c + m[1] * r + b[0];
            const double py = m[2] * c + m[3] * r + b[1];
            const double fx = floor(px), fy = floor(py);
            uint8_t* o = out + ((size_t)r * ow + c) * 3;
            if (!(fx >= 0 && fy >= 0 && fx + 1 < sw && fy + 1 < sh)) { o[0] = o[1] = o[2] = 0; continue; }
            const int left = (int)fx, top = (int)fy;
            const double lr = px - left, tb = py - top;
            const uint8_t* ptl = img + ((size_t)(y0 + top) * stride_w + (x0 + left)) * 3;
            const uint8_t* pbl = ptl + (size_t)stride_w * 3;
            for (int k = 0; k < 3; ++k) {
                const double tl = ptl[k], tr = ptl[3 + k], bl = pbl[k], br = pbl[3 + k];
                const double v = (1 - tb) * ((1 - lr) * tl + lr * tr) + tb * ((1 - lr) * bl + lr * br);
                o[k] = (uint8_t)v;
            }
        }
}

void pvo_transform_image_rgb(const uint8_t* img, int h, int w, const double m[4], const double b[2],
                             uint8_t* out, int oh, int ow)
{
    transform_sub(img, w, 0, 0, w, h, m, b, out, oh, ow);
}

/* pyramid_down<2>::point_down on a drectangle [EXT]: p/2 - (1.25,0.75) */
static void rect_down2(double r[4])
{
    r[0] = r[0] / 2.0 - 1.25; r[1] = r[1] / 2.0 - 0.75;
    r[2] = r[2] / 2.0 - 1.25; r[3] = r[3] / 2.0 - 0.75;
}
static double drect_area(const double r[4])
{
    if (r[0] > r[2] || r[1] > r[3]) return 0; /* drectangle::is_empty */
    return (r[2] - r[0]) * (r[3] - r[1]);
}
static void rot(double cx, double cy, double x, double y, double cs, double sn, double* ox, double* oy)
{
    const double dx = x - cx, dy = y - cy;
    *ox = cs * dx - sn * dy + cx;
    *oy = sn * dx + cs * dy + cy;
}

/* [EXT dlib extract_image_chips] for ONE chip (what extract_image_chip forwards to when the chip is
 * scaled or rotated; the unscaled/unrotated integer case degenerates to the same bilinear sampling at
 * integer positions, so no separate fast path is needed for equal results inside the image). */
void pvo_extract_chip_rgb(const uint8_t* img, int h, int w, const pvo_chip_details* d, uint8_t* chip)
{
    const double size = (double)d->rows * d->cols;
    const double R[4] = {d->l, d->t, d->r, d->b};
    /* depth + grow */
    int depth = 0;
    double grow = 2;
    double rect[4] = {R[0], R[1], R[2], R[3]};
    rect_down2(rect);
    while (drect_area(rect) > size) { rect_down2(rect); ++depth; grow = grow * 2 + 2; }
    /* rotated bounding rect */
    const double cx = (R[0] + R[2]) / 2, cy = (R[1] + R[3]) / 2;
    double xs[4], ys[4];
    rot(cx, cy, R[0], R[1], d->cs, d->sn, &xs[0], &ys[0]);
    rot(cx, cy, R[2], R[1], d->cs, d->sn, &xs[1], &ys[1]);
    rot(cx, cy, R[0], R[3], d->cs, d->sn, &xs[2], &ys[2]);
    rot(cx, cy, R[2], R[3], d->cs, d->sn, &xs[3], &ys[3]);
    double bl = xs[0], bt = ys[0], br = xs[0], bb = ys[0];
    for (int i = 1; i < 4; ++i) {
        if (xs[i] < bl) bl = xs[i];
        if (xs[i] > br) br = xs[i];
        if (ys[i] < bt) bt = ys[i];
        if (ys[i] > bb) bb = ys[i];
    }
    bl -= grow; bt -= grow; br += grow; bb += grow;
    /* intersect with image rect (0,0,w-1,h-1) */
    if (bl < 0) bl = 0;
    if (bt < 0) bt = 0;
    if (br > w - 1) br = w - 1;
    if (bb > h - 1) bb = h - 1;
    memset(chip, 0, (size_t)d->rows * d->cols * 3);
    if (bl > br || bt > bb) return;
    const int bx0 = (int)floor(bl + 0.5), by0 = (int)floor(bt + 0.5);
    const int bx1 = (int)floor(br + 0.5), by1 = (int)floor(bb + 0.5);
    const int sw = bx1 - bx0 + 1, sh = by1 - by0 + 1;
    if (sw <= 0 || sh <= 0) return;

    /* level selection in bounding-box coordinates */
    int level = -1;
    double lr_[4] = {R[0] - bx0, R[1] - by0, R[2] - bx0, R[3] - by0};
    for (;;) {
        double nxt[4] = {lr_[0], lr_[1], lr_[2], lr_[3]};
        rect_down2(nxt);
        if (!(drect_area(nxt) > size)) break;
        ++level;
        memcpy(lr_, nxt, sizeof nxt);
    }
    /* pyramid of the sub image, levels 0..level */
    uint8_t* cur = NULL;
    int ch_ = sh, cw_ = sw;
    const uint8_t* src = NULL;
    int src_stride = w, sx0 = bx0, sy0 = by0;
    for (int l = 0; l <= level; ++l) {
        int nh, nw;
        pvo_pyr_down2_dims(ch_, cw_, &nh, &nw);
        uint8_t* nxt = (uint8_t*)calloc((size_t)(nh > 0 ? nh : 1) * (nw > 0 ? nw : 1) * 3, 1);
        if (nh > 0 && nw > 0) {
            if (l == 0) {
                /* copy sub image to contiguous buffer first */
                uint8_t* sub = (uint8_t*)malloc((size_t)sh * sw * 3);
                for (int r = 0; r < sh; ++r)
                    memcpy(sub + (size_t)r * sw * 3, img + ((size_t)(by0 + r) * w + bx0) * 3, (size_t)sw * 3);
                pvo_pyr_down2_rgb(sub, sh, sw, nxt);
                free(sub);
            } else {
                pvo_pyr_down2_rgb(cur, ch_, cw_, nxt);
            }
        }
        free(cur);
        cur = nxt; ch_ = nh; cw_ = nw;
    }
    int th, tw;
    if (level == -1) { src = img; src_stride = w; sx0 = bx0; sy0 = by0; th = sh; tw = sw; }
    else { src = cur; src_stride = cw_; sx0 = 0; sy0 = 0; th = ch_; tw = cw_; }

    /* affine map chip -> level image from three rotated corners of lr_ */
    const double lcx = (lr_[0] + lr_[2]) / 2, lcy = (lr_[1] + lr_[3]) / 2;
    double tlx, tly, trx, try_, blx, bly;
    rot(lcx, lcy, lr_[0], lr_[1], d->cs, d->sn, &tlx, &tly);
    rot(lcx, lcy, lr_[2], lr_[1], d->cs, d->sn, &trx, &try_);
    rot(lcx, lcy, lr_[0], lr_[3], d->cs, d->sn, &blx, &bly);
    double m[4], b[2];
    m[0] = (trx - tlx) / (double)(d->cols - 1);
    m[2] = (try_ - tly) / (double)(d->cols - 1);
    m[1] = (blx - tlx) / (double)(d->rows - 1);
    m[3] = (bly - tly) / (double)(d->rows - 1);
    b[0] = tlx; b[1] = tly;
    if (th > 0 && tw > 0) transform_sub(src, src_stride, sx0, sy0, tw, th, m, b, chip, d->rows, d->cols);
    free(cur);
}
