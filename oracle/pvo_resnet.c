/*
 * pvo_resnet.c -- ORACLE (test infrastructure): dlib face chip alignment + face_recognition_model_v1 forward.
 *   reference: pyannote/video/face/face.py:62,73-76 ; caller scripts/pyannote-face.py:297
 * PARITY UNPINNED ([EXT] restatement of dlib get_face_chip_details / extract_image_chips /
 * dnn_face_recognition_ex anet_type; no model file available, weights are synthetic).
 *
 * Parameter blob walk order (floats):  conv1 w[32][3][7][7] b[32] ; affine g[32] beta[32] ;
 *   14 residual units {conv_a w[N][Cin][3][3] b[N]; affine g,beta ; conv_b w[N][N][3][3] b[N]; affine g,beta} ;
 *   fc w[256][128]  ([in][out], no bias).
 * The embedding is compared with tolerance (L2 <= 1e-4), so summation order here is free: (c, r, s) sequential.
 */
#include "pvo.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { int cin, n, down; } unit_t;
static const unit_t UNITS[14] = {
    {32, 32, 0}, {32, 32, 0}, {32, 32, 0},
    {32, 64, 1}, {64, 64, 0}, {64, 64, 0}, {64, 64, 0},
    {64, 128, 1}, {128, 128, 0}, {128, 128, 0},
    {128, 256, 1}, {256, 256, 0}, {256, 256, 0},
    {256, 256, 1}};

size_t pvo_resnet_param_count(void)
{
    size_t n = 32 * 3 * 7 * 7 + 32 + 64;
    for (int u = 0; u < 14; ++u) {
        n += (size_t)UNITS[u].n * UNITS[u].cin * 9 + UNITS[u].n + 2 * UNITS[u].n;
        n += (size_t)UNITS[u].n * UNITS[u].n * 9 + UNITS[u].n + 2 * UNITS[u].n;
    }
    n += 256 * 128;
    return n;
}

/* dlib con_: cross-correlation, out = 1 + (in + 2*pad - k)/stride ; NCHW planar */
static void conv(const float* in, int c, int h, int w, const float* wt, const float* bias, int n, int k,
                 int stride, int pad, float* out, int oh, int ow)
{
    for (int o = 0; o < n; ++o) {
        float* op = out + (size_t)o * oh * ow;
        for (int i = 0; i < oh * ow; ++i) op[i] = 0.0f;
        for (int ci = 0; ci < c; ++ci)
            for (int r = 0; r < k; ++r)
                for (int s = 0; s < k; ++s) {
                    const float wv = wt[(((size_t)o * c + ci) * k + r) * k + s];
                    for (int y = 0; y < oh; ++y) {
                        const int iy = y * stride - pad + r;
                        if (iy < 0 || iy >= h) continue;
                        const float* ip = in + ((size_t)ci * h + iy) * w;
                        float* orow = op + (size_t)y * ow;
                        for (int x = 0; x < ow; ++x) {
                            const int ix = x * stride - pad + s;
                            if (ix < 0 || ix >= w) continue;
                            orow[x] += ip[ix] * wv;
                        }
                    }
                }
        for (int i = 0; i < oh * ow; ++i) op[i] += bias[o];
    }
}
static void affine_relu(float* x, int n, int hw, const float* g, const float* b, int relu)
{
    for (int o = 0; o < n; ++o)
        for (int i = 0; i < hw; ++i) {
            float v = x[(size_t)o * hw + i] * g[o] + b[o];
            if (relu && v < 0) v = 0;
            x[(size_t)o * hw + i] = v;
        }
}

void pvo_resnet_forward(const uint8_t* chip, const pvo_embed_model* m, float* out128)
{
    const float* p = m->blob;
    const int S = m->chip_size;
    float* x = (float*)malloc(sizeof(float) * 3 * S * S);
    static const float avg[3] = {122.782f, 117.001f, 104.298f};
    for (int c = 0; c < 3; ++c)
        for (int i = 0; i < S * S; ++i) x[(size_t)c * S * S + i] = ((float)chip[(size_t)i * 3 + c] - avg[c]) / 256.0f;
    /* conv1 7x7 s2 p0 -> affine -> relu -> maxpool 3x3 s2 p0 */
    int h = 1 + (S - 7) / 2, w = h, c = 32;
    float* y = (float*)malloc(sizeof(float) * c * h * w);
    conv(x, 3, S, S, p, p + 32 * 3 * 49, 32, 7, 2, 0, y, h, w);
    p += 32 * 3 * 49 + 32;
    affine_relu(y, 32, h * w, p, p + 32, 1);
    p += 64;
    free(x);
    int ph = 1 + (h - 3) / 2, pw = ph;
    x = (float*)malloc(sizeof(float) * c * ph * pw);
    for (int o = 0; o < c; ++o)
        for (int yy = 0; yy < ph; ++yy)
            for (int xx = 0; xx < pw; ++xx) {
                float mx = -INFINITY;
                for (int r = 0; r < 3; ++r)
                    for (int s = 0; s < 3; ++s) {
                        const float v = y[((size_t)o * h + yy * 2 + r) * w + xx * 2 + s];
                        if (v > mx) mx = v;
                    }
                x[((size_t)o * ph + yy) * pw + xx] = mx;
            }
    free(y);
    h = ph; w = pw;
    for (int u = 0; u < 14; ++u) {
        const int cin = UNITS[u].cin, n = UNITS[u].n, down = UNITS[u].down;
        const int stride = down ? 2 : 1, pad = down ? 0 : 1;
        const int ah = 1 + (h + 2 * pad - 3) / stride, aw = 1 + (w + 2 * pad - 3) / stride;
        float* a = (float*)malloc(sizeof(float) * n * ah * aw);
        conv(x, cin, h, w, p, p + (size_t)n * cin * 9, n, 3, stride, pad, a, ah, aw);
        p += (size_t)n * cin * 9 + n;
        affine_relu(a, n, ah * aw, p, p + n, 1);
        p += 2 * n;
        float* b = (float*)malloc(sizeof(float) * n * ah * aw);
        conv(a, n, ah, aw, p, p + (size_t)n * n * 9, n, 3, 1, 1, b, ah, aw);
        p += (size_t)n * n * 9 + n;
        affine_relu(b, n, ah * aw, p, p + n, 0);
        p += 2 * n;
        free(a);
        /* skip path: identity, or avg_pool 2x2 s2 p0; add_prev zero-extends the smaller tensor */
        int sh = h, sw = w;
        float* sk = x;
        if (down) {
            sh = 1 + (h - 2) / 2; sw = 1 + (w - 2) / 2;
            sk = (float*)malloc(sizeof(float) * cin * sh * sw);
            for (int o = 0; o < cin; ++o)
                for (int yy = 0; yy < sh; ++yy)
                    for (int xx = 0; xx < sw; ++xx) {
                        const float* q = x + ((size_t)o * h + yy * 2) * w + xx * 2;
                        sk[((size_t)o * sh + yy) * sw + xx] = (((q[0] + q[1]) + q[w]) + q[w + 1]) * 0.25f;
                    }
        }
        const int oh = ah > sh ? ah : sh, ow = aw > sw ? aw : sw;
        float* o_ = (float*)calloc((size_t)n * oh * ow, sizeof(float));
        for (int o = 0; o < n; ++o)
            for (int yy = 0; yy < oh; ++yy)
                for (int xx = 0; xx < ow; ++xx) {
                    float v = 0;
                    if (yy < ah && xx < aw) v += b[((size_t)o * ah + yy) * aw + xx];
                    if (o < cin && yy < sh && xx < sw) v += sk[((size_t)o * sh + yy) * sw + xx];
                    o_[((size_t)o * oh + yy) * ow + xx] = v > 0 ? v : 0;
                }
        if (down) free(sk);
        free(b); free(x);
        x = o_; h = oh; w = ow; c = n;
    }
    /* avg_pool_everything + fc_no_bias<128> */
    float feat[256];
    for (int o = 0; o < 256; ++o) {
        float s = 0;
        for (int i = 0; i < h * w; ++i) s += x[(size_t)o * h * w + i];
        feat[o] = s / (float)(h * w);
    }
    for (int j = 0; j < 128; ++j) {
        float s = 0;
        for (int o = 0; o < 256; ++o) s += feat[o] * p[(size_t)o * 128 + j];
        out128[j] = s;
    }
    free(x);
}

/* [EXT get_face_chip_details(det, size, padding)] + chip_details(from,to,dims): similarity fit of the mean-face
 * template (landmarks 17..67 minus eyebrows 17..26 and lower lip 55..59, 65..67) to the detected points.
 * Sums over the used points in index order, double. */
void pvo_face_chip_details(const int32_t* pts, const pvo_embed_model* m, pvo_chip_details* out)
{
    const double size = m->chip_size, padding = m->chip_padding;
    double fx[51], fy[51], tx[51], ty[51];
    int n = 0;
    for (int i = 17; i < 68; ++i) {
        if ((55 <= i && i <= 59) || (65 <= i && i <= 67)) continue;
        if (17 <= i && i <= 26) continue;
        fx[n] = ((padding + (double)m->mean_shape_xy[2 * (i - 17)]) / (2 * padding + 1)) * size;
        fy[n] = ((padding + (double)m->mean_shape_xy[2 * (i - 17) + 1]) / (2 * padding + 1)) * size;
        tx[n] = pts[2 * i]; ty[n] = pts[2 * i + 1];
        ++n;
    }
    double mfx = 0, mfy = 0, mtx = 0, mty = 0;
    for (int i = 0; i < n; ++i) { mfx += fx[i]; mfy += fy[i]; mtx += tx[i]; mty += ty[i]; }
    mfx /= n; mfy /= n; mtx /= n; mty /= n;
    double a = 0, b = 0, s = 0;
    for (int i = 0; i < n; ++i) {
        const double ax = fx[i] - mfx, ay = fy[i] - mfy, bx = tx[i] - mtx, by = ty[i] - mty;
        a += ax * bx + ay * by;
        b += ax * by - ay * bx;
        s += ax * ax + ay * ay;
    }
    const double ca = a / s, cb = b / s; /* M = [[ca,-cb],[cb,ca]] ; M*(1,0) = (ca, cb) */
    const double scale = sqrt(ca * ca + cb * cb);
    /* centre of the chip in image space */
    const double hx = size / 2.0, hy = size / 2.0;
    const double cx = (ca * (hx - mfx) - cb * (hy - mfy)) + mtx;
    const double cy = (cb * (hx - mfx) + ca * (hy - mfy)) + mty;
    const double wv = size * scale;
    out->l = cx - wv / 2; out->t = cy - wv / 2; out->r = cx + wv / 2; out->b = cy + wv / 2;
    out->cs = ca / scale; out->sn = cb / scale;
    out->rows = m->chip_size; out->cols = m->chip_size;
}

void pvo_embed(const uint8_t* rgb, int h, int w, const int32_t* pts68, const pvo_embed_model* m, float* out128)
{
    pvo_chip_details d;
    pvo_face_chip_details(pts68, m, &d);
    uint8_t* chip = (uint8_t*)malloc((size_t)d.rows * d.cols * 3);
    pvo_extract_chip_rgb(rgb, h, w, &d, chip);
    pvo_resnet_forward(chip, m, out128);
    free(chip);
}
