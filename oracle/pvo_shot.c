/* ORACLE (test infrastructure only) -- displaced frame difference of the reference's shot boundary detector.
 *
 * Reference: pyannote/video/structure/shot.py
 *   :71-73  _convert: gray = cv2.cvtColor(rgb, cv2.COLOR_RGB2GRAY); cv2.resize(gray, self._resize)
 *           with self._resize = (height, int(w * height / h)) (:62) handed to cv2.resize as dsize = (WIDTH, HEIGHT): the small image is
 *           `height` pixels WIDE and int(w * height / h) pixels HIGH (50 x 88 for 1080p) -- kept as the reference does it.
 *   :75-99  dfd: flow = cv2.calcOpticalFlowFarneback(previous, current, None, 0.5, 3, 15, 3, 5, 1.1, 0); every pixel (x, y) of
 *           `previous` is compared with current[int(clamp(y + dy)), int(clamp(x + dx))] where `dy, dx = flow[y, x]` -- i.e. the flow's
 *           x component is added to y and its y component to x, as the reference writes it; mean absolute difference.
 *
 * PARITY UNPINNED: cv2 is not installed here and OpenCV's sources are not in the container; cvtColor, the 8-bit linear resize and
 * Farneback's algorithm (optflowgf.cpp) are restated from their published descriptions [EXT]:
 *   RGB2GRAY      (R * 4899 + G * 9617 + B * 1868 + 8192) >> 14
 *   resize        pvo_image.c's INTER_LINEAR restatement, one channel
 *   Farneback     number of pyramid levels: scale *= 0.5 while both sides * scale >= 32; an image `height` < 64 pixels wide has NO coarser
 *                 level, so the flow comes from the full-size level alone (this file implements that case: levels == 0):
 *                 float image, GaussianBlur 3 x 3 with sigma <= 0 (fixed kernel 1/4 1/2 1/4, BORDER_REFLECT_101), polynomial expansion
 *                 (poly_n 5, poly_sigma 1.1, rows / columns clamped at the border), matrices from zero flow, then 3 x (15 x 15 box sums of
 *                 the matrices with clamped borders -> 2 x 2 solve -> matrices from the new flow).
 * Every float operation is written out in one order (no contraction); csrc/shot.hip repeats that order, so the two agree bit for bit.
 */
#include "pvo.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static int imin_(int a, int b) { return a < b ? a : b; }
static int imax_(int a, int b) { return a > b ? a : b; }

static void cv_coeffs1(int in, int out, int d, int* idx, int* c0, int* c1)
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

static inline int gray_of(const uint8_t* p) { return (p[0] * 4899 + p[1] * 9617 + p[2] * 1868 + 8192) >> 14; }

/* shot.py:71-73 on one RGB frame: out[oh][ow] bytes, ow = `height`, oh = int(w * height / h) */
void pvo_shot_convert(const uint8_t* rgb, int ih, int iw, uint8_t* out, int oh, int ow)
{
    for (int y = 0; y < oh; ++y) {
        int sy, b0, b1;
        cv_coeffs1(ih, oh, y, &sy, &b0, &b1);
        const int sy1 = imin_(sy + 1, ih - 1);
        for (int x = 0; x < ow; ++x) {
            int sx, a0, a1;
            cv_coeffs1(iw, ow, x, &sx, &a0, &a1);
            const int sx1 = imin_(sx + 1, iw - 1);
            const int S0 = gray_of(rgb + ((size_t)sy * iw + sx) * 3) * a0 + gray_of(rgb + ((size_t)sy * iw + sx1) * 3) * a1;
            const int S1 = gray_of(rgb + ((size_t)sy1 * iw + sx) * 3) * a0 + gray_of(rgb + ((size_t)sy1 * iw + sx1) * 3) * a1;
            out[(size_t)y * ow + x] = (uint8_t)((((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2);
        }
    }
}

/* tables shared with the device: g[0..5], xg[0..5], xxg[0..5], then ig11, ig03, ig33, ig55 (22 floats).
 * FarnebackPrepareGaussian(n = 5, sigma = 1.1): normalised Gaussian, its first two moments, and four entries of the inverse of the
 * 6 x 6 moment matrix G (by symmetry only G00, G11, G33, G55 are independent; the inverse is taken in double, Gauss-Jordan). */
void pvo_shot_tables(float* t)
{
    const int n = 5;
    const double sigma = 1.1;
    double g[11], s = 0;
    for (int x = -n; x <= n; ++x) { g[x + n] = exp(-x * x / (2 * sigma * sigma)); s += g[x + n]; }
    float gf[11];
    for (int x = -n; x <= n; ++x) gf[x + n] = (float)(g[x + n] * (1.0 / s));
    for (int k = 0; k <= n; ++k) { t[k] = gf[k + n]; t[6 + k] = (float)(k * gf[k + n]); t[12 + k] = (float)(k * k * gf[k + n]); }
    double G[6][6];
    memset(G, 0, sizeof G);
    for (int y = -n; y <= n; ++y)
        for (int x = -n; x <= n; ++x) {
            const double w = (double)gf[y + n] * (double)gf[x + n];
            G[0][0] += w; G[1][1] += w * x * x; G[3][3] += w * x * x * x * x; G[5][5] += w * x * x * y * y;
        }
    G[2][2] = G[0][3] = G[0][4] = G[3][0] = G[4][0] = G[1][1];
    G[4][4] = G[3][3];
    G[3][4] = G[4][3] = G[5][5];
    double A[6][12];
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 12; ++j) A[i][j] = j < 6 ? G[i][j] : (j - 6 == i ? 1.0 : 0.0);
    for (int c = 0; c < 6; ++c) {
        int p = c;
        for (int r = c + 1; r < 6; ++r) if (fabs(A[r][c]) > fabs(A[p][c])) p = r;
        for (int j = 0; j < 12; ++j) { const double tmp = A[c][j]; A[c][j] = A[p][j]; A[p][j] = tmp; }
        const double d = A[c][c];
        for (int j = 0; j < 12; ++j) A[c][j] /= d;
        for (int r = 0; r < 6; ++r) {
            if (r == c) continue;
            const double f = A[r][c];
            for (int j = 0; j < 12; ++j) A[r][j] -= f * A[c][j];
        }
    }
    t[18] = (float)A[1][7]; t[19] = (float)A[0][9]; t[20] = (float)A[3][9]; t[21] = (float)A[5][11];
}

static inline int reflect101(int i, int n) { if (i < 0) i = -i; if (i >= n) i = 2 * n - 2 - i; return i; }

/* float image, 3 x 3 blur (1/4 1/2 1/4 in both directions, rows first then columns, BORDER_REFLECT_101) */
static void blur3(const uint8_t* src, int h, int w, float* dst, float* tmp)
{
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const float a = (float)src[(size_t)y * w + reflect101(x - 1, w)], b = (float)src[(size_t)y * w + x],
                        c = (float)src[(size_t)y * w + reflect101(x + 1, w)];
            tmp[(size_t)y * w + x] = (a * 0.25f + b * 0.5f) + c * 0.25f;
        }
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const float a = tmp[(size_t)reflect101(y - 1, h) * w + x], b = tmp[(size_t)y * w + x], c = tmp[(size_t)reflect101(y + 1, h) * w + x];
            dst[(size_t)y * w + x] = (a * 0.25f + b * 0.5f) + c * 0.25f;
        }
}

/* polynomial expansion: R[y][x][5]; vertical sums with clamped rows, horizontal sums with clamped columns */
static void polyexp(const float* src, int h, int w, const float* t, float* R, float* row)
{
    const int n = 5;
    const float *g = t, *xg = t + 6, *xxg = t + 12;
    const float ig11 = t[18], ig03 = t[19], ig33 = t[20], ig55 = t[21];
    for (int y = 0; y < h; ++y) {
        for (int x = 0; x < w; ++x) {
            float s0 = src[(size_t)y * w + x] * g[0], s1 = 0.f, s2 = 0.f;
            for (int k = 1; k <= n; ++k) {
                const float p = src[(size_t)imin_(y + k, h - 1) * w + x], m = src[(size_t)imax_(y - k, 0) * w + x];
                s0 = s0 + g[k] * (p + m);
                s1 = s1 + xg[k] * (p - m);
                s2 = s2 + xxg[k] * (p + m);
            }
            row[x * 3] = s0; row[x * 3 + 1] = s1; row[x * 3 + 2] = s2;
        }
        for (int x = 0; x < w; ++x) {
            float b1 = row[x * 3] * g[0], b2 = 0.f, b3 = row[x * 3 + 1] * g[0], b4 = 0.f, b5 = row[x * 3 + 2] * g[0], b6 = 0.f;
            for (int k = 1; k <= n; ++k) {
                const float* rp = row + imin_(x + k, w - 1) * 3;
                const float* rm = row + imax_(x - k, 0) * 3;
                const float tg = rp[0] + rm[0];
                b1 = b1 + tg * g[k];
                b4 = b4 + tg * xxg[k];
                b2 = b2 + (rp[0] - rm[0]) * xg[k];
                b3 = b3 + (rp[1] + rm[1]) * g[k];
                b6 = b6 + (rp[1] - rm[1]) * xg[k];
                b5 = b5 + (rp[2] + rm[2]) * g[k];
            }
            float* d = R + ((size_t)y * w + x) * 5;
            d[1] = b2 * ig11;
            d[0] = b3 * ig11;
            d[3] = b1 * ig03 + b4 * ig33;
            d[2] = b1 * ig03 + b5 * ig33;
            d[4] = b6 * ig55;
        }
    }
}

/* FarnebackUpdateMatrices: M[y][x][5] from the two expansions and the current flow */
static void update_matrices(const float* R0, const float* R1, const float* flow, int h, int w, float* M)
{
    static const float border[5] = {0.14f, 0.14f, 0.4472f, 0.4472f, 0.4472f};
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const float dx = flow[((size_t)y * w + x) * 2], dy = flow[((size_t)y * w + x) * 2 + 1];
            float fx = (float)x + dx, fy = (float)y + dy;
            const int x1 = (int)floorf(fx), y1 = (int)floorf(fy);
            const float* r0 = R0 + ((size_t)y * w + x) * 5;
            float r2, r3, r4, r5, r6;
            fx -= (float)x1; fy -= (float)y1;
            if ((unsigned)x1 < (unsigned)(w - 1) && (unsigned)y1 < (unsigned)(h - 1)) {
                const float a00 = (1.f - fx) * (1.f - fy), a01 = fx * (1.f - fy), a10 = (1.f - fx) * fy, a11 = fx * fy;
                const float* p = R1 + ((size_t)y1 * w + x1) * 5;
                const float* q = p + (size_t)w * 5;
                r2 = ((a00 * p[0] + a01 * p[5]) + a10 * q[0]) + a11 * q[5];
                r3 = ((a00 * p[1] + a01 * p[6]) + a10 * q[1]) + a11 * q[6];
                r4 = ((a00 * p[2] + a01 * p[7]) + a10 * q[2]) + a11 * q[7];
                r5 = ((a00 * p[3] + a01 * p[8]) + a10 * q[3]) + a11 * q[8];
                r6 = ((a00 * p[4] + a01 * p[9]) + a10 * q[4]) + a11 * q[9];
                r4 = (r0[2] + r4) * 0.5f;
                r5 = (r0[3] + r5) * 0.5f;
                r6 = (r0[4] + r6) * 0.25f;
            } else {
                r2 = r3 = 0.f;
                r4 = r0[2]; r5 = r0[3]; r6 = r0[4] * 0.5f;
            }
            r2 = (r0[0] - r2) * 0.5f;
            r3 = (r0[1] - r3) * 0.5f;
            r2 = r2 + (r4 * dy + r6 * dx);
            r3 = r3 + (r6 * dy + r5 * dx);
            if ((unsigned)(x - 5) >= (unsigned)(w - 10) || (unsigned)(y - 5) >= (unsigned)(h - 10)) {
                const float scale = (x < 5 ? border[x] : 1.f) * (x >= w - 5 ? border[w - x - 1] : 1.f) * (y < 5 ? border[y] : 1.f) *
                                    (y >= h - 5 ? border[h - y - 1] : 1.f);
                r2 *= scale; r3 *= scale; r4 *= scale; r5 *= scale; r6 *= scale;
            }
            float* m = M + ((size_t)y * w + x) * 5;
            m[0] = r4 * r4 + r6 * r6;
            m[1] = (r4 + r5) * r6;
            m[2] = r5 * r5 + r6 * r6;
            m[3] = r4 * r2 + r6 * r3;
            m[4] = r6 * r2 + r5 * r3;
        }
}

/* FarnebackUpdateFlow_Blur: 15 x 15 box sums of M (clamped rows / columns; columns summed first per row, then rows top to bottom), solve */
static void update_flow(const float* M, int h, int w, float* flow, float* colsum)
{
    const int m = 7;
    const float scale = 1.f / (float)(15 * 15);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            float* c = colsum + ((size_t)y * w + x) * 5;
            for (int k = 0; k < 5; ++k) {
                float s = 0.f;
                for (int dxx = -m; dxx <= m; ++dxx) s = s + M[((size_t)y * w + imin_(imax_(x + dxx, 0), w - 1)) * 5 + k];
                c[k] = s;
            }
        }
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            float v[5];
            for (int k = 0; k < 5; ++k) {
                float s = 0.f;
                for (int dyy = -m; dyy <= m; ++dyy) s = s + colsum[((size_t)imin_(imax_(y + dyy, 0), h - 1) * w + x) * 5 + k];
                v[k] = s * scale;
            }
            const float idet = 1.f / ((v[0] * v[2] - v[1] * v[1]) + 1e-3f);
            flow[((size_t)y * w + x) * 2] = (v[0] * v[4] - v[1] * v[3]) * idet;
            flow[((size_t)y * w + x) * 2 + 1] = (v[2] * v[3] - v[1] * v[4]) * idet;
        }
}

/* ---- the coarser pyramid levels (an image side of 64 pixels or more: Shot(height >= 64), scripts/pyannote-structure.py:45,111) -------------
 * [EXT optflowgf.cpp, FarnebackOpticalFlowImpl::calc], restated, PARITY UNPINNED like everything above:
 *   levels: scale = 1; for k in 0..2: scale *= 0.5; stop when width * scale < 32 or height * scale < 32   => `levels` coarser levels
 *   for k = levels .. 0 (coarse to fine): scale = 0.5^k, sigma = (1 / scale - 1) / 2, smooth_sz = max(cvRound(5 sigma) | 1, 3),
 *       level size = (cvRound(width * scale), cvRound(height * scale));
 *       flow = 0 on the coarsest level, otherwise resize(previous level's flow, INTER_LINEAR) * 2;
 *       each image: float, GaussianBlur(smooth_sz x smooth_sz, sigma) at FULL size, resize(INTER_LINEAR) to the level size, polynomial expansion;
 *       matrices from the flow, then 3 x (box sums -> solve -> matrices).
 *   GaussianBlur, sigma > 0 [getGaussianKernel(n, sigma, CV_32F)]: t_i = exp(-x_i^2 / (2 sigma^2)) in double, stored as float, normalised by the
 *       double sum of the floats, stored as float; rows first (taps left to right), then columns (centre tap, then the symmetric pairs);
 *       BORDER_REFLECT_101.  Level 0 (sigma = 0) keeps the fixed 1/4 1/2 1/4 kernel and the arithmetic of blur3 above.
 *   resize of float images [resize, INTER_LINEAR, CV_32F]: fx = (float)((dx + 0.5) * (src / dst) - 0.5), sx = floor(fx), fx -= sx, clamped at
 *       both ends with fx = 0; a row pair is blended horizontally first (S[sx] * (1 - fx) + S[sx + 1] * fx), then vertically. */
static int cv_round(double v) { return (int)nearbyint(v); }            /* cvRound: to nearest, ties to even */

static void gauss_kernel(int n, double sigma, float* k)
{
    const double s2 = -0.5 / (sigma * sigma);
    double sum = 0;
    for (int i = 0; i < n; ++i) {
        const double x = i - (n - 1) * 0.5;
        k[i] = (float)exp(s2 * x * x);
        sum += k[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < n; ++i) k[i] = (float)(k[i] * sum);
}

static void blur_n(const float* src, int h, int w, const float* k, int n, float* dst, float* tmp)
{
    const int r = n / 2;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            float s = k[0] * src[(size_t)y * w + reflect101(x - r, w)];
            for (int j = 1; j < n; ++j) s = s + k[j] * src[(size_t)y * w + reflect101(x - r + j, w)];
            tmp[(size_t)y * w + x] = s;
        }
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            float s = k[r] * tmp[(size_t)y * w + x];
            for (int d = 1; d <= r; ++d) s = s + k[r + d] * (tmp[(size_t)reflect101(y + d, h) * w + x] + tmp[(size_t)reflect101(y - d, h) * w + x]);
            dst[(size_t)y * w + x] = s;
        }
}

static void resize_coeff_f(int in, int out, int d, int* idx, float* a0, float* a1)
{
    const double scale = (double)in / out;
    float f = (float)((d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { f = 0; s = 0; }
    if (s >= in - 1) { f = 0; s = in - 1; }
    *idx = s; *a0 = 1.f - f; *a1 = f;
}

/* src [ih][iw][cn] -> dst [oh][ow][cn] */
static void resize_linear_f(const float* src, int ih, int iw, int cn, float* dst, int oh, int ow)
{
    for (int y = 0; y < oh; ++y) {
        int sy; float b0, b1;
        resize_coeff_f(ih, oh, y, &sy, &b0, &b1);
        const int sy1 = imin_(sy + 1, ih - 1);
        for (int x = 0; x < ow; ++x) {
            int sx; float a0, a1;
            resize_coeff_f(iw, ow, x, &sx, &a0, &a1);
            const int sx1 = imin_(sx + 1, iw - 1);
            for (int c = 0; c < cn; ++c) {
                const float r0 = src[((size_t)sy * iw + sx) * cn + c] * a0 + src[((size_t)sy * iw + sx1) * cn + c] * a1;
                const float r1 = src[((size_t)sy1 * iw + sx) * cn + c] * a0 + src[((size_t)sy1 * iw + sx1) * cn + c] * a1;
                dst[((size_t)y * ow + x) * cn + c] = r0 * b0 + r1 * b1;
            }
        }
    }
}

int pvo_farneback_levels(int h, int w)
{
    int k = 0;
    double scale = 1;
    for (; k < 3; ++k) {
        scale *= 0.5;
        if (w * scale < 32 || h * scale < 32) break;
    }
    return k;
}

/* cv2.calcOpticalFlowFarneback(prev, cur, None, 0.5, 3, 15, 3, 5, 1.1, 0), any size: flow [h][w][2] */
int pvo_farneback(const uint8_t* prev, const uint8_t* cur, int h, int w, const float* tables, float* flow)
{
    const int levels = pvo_farneback_levels(h, w);
    const size_t px = (size_t)h * w;
    float* buf = (float*)malloc(sizeof(float) * (px * 4 + px * 5 * 4 + px * 4 + (size_t)w * 3));
    if (!buf) return -2;
    float *F = buf, *tmp = F + px, *I0 = tmp + px, *I1 = I0 + px, *R0 = I1 + px, *R1 = R0 + px * 5, *M = R1 + px * 5, *cs = M + px * 5,
          *fa = cs + px * 5, *fb = fa + px * 2, *row = fb + px * 2;
    const uint8_t* img[2] = {prev, cur};
    float* I[2] = {I0, I1};
    float* R[2] = {R0, R1};
    float *fl = fa, *fl_prev = fb;
    int plh = 0, plw = 0;
    for (int k = levels; k >= 0; --k) {
        double scale = 1;
        for (int i = 0; i < k; ++i) scale *= 0.5;
        const double sigma = (1. / scale - 1) * 0.5;
        int smooth_sz = cv_round(sigma * 5) | 1;
        if (smooth_sz < 3) smooth_sz = 3;
        const int lw = cv_round(w * scale), lh = cv_round(h * scale);
        if (k == levels) memset(fl, 0, sizeof(float) * (size_t)lh * lw * 2);
        else {
            resize_linear_f(fl_prev, plh, plw, 2, fl, lh, lw);
            for (size_t i = 0; i < (size_t)lh * lw * 2; ++i) fl[i] = fl[i] * 2.f;
        }
        for (int s = 0; s < 2; ++s) {
            if (k == 0) blur3(img[s], h, w, I[s], tmp);
            else {
                float kern[32];
                gauss_kernel(smooth_sz, sigma, kern);
                float* B = M;                                   /* (free until the matrices are formed) */
                for (size_t i = 0; i < px; ++i) F[i] = (float)img[s][i];
                blur_n(F, h, w, kern, smooth_sz, B, tmp);
                resize_linear_f(B, h, w, 1, I[s], lh, lw);
            }
            polyexp(I[s], lh, lw, tables, R[s], row);
        }
        update_matrices(R0, R1, fl, lh, lw, M);
        for (int it = 0; it < 3; ++it) {
            update_flow(M, lh, lw, fl, cs);
            if (it < 2) update_matrices(R0, R1, fl, lh, lw, M);
        }
        float* t2 = fl; fl = fl_prev; fl_prev = t2;
        plh = lh; plw = lw;
    }
    memcpy(flow, fl_prev, sizeof(float) * px * 2);
    free(buf);
    return 0;
}

/* the one-level case under its old name (an image side below 64 pixels) */
int pvo_farneback_small(const uint8_t* prev, const uint8_t* cur, int h, int w, const float* tables, float* flow)
{
    if (pvo_farneback_levels(h, w) != 0) return -1;
    return pvo_farneback(prev, cur, h, w, tables, flow);
}

/* shot.py:89-99: reconstruct[y, x] = current[ry, rx] with `dy, dx = flow[y, x]`; mean |previous - reconstruct| */
double pvo_shot_dfd_from_flow(const uint8_t* prev, const uint8_t* cur, int h, int w, const float* flow)
{
    long sum = 0;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            /* `x + dx` with a Python int and a numpy.float32 stays float32 under NumPy >= 2 (NEP 50) -- the NumPy this container runs the
             * reference with, i.e. what the verbatim run pins; NumPy 1.x promoted the sum to float64 (a one-pixel difference in roughly one
             * lookup per 1e5).  min / max / int() are exact. */
            const float dy = flow[((size_t)y * w + x) * 2], dx = flow[((size_t)y * w + x) * 2 + 1];
            float fx = (float)x + dx, fy = (float)y + dy;
            if (fx > (float)(w - 1)) fx = (float)(w - 1);
            if (fx < 0) fx = 0;
            if (fy > (float)(h - 1)) fy = (float)(h - 1);
            if (fy < 0) fy = 0;
            const int rx = (int)fx, ry = (int)fy;
            const int d = (int)prev[(size_t)y * w + x] - (int)cur[(size_t)ry * w + rx];
            sum += d < 0 ? -d : d;
        }
    return (double)sum / (double)((size_t)h * w);
}

double pvo_shot_dfd(const uint8_t* prev, const uint8_t* cur, int h, int w, const float* tables)
{
    float* flow = (float*)malloc(sizeof(float) * (size_t)h * w * 2);
    if (!flow) return -1.0;
    double r = -1.0;
    if (pvo_farneback(prev, cur, h, w, tables, flow) == 0) r = pvo_shot_dfd_from_flow(prev, cur, h, w, flow);
    free(flow);
    return r;
}
