/*
 * pvo_dsst.c -- ORACLE (test infrastructure): dlib correlation_tracker (DSST, Danelljan et al. 2014),
 * default constructor parameters.
 *   reference: pyannote/video/tracking.py:250-251 (start_track), :203 (update -> PSR), :231,165 (get_position)
 * PARITY UNPINNED ([EXT] restatement of dlib/image_processing/correlation_tracker.h).
 *
 * Deterministic forms shared with the HIP kernels:
 *   - FFT: radix-2 DIT, bit-reversal first, stages s=1..log2N, twiddles from the host table, complex product
 *     (ac-bd, ad+bc) without fma; 2-D = all rows then all columns; inverse = conjugate twiddles, scale 1/N last.
 *   - exp() -> pvo_det_exp (range reduction + degree-13 Horner polynomial, only + and *).
 *   - sums over the 32 feature planes sequential in plane order; PSR statistics: per-row sequential sums
 *     (columns ascending), then rows ascending.
 */
#include "pvo.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define FS 64      /* filter size 1<<6 */
#define NPL 32     /* 31 fhog planes + gray/255 */
#define NSC 32     /* scale levels 1<<5 */
#define SWIN 23    /* scale window size */
#define SDIM (4 * 4 * 32)

static const double REG_SPACE = 0.001, NU_SPACE = 0.025, REG_SCALE = 0.001, NU_SCALE = 0.025, ALPHA = 1.020;

struct pvo_tracker {
    pvo_dsst_tables tb;
    double* F;    /* [NPL][FS][FS][2] scratch: features of the last chip, FFT'd */
    double* A;    /* [NPL][FS][FS][2] */
    double* B;    /* [FS][FS] */
    double* Fs;   /* [SDIM][NSC][2] */
    double* As;   /* [SDIM][NSC][2] */
    double Bs[NSC];
    double pos[4];
};

double pvo_det_exp(double x)
{
    /* exp(x) = 2^k * exp(r), r = x - k ln2 in [-ln2/2, ln2/2] */
    const double inv_ln2 = 1.4426950408889634074, ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    if (x < -700.0) return 0.0;
    const double kf = floor(x * inv_ln2 + 0.5);
    const double r = (x - kf * ln2_hi) - kf * ln2_lo;
    static const double c[14] = {1.0, 1.0, 1.0 / 2, 1.0 / 6, 1.0 / 24, 1.0 / 120, 1.0 / 720, 1.0 / 5040, 1.0 / 40320,
                                 1.0 / 362880, 1.0 / 3628800, 1.0 / 39916800, 1.0 / 479001600, 1.0 / 6227020800.0};
    double p = c[13];
    for (int i = 12; i >= 0; --i) p = p * r + c[i];
    const int64_t k = (int64_t)kf;
    union { uint64_t u; double d; } s;
    s.u = (uint64_t)(k + 1023) << 52;
    return p * s.d;
}

static void fft1d(double* x, int stride, int n, int logn, const double* tw, int inverse)
{
    /* bit reversal */
    for (int i = 0; i < n; ++i) {
        int j = 0;
        for (int b = 0; b < logn; ++b) j |= ((i >> b) & 1) << (logn - 1 - b);
        if (j > i) {
            double tr = x[2 * i * stride], ti = x[2 * i * stride + 1];
            x[2 * i * stride] = x[2 * j * stride]; x[2 * i * stride + 1] = x[2 * j * stride + 1];
            x[2 * j * stride] = tr; x[2 * j * stride + 1] = ti;
        }
    }
    for (int s = 1; s <= logn; ++s) {
        const int m = 1 << s, half = m >> 1, tstep = n / m;
        for (int k = 0; k < n; k += m)
            for (int j = 0; j < half; ++j) {
                const double wr = tw[2 * j * tstep], wi = inverse ? tw[2 * j * tstep + 1] : -tw[2 * j * tstep + 1];
                double* a = x + 2 * (size_t)(k + j) * stride;
                double* b = x + 2 * (size_t)(k + j + half) * stride;
                const double tr = wr * b[0] - wi * b[1];
                const double ti = wr * b[1] + wi * b[0];
                const double ur = a[0], ui = a[1];
                a[0] = ur + tr; a[1] = ui + ti;
                b[0] = ur - tr; b[1] = ui - ti;
            }
    }
    if (inverse) {
        const double sc = 1.0 / n;
        for (int i = 0; i < n; ++i) { x[2 * i * stride] *= sc; x[2 * i * stride + 1] *= sc; }
    }
}

void pvo_fft64x64(double* d, const double* tw64, int inverse)
{
    for (int r = 0; r < FS; ++r) fft1d(d + (size_t)r * FS * 2, 1, FS, 6, tw64, inverse);
    for (int c = 0; c < FS; ++c) fft1d(d + (size_t)c * 2, FS, FS, 6, tw64, inverse);
}

static void scale_rect(double r[4], double s)
{
    const double cx = (r[0] + r[2]) / 2, cy = (r[1] + r[3]) / 2;
    const double w = (r[2] - r[0]) * s, h = (r[3] - r[1]) * s;
    r[0] = cx - w / 2; r[1] = cy - h / 2; r[2] = cx + w / 2; r[3] = cy + h / 2;
}

/* make_chip: 64x64 chip of (p * 1.4) -> FHOG(cell 1, pad 3x3) + gray/255, times the cosine mask, into F (real parts).
 * chip (x,y) -> image (L + x*(R-L)/63, T + y*(Bm-T)/63) is returned through map[4] = {L, T, sx, sy}. */
static void make_chip(pvo_tracker* tk, const uint8_t* rgb, int h, int w, const double p[4], double map[4])
{
    pvo_chip_details d;
    double r[4] = {p[0], p[1], p[2], p[3]};
    scale_rect(r, 1.4);
    d.l = r[0]; d.t = r[1]; d.r = r[2]; d.b = r[3]; d.cs = 1.0; d.sn = 0.0; d.rows = FS; d.cols = FS;
    uint8_t* chip = (uint8_t*)malloc(FS * FS * 3);
    pvo_extract_chip_rgb(rgb, h, w, &d, chip);
    float* hog = (float*)malloc(sizeof(float) * FS * FS * PVO_FHOG_STRIDE);
    pvo_fhog(chip, FS, FS, 1, 3, 3, hog);
    for (int i = 0; i < NPL; ++i)
        for (int y = 0; y < FS; ++y)
            for (int x = 0; x < FS; ++x) {
                float v;
                if (i < 31) v = hog[((size_t)y * FS + x) * PVO_FHOG_STRIDE + i];
                else {
                    const uint8_t* q = chip + ((size_t)y * FS + x) * 3;
                    v = (float)(((unsigned)q[0] + q[1] + q[2]) / 3) / 255.0f;
                }
                double* f = tk->F + (((size_t)i * FS + y) * FS + x) * 2;
                f[0] = (double)v * tk->tb.mask64[y * FS + x];
                f[1] = 0.0;
            }
    free(hog); free(chip);
    map[0] = r[0]; map[1] = r[1];
    map[2] = (r[2] - r[0]) / (double)(FS - 1);
    map[3] = (r[3] - r[1]) / (double)(FS - 1);
    for (int i = 0; i < NPL; ++i) pvo_fft64x64(tk->F + (size_t)i * FS * FS * 2, tk->tb.tw64, 0);
}

static void target_image(const pvo_tracker* tk, double px, double py, double* g /* [FS][FS][2] */)
{
    memset(g, 0, sizeof(double) * FS * FS * 2);
    const long cx = (long)floor(px + 0.5), cy = (long)floor(py + 0.5);
    long x0 = cx - 10, x1 = cx + 10, y0 = cy - 10, y1 = cy + 10;
    if (x0 < 0) x0 = 0;
    if (y0 < 0) y0 = 0;
    if (x1 > FS - 1) x1 = FS - 1;
    if (y1 > FS - 1) y1 = FS - 1;
    for (long r = y0; r <= y1; ++r)
        for (long c = x0; c <= x1; ++c) {
            const double dx = (double)c - px, dy = (double)r - py;
            const double dist = sqrt(dx * dx + dy * dy);
            g[((size_t)r * FS + c) * 2] = pvo_det_exp(-dist / 3.0);
        }
    pvo_fft64x64(g, tk->tb.tw64, 0);
    for (int i = 0; i < FS * FS; ++i) g[2 * i + 1] = -g[2 * i + 1];
}

/* make_scale_space: 32 chips 23x23 around `position` scaled alpha^(k-16); FHOG cell 4 -> 4x4x31 (+ gray/255 top-left 4x4);
 * Fs[(r*4+c)*32 + plane][k] = value * mask_scale[k]; then 1-D FFT over k. */
static void make_scale_space(pvo_tracker* tk, const uint8_t* rgb, int h, int w)
{
    double ppp[4] = {tk->pos[0], tk->pos[1], tk->pos[2], tk->pos[3]};
    scale_rect(ppp, tk->tb.alpha_pow_m16);
    uint8_t chip[SWIN * SWIN * 3];
    float hog[4 * 4 * PVO_FHOG_STRIDE];
    for (int k = 0; k < NSC; ++k) {
        double m[4], b[2];
        m[0] = (ppp[2] - ppp[0]) / (double)(SWIN - 1); m[1] = 0;
        m[2] = 0; m[3] = (ppp[3] - ppp[1]) / (double)(SWIN - 1);
        b[0] = ppp[0]; b[1] = ppp[1];
        pvo_transform_image_rgb(rgb, h, w, m, b, chip, SWIN, SWIN);
        pvo_fhog(chip, SWIN, SWIN, 4, 1, 1, hog);
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c)
                for (int j = 0; j < 32; ++j) {
                    float v;
                    if (j < 31) v = hog[(r * 4 + c) * PVO_FHOG_STRIDE + j];
                    else {
                        const uint8_t* q = chip + (r * SWIN + c) * 3;
                        v = (float)(((unsigned)q[0] + q[1] + q[2]) / 3) / 255.0f;
                    }
                    double* f = tk->Fs + (((size_t)(r * 4 + c) * 32 + j) * NSC + k) * 2;
                    f[0] = (double)v * tk->tb.mask_scale[k];
                    f[1] = 0.0;
                }
        scale_rect(ppp, ALPHA);
    }
    for (int i = 0; i < SDIM; ++i) fft1d(tk->Fs + (size_t)i * NSC * 2, 1, NSC, 5, tk->tb.tw32, 0);
}

static void scale_target(const pvo_tracker* tk, double pos, double* g /* [NSC][2] */)
{
    for (int i = 0; i < NSC; ++i) {
        const double dist = fabs((double)i - pos);
        g[2 * i] = pvo_det_exp(-dist / 1.000);
        g[2 * i + 1] = 0;
    }
    fft1d(g, 1, NSC, 5, tk->tb.tw32, 0);
    for (int i = 0; i < NSC; ++i) g[2 * i + 1] = -g[2 * i + 1];
}

pvo_tracker* pvo_tracker_new(const pvo_dsst_tables* t)
{
    pvo_tracker* tk = (pvo_tracker*)calloc(1, sizeof(pvo_tracker));
    tk->tb = *t;
    tk->F = (double*)calloc((size_t)NPL * FS * FS * 2, sizeof(double));
    tk->A = (double*)calloc((size_t)NPL * FS * FS * 2, sizeof(double));
    tk->B = (double*)calloc((size_t)FS * FS, sizeof(double));
    tk->Fs = (double*)calloc((size_t)SDIM * NSC * 2, sizeof(double));
    tk->As = (double*)calloc((size_t)SDIM * NSC * 2, sizeof(double));
    return tk;
}
void pvo_tracker_free(pvo_tracker* tk)
{
    if (!tk) return;
    free(tk->F); free(tk->A); free(tk->B); free(tk->Fs); free(tk->As); free(tk);
}
void pvo_tracker_position(const pvo_tracker* tk, double box[4]) { memcpy(box, tk->pos, sizeof(double) * 4); }
void pvo_tracker_debug_F(const pvo_tracker* tk, double* out) { memcpy(out, tk->F, sizeof(double) * NPL * FS * FS * 2); }
void pvo_tracker_debug_state(const pvo_tracker* tk, double* A, double* B)
{
    memcpy(A, tk->A, sizeof(double) * NPL * FS * FS * 2);
    memcpy(B, tk->B, sizeof(double) * FS * FS);
}

void pvo_tracker_start(pvo_tracker* tk, const uint8_t* rgb, int h, int w, const double box[4])
{
    double map[4];
    make_chip(tk, rgb, h, w, box, map);
    /* object centre in chip coordinates */
    const double cx = ((box[0] + box[2]) / 2 - map[0]) / map[2];
    const double cy = ((box[1] + box[3]) / 2 - map[1]) / map[3];
    double* G = (double*)malloc(sizeof(double) * FS * FS * 2);
    target_image(tk, cx, cy, G);
    for (int q = 0; q < FS * FS; ++q) {
        double bsum = 0;
        for (int i = 0; i < NPL; ++i) {
            const double* f = tk->F + ((size_t)i * FS * FS + q) * 2;
            double* a = tk->A + ((size_t)i * FS * FS + q) * 2;
            a[0] = G[2 * q] * f[0] - G[2 * q + 1] * f[1];
            a[1] = G[2 * q] * f[1] + G[2 * q + 1] * f[0];
            bsum = bsum + (f[0] * f[0] + f[1] * f[1]);
        }
        tk->B[q] = bsum;
    }
    free(G);
    memcpy(tk->pos, box, sizeof(double) * 4);
    make_scale_space(tk, rgb, h, w);
    double Gs[NSC * 2];
    scale_target(tk, NSC / 2, Gs);
    for (int k = 0; k < NSC; ++k) {
        double bsum = 0;
        for (int i = 0; i < SDIM; ++i) {
            const double* f = tk->Fs + ((size_t)i * NSC + k) * 2;
            double* a = tk->As + ((size_t)i * NSC + k) * 2;
            a[0] = Gs[2 * k] * f[0] - Gs[2 * k + 1] * f[1];
            a[1] = Gs[2 * k] * f[1] + Gs[2 * k + 1] * f[0];
            bsum = bsum + (f[0] * f[0] + f[1] * f[1]);
        }
        tk->Bs[k] = bsum;
    }
}

/* [EXT max_point_interpolated]: least-squares quadratic surface on the 3x3 neighbourhood of the arg-max */
static void peak_interp(const double* R, int n, int* ipx, int* ipy, double* ox, double* oy)
{
    int bi = 0;
    double bv = R[0];
    for (int i = 1; i < n * n; ++i) if (R[i] > bv) { bv = R[i]; bi = i; }
    const int py = bi / n, px = bi % n;
    *ipx = px; *ipy = py; *ox = px; *oy = py;
    if (px < 1 || py < 1 || px > n - 2 || py > n - 2) return;
    double z[3][3];
    for (int r = -1; r <= 1; ++r) for (int c = -1; c <= 1; ++c) z[r + 1][c + 1] = R[(py + r) * n + (px + c)];
    const double sx = ((z[0][2] + z[1][2]) + z[2][2]) - ((z[0][0] + z[1][0]) + z[2][0]);
    const double sy = ((z[2][0] + z[2][1]) + z[2][2]) - ((z[0][0] + z[0][1]) + z[0][2]);
    const double sxy = (z[0][0] + z[2][2]) - (z[0][2] + z[2][0]);
    const double sxx = ((z[0][0] + z[1][0]) + z[2][0]) + ((z[0][2] + z[1][2]) + z[2][2]);
    const double syy = ((z[0][0] + z[0][1]) + z[0][2]) + ((z[2][0] + z[2][1]) + z[2][2]);
    const double sall = sxx + ((z[0][1] + z[1][1]) + z[2][1]);
    const double k2 = sx / 6.0, k3 = sy / 6.0, k5 = sxy / 4.0;
    const double k4 = sxx / 2.0 - sall / 3.0, k6 = syy / 2.0 - sall / 3.0;
    const double h00 = 2 * k4, h01 = k5, h11 = 2 * k6;
    const double det = h00 * h11 - h01 * h01;
    if (det == 0) return;
    double dx = -((h11 * k2 - h01 * k3) / det);
    double dy = -((h00 * k3 - h01 * k2) / det);
    if (dx * k2 + dy * k3 < 0) return;
    if (dx < -1) dx = -1;
    if (dx > 1) dx = 1;
    if (dy < -1) dy = -1;
    if (dy > 1) dy = 1;
    *ox = px + dx; *oy = py + dy;
}

double pvo_tracker_update(pvo_tracker* tk, const uint8_t* rgb, int h, int w)
{
    double guess[4];
    memcpy(guess, tk->pos, sizeof guess);
    double map[4];
    make_chip(tk, rgb, h, w, guess, map);
    double* G = (double*)malloc(sizeof(double) * FS * FS * 2);
    for (int q = 0; q < FS * FS; ++q) {
        double gr = 0, gi = 0;
        for (int i = 0; i < NPL; ++i) {
            const double* f = tk->F + ((size_t)i * FS * FS + q) * 2;
            const double* a = tk->A + ((size_t)i * FS * FS + q) * 2;
            gr = gr + (f[0] * a[0] + f[1] * a[1]);
            gi = gi + (f[1] * a[0] - f[0] * a[1]);
        }
        const double rec = 1.0 / (tk->B[q] + REG_SPACE);
        G[2 * q] = gr * rec; G[2 * q + 1] = gi * rec;
    }
    pvo_fft64x64(G, tk->tb.tw64, 1);
    double* R = (double*)malloc(sizeof(double) * FS * FS);
    for (int q = 0; q < FS * FS; ++q) R[q] = G[2 * q];
    int ipx, ipy;
    double ppx, ppy;
    peak_interp(R, FS, &ipx, &ipy, &ppx, &ppy);
    /* PSR: point p = pp (rounded); exclude centered_rect(p,8,8) = [p-4, p+3] */
    const long rx = (long)floor(ppx + 0.5), ry = (long)floor(ppy + 0.5);
    double sum = 0, sumsq = 0, cnt = 0;
    for (int r = 0; r < FS; ++r) {
        double rs = 0, rq = 0;
        for (int c = 0; c < FS; ++c) {
            if (c >= rx - 4 && c <= rx + 3 && r >= ry - 4 && r <= ry + 3) continue;
            rs = rs + R[r * FS + c];
            rq = rq + R[r * FS + c] * R[r * FS + c];
            cnt += 1;
        }
        sum = sum + rs; sumsq = sumsq + rq;
    }
    const double mean = sum / cnt;
    double var = (1.0 / (cnt - 1)) * (sumsq - sum * sum / cnt);
    if (!(var >= 0)) var = 0;
    long qx = rx, qy = ry;
    if (qx < 0) qx = 0;
    if (qy < 0) qy = 0;
    if (qx > FS - 1) qx = FS - 1;
    if (qy > FS - 1) qy = FS - 1;
    const double psr = (R[qy * FS + qx] - mean) / sqrt(var);
    /* position = translate_rect(guess, tform(pp) - center(guess)) */
    const double ix = map[0] + ppx * map[2], iy = map[1] + ppy * map[3];
    const double vx = ix - (guess[0] + guess[2]) / 2, vy = iy - (guess[1] + guess[3]) / 2;
    tk->pos[0] = guess[0] + vx; tk->pos[1] = guess[1] + vy; tk->pos[2] = guess[2] + vx; tk->pos[3] = guess[3] + vy;
    /* filter update */
    target_image(tk, ppx, ppy, G);
    for (int q = 0; q < FS * FS; ++q) {
        double bq = tk->B[q] * (1 - NU_SPACE);
        for (int i = 0; i < NPL; ++i) {
            const double* f = tk->F + ((size_t)i * FS * FS + q) * 2;
            double* a = tk->A + ((size_t)i * FS * FS + q) * 2;
            const double nr = G[2 * q] * f[0] - G[2 * q + 1] * f[1];
            const double ni = G[2 * q] * f[1] + G[2 * q + 1] * f[0];
            a[0] = NU_SPACE * nr + (1 - NU_SPACE) * a[0];
            a[1] = NU_SPACE * ni + (1 - NU_SPACE) * a[1];
            bq = bq + NU_SPACE * (f[0] * f[0] + f[1] * f[1]);
        }
        tk->B[q] = bq;
    }
    free(G); free(R);

    /* scale */
    make_scale_space(tk, rgb, h, w);
    double Gs[NSC * 2], Rs[NSC];
    for (int k = 0; k < NSC; ++k) {
        double gr = 0, gi = 0;
        for (int i = 0; i < SDIM; ++i) {
            const double* f = tk->Fs + ((size_t)i * NSC + k) * 2;
            const double* a = tk->As + ((size_t)i * NSC + k) * 2;
            gr = gr + (f[0] * a[0] + f[1] * a[1]);
            gi = gi + (f[1] * a[0] - f[0] * a[1]);
        }
        const double rec = 1.0 / (tk->Bs[k] + REG_SCALE);
        Gs[2 * k] = gr * rec; Gs[2 * k + 1] = gi * rec;
    }
    fft1d(Gs, 1, NSC, 5, tk->tb.tw32, 1);
    for (int k = 0; k < NSC; ++k) Rs[k] = Gs[2 * k];
    int bk = 0;
    for (int k = 1; k < NSC; ++k) if (Rs[k] > Rs[bk]) bk = k;
    double pos = bk;
    if (bk > 0 && bk + 1 < NSC) {
        /* lagrange_poly_min_extrap(p1,p2,p3, -v1,-v2,-v3) */
        const double p1 = bk - 1, p2 = bk, p3 = bk + 1, f1 = -Rs[bk - 1], f2 = -Rs[bk], f3 = -Rs[bk + 1];
        const double d1 = p2 * p2 - p3 * p3, d2 = p3 * p3 - p1 * p1, d3 = p1 * p1 - p2 * p2;
        const double t1 = (d1 * f1 + d2 * f2) + d3 * f3;
        const double d4 = p2 - p3, d5 = p3 - p1, d6 = p1 - p2;
        const double t2 = 2 * ((d4 * f1 + d5 * f2) + d6 * f3);
        if (t1 != 0 && t2 != 0) {
            pos = t1 / t2;
            if (pos < p1) pos = p1;
            if (pos > p3) pos = p3;
        }
    }
    scale_rect(tk->pos, pvo_det_exp((pos - (double)NSC / 2) * tk->tb.ln_alpha));
    scale_target(tk, pos, Gs);
    for (int k = 0; k < NSC; ++k) {
        double bq = tk->Bs[k] * (1 - NU_SCALE);
        for (int i = 0; i < SDIM; ++i) {
            const double* f = tk->Fs + ((size_t)i * NSC + k) * 2;
            double* a = tk->As + ((size_t)i * NSC + k) * 2;
            const double nr = Gs[2 * k] * f[0] - Gs[2 * k + 1] * f[1];
            const double ni = Gs[2 * k] * f[1] + Gs[2 * k + 1] * f[0];
            a[0] = NU_SCALE * nr + (1 - NU_SCALE) * a[0];
            a[1] = NU_SCALE * ni + (1 - NU_SCALE) * a[1];
            bq = bq + NU_SCALE * (f[0] * f[0] + f[1] * f[1]);
        }
        tk->Bs[k] = bq;
    }
    return psr;
}
