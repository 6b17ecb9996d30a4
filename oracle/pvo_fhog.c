/*
 * pvo_fhog.c -- ORACLE (test infrastructure): Felzenszwalb 31-D HOG as dlib extracts it.
 * PARITY UNPINNED (restated from the published dlib/image_transforms/fhog.h, [EXT]).
 * Used by: frontal face detector (cell 8, padding 10x10)     reference face.py:54,66
 *          correlation tracker  (cell 1 / 3x3, cell 4 / 1x1)  reference tracking.py:203,250-251
 *
 * Summation orders (the HIP kernels reproduce them exactly):
 *   hist[cell][bin]: contributions added in row-major pixel order, starting from 0;
 *   norm[cell]     : o = 0..8 sequential;
 *   block sums     : (((z1+z2)+z3)+z4)+eps ; 4-lane sum = (h0+h1)+(h2+h3) ; t += (a+b)+c.
 */
#include "pvo.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

static const float DIRX[9] = {1.0000f, 0.9397f, 0.7660f, 0.500f, 0.1736f, -0.1736f, -0.5000f, -0.7660f, -0.9397f};
static const float DIRY[9] = {0.0000f, 0.3420f, 0.6428f, 0.8660f, 0.9848f, 0.9848f, 0.8660f, 0.6428f, 0.3420f};

/* gradient of the colour channel with the largest magnitude (first wins ties); returns |g|^2 */
static inline float gradient(const uint8_t* img, int iw, int y, int x, float* gx, float* gy)
{
    const uint8_t* pl = img + ((size_t)y * iw + (x - 1)) * 3;
    const uint8_t* pr = img + ((size_t)y * iw + (x + 1)) * 3;
    const uint8_t* pu = img + ((size_t)(y - 1) * iw + x) * 3;
    const uint8_t* pd = img + ((size_t)(y + 1) * iw + x) * 3;
    int bx = (int)pr[0] - (int)pl[0], by = (int)pd[0] - (int)pu[0];
    int bv = bx * bx + by * by;
    for (int k = 1; k < 3; ++k) {
        const int cx = (int)pr[k] - (int)pl[k], cy = (int)pd[k] - (int)pu[k];
        const int cv = cx * cx + cy * cy;
        if (cv > bv) { bv = cv; bx = cx; by = cy; }
    }
    *gx = (float)bx; *gy = (float)by;
    return (float)bv;
}

static inline int snap_orientation(float gx, float gy)
{
    float best_dot = 0;
    int best_o = 0;
    for (int o = 0; o < 9; ++o) {
        const float dot = gx * DIRX[o] + gy * DIRY[o];
        if (dot > best_dot) { best_dot = dot; best_o = o; }
        else if (-dot > best_dot) { best_dot = -dot; best_o = o + 9; }
    }
    return best_o;
}

void pvo_fhog_dims(int ih, int iw, int cell, int pad_r, int pad_c, int* fh, int* fw)
{
    int hog_nr, hog_nc;
    if (cell == 1) { hog_nr = ih - 2; hog_nc = iw - 2; }
    else {
        const int cells_nr = (int)((double)ih / (double)cell + 0.5);
        const int cells_nc = (int)((double)iw / (double)cell + 0.5);
        hog_nr = cells_nr - 2; hog_nc = cells_nc - 2;
    }
    if (hog_nr <= 0 || hog_nc <= 0) { *fh = 0; *fw = 0; return; }
    *fh = hog_nr + pad_r - 1;
    *fw = hog_nc + pad_c - 1;
}

/* 27 + 4 features of one cell from its 18-bin histogram h and the 3x3 neighbourhood of norms
 * n[0..8] (row-major, centre = n[4]).  o points at 32 floats. */
static void cell_features(const float* h, const float* n, float* o)
{
    const float eps = 0.0001f;
    /* lanes: 0:(y+1,x+1) block  1:(y,x+1)  2:(y+1,x)  3:(y,x)  in the notation of fhog.h */
    const float z1[4] = {n[4], n[1], n[3], n[0]};
    const float z2[4] = {n[5], n[2], n[4], n[1]};
    const float z3[4] = {n[7], n[4], n[6], n[3]};
    const float z4[4] = {n[8], n[5], n[7], n[4]};
    float nn[4], nv[4], t[4] = {0, 0, 0, 0};
    for (int k = 0; k < 4; ++k) {
        nn[k] = 0.2f * sqrtf((((z1[k] + z2[k]) + z3[k]) + z4[k]) + eps);
        nv[k] = 0.1f / nn[k];
    }
    for (int g = 0; g < 18; g += 3) {
        float hh[3][4];
        for (int j = 0; j < 3; ++j) {
            for (int k = 0; k < 4; ++k) hh[j][k] = fminf(h[g + j], nn[k]) * nv[k];
            o[g + j] = (hh[j][0] + hh[j][1]) + (hh[j][2] + hh[j][3]);
        }
        for (int k = 0; k < 4; ++k) t[k] = t[k] + ((hh[0][k] + hh[1][k]) + hh[2][k]);
    }
    const float tscale = (float)(2 * 0.2357);
    for (int k = 0; k < 4; ++k) t[k] = t[k] * tscale;
    for (int g = 0; g < 9; ++g) {
        const float s = h[g] + h[g + 9];
        float hh[4];
        for (int k = 0; k < 4; ++k) hh[k] = fminf(s, nn[k]) * nv[k];
        o[18 + g] = (hh[0] + hh[1]) + (hh[2] + hh[3]);
    }
    o[27] = t[0]; o[28] = t[1]; o[29] = t[2]; o[30] = t[3];
    o[31] = 0.0f;
}

static void fhog_cell1(const uint8_t* img, int ih, int iw, int pad_r, int pad_c, float* out, int fh, int fw)
{
    /* [EXT impl_extract_fhog_features_cell_size_1]: every pixel is a cell; norm = |g|^2, hist has the
     * single bin `angle` holding |g|. */
    float* norm = (float*)calloc((size_t)ih * iw, sizeof(float));
    uint8_t* angle = (uint8_t*)calloc((size_t)ih * iw, 1);
    for (int y = 1; y < ih - 1; ++y)
        for (int x = 1; x < iw - 1; ++x) {
            float gx, gy;
            const float v = gradient(img, iw, y, x, &gx, &gy);
            norm[(size_t)y * iw + x] = v;
            angle[(size_t)y * iw + x] = (uint8_t)snap_orientation(gx, gy);
        }
    const int hog_nr = ih - 2, hog_nc = iw - 2;
    const int oy = (pad_r - 1) / 2, ox = (pad_c - 1) / 2;
    for (int y = 0; y < hog_nr; ++y)
        for (int x = 0; x < hog_nc; ++x) {
            float n[9], h[18];
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) n[i * 3 + j] = norm[(size_t)(y + i) * iw + (x + j)];
            for (int o = 0; o < 18; ++o) h[o] = 0;
            h[angle[(size_t)(y + 1) * iw + (x + 1)]] = sqrtf(n[4]);
            cell_features(h, n, out + ((size_t)(y + oy) * fw + (x + ox)) * PVO_FHOG_STRIDE);
        }
    free(norm); free(angle);
    (void)fh;
}

void pvo_fhog(const uint8_t* img, int ih, int iw, int cell, int pad_r, int pad_c, float* out)
{
    int fh, fw;
    pvo_fhog_dims(ih, iw, cell, pad_r, pad_c, &fh, &fw);
    if (fh == 0 || fw == 0) return;
    memset(out, 0, (size_t)fh * fw * PVO_FHOG_STRIDE * sizeof(float));
    if (cell == 1) { fhog_cell1(img, ih, iw, pad_r, pad_c, out, fh, fw); return; }

    const int cells_nr = (int)((double)ih / (double)cell + 0.5);
    const int cells_nc = (int)((double)iw / (double)cell + 0.5);
    const int hr = cells_nr + 2, hc = cells_nc + 2;
    float* hist = (float*)calloc((size_t)hr * hc * 18, sizeof(float));
    float* norm = (float*)calloc((size_t)cells_nr * cells_nc, sizeof(float));
    const int visible_nr = (cells_nr * cell < ih ? cells_nr * cell : ih) - 1;
    const int visible_nc = (cells_nc * cell < iw ? cells_nc * cell : iw) - 1;

    /* dlib's loop visits the pixels in row-major order and scatters each vote into 4 cells, so a cell's bins are summed in
     * row-major order of the pixels that vote for it.  For the all-core CPU baseline the histogram ROWS are dealt out to threads:
     * each thread walks the pixel rows that vote into its rows, in the same order, and adds only to its own rows -- the sums are
     * formed in exactly the sequential order, for any thread count (band = 1 reproduces the plain loop). */
    const int n_bands = (ih * iw > 65536) ? omp_get_max_threads() : 1;
    const int band_rows = (hr + n_bands - 1) / n_bands;
    #pragma omp parallel for schedule(static) if (n_bands > 1)
    for (int band = 0; band < n_bands; ++band) {
        const int r_lo = band * band_rows, r_hi = (r_lo + band_rows < hr) ? r_lo + band_rows : hr;   /* histogram rows [r_lo, r_hi) */
        if (r_lo >= r_hi) continue;
        /* pixel row y votes into histogram rows iyp + 1 and iyp + 2 with iyp + 1 = (y + 4) / 8 (cell = 8); generic: compute and test */
        for (int y = 1; y < visible_nr; ++y) {
            const float yp = ((float)y + 0.5f) / (float)cell - 0.5f;
            const int iyp = (int)floorf(yp);
            const int ra = iyp + 1, rb2 = iyp + 2;
            const int in_a = (ra >= r_lo && ra < r_hi), in_b = (rb2 >= r_lo && rb2 < r_hi);
            if (!in_a && !in_b) continue;
            const float vy0 = yp - (float)iyp;
            const float vy1 = 1.0f - vy0;
            for (int x = 1; x < visible_nc; ++x) {
                float gx, gy;
                float v = gradient(img, iw, y, x, &gx, &gy);
                const int bo = snap_orientation(gx, gy);
                v = sqrtf(v);
                const float xp = ((float)x + 0.5f) / (float)cell - 0.5f;
                const int ixp = (int)floorf(xp);
                const float vx0 = xp - (float)ixp;
                const float vx1 = 1.0f - vx0;
                if (in_a) hist[((size_t)(iyp + 1) * hc + (ixp + 1)) * 18 + bo] += (vy1 * vx1) * v;
                if (in_b) hist[((size_t)(iyp + 2) * hc + (ixp + 1)) * 18 + bo] += (vy0 * vx1) * v;
                if (in_a) hist[((size_t)(iyp + 1) * hc + (ixp + 2)) * 18 + bo] += (vy1 * vx0) * v;
                if (in_b) hist[((size_t)(iyp + 2) * hc + (ixp + 2)) * 18 + bo] += (vy0 * vx0) * v;
            }
        }
    }
    #pragma omp parallel for schedule(static) if (cells_nr * cells_nc > 4096)
    for (int r = 0; r < cells_nr; ++r)
        for (int c = 0; c < cells_nc; ++c) {
            const float* h = hist + ((size_t)(r + 1) * hc + (c + 1)) * 18;
            float acc = 0;
            for (int o = 0; o < 9; ++o) { const float s = h[o] + h[o + 9]; acc = acc + s * s; }
            norm[(size_t)r * cells_nc + c] = acc;
        }
    const int hog_nr = cells_nr - 2, hog_nc = cells_nc - 2;
    const int oy = (pad_r - 1) / 2, ox = (pad_c - 1) / 2;
    #pragma omp parallel for schedule(static) if (hog_nr * hog_nc > 4096)
    for (int y = 0; y < hog_nr; ++y)
        for (int x = 0; x < hog_nc; ++x) {
            float n[9];
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) n[i * 3 + j] = norm[(size_t)(y + i) * cells_nc + (x + j)];
            cell_features(hist + ((size_t)(y + 2) * hc + (x + 2)) * 18, n,
                          out + ((size_t)(y + oy) * fw + (x + ox)) * PVO_FHOG_STRIDE);
        }
    free(hist); free(norm);
}
