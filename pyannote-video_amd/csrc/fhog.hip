// fhog.hip -- FHOG of small images, one launch sequence per batch of equally sized images: the correlation tracker's chips
// (cell 1: 64 x 64 translation windows; cell 4: 23 x 23 scale samples; reference tracking.py:203,250-251) and the stage-access
// entry used by the parity tests (any of cell 1 / 4 / 8).  The detector's pyramid has its own fused kernel (detect.hip).
#include "fhog_dev.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>

// Orientation snap of dlib's FHOG: arg-max over 9 directions of +-dot(direction, gradient), first maximum wins.
// The gradient of a uint8 image is a pair of integers in [-255,255]^2, so the bin is a pure function of 511x511 inputs:
// it is tabulated once on the host with exactly these float operations (mul, mul, add, strict compares; no contraction),
// which makes the table bit-identical to evaluating the chain per pixel.
static const float h_dirx[9] = {1.0000f, 0.9397f, 0.7660f, 0.500f, 0.1736f, -0.1736f, -0.5000f, -0.7660f, -0.9397f};
static const float h_diry[9] = {0.0000f, 0.3420f, 0.6428f, 0.8660f, 0.9848f, 0.9848f, 0.8660f, 0.6428f, 0.3420f};

static std::recursive_mutex g_lut_mu;      // the two tables are built on first use, by whichever side (detector / tracker) gets there first

const uint8_t* orientation_lut(Ctx* c)
{
    std::lock_guard<std::recursive_mutex> lk(g_lut_mu);
    if (c->d_orient_lut) return c->d_orient_lut;
    std::vector<uint8_t> lut((size_t)511 * 511);
    for (int by = -255; by <= 255; ++by)
        for (int bx = -255; bx <= 255; ++bx) {
            const volatile float gx = (float)bx, gy = (float)by;
            float best_dot = 0.0f;
            int best_o = 0;
            for (int o = 0; o < 9; ++o) {
                const volatile float a = gx * h_dirx[o];
                const volatile float b = gy * h_diry[o];
                const float dot = a + b;
                if (dot > best_dot) { best_dot = dot; best_o = o; }
                else if (-dot > best_dot) { best_dot = -dot; best_o = o + 9; }
            }
            lut[(size_t)(by + 255) * 511 + (bx + 255)] = (uint8_t)best_o;
        }
    HIP_CHECK(hipMalloc((void**)&c->d_orient_lut, lut.size()));
    HIP_CHECK(hipMemcpy(c->d_orient_lut, lut.data(), lut.size(), hipMemcpyHostToDevice));
    return c->d_orient_lut;
}

// The same table in 8 x 8 tiles (one 64-byte line each): with Y = by + 255, X = bx + 255 the entry sits at
// (Y >> 3) << 12 | (X >> 3) << 6 | (Y & 7) << 3 | (X & 7).  Neighbouring pixels mostly have small, similar gradients, so a
// wave's 64 look-ups touch a handful of lines instead of one line per table row (the row-major form cost ~50 L1 accesses
// per gather and made the gradient pass texture-addresser bound).
const uint8_t* orientation_lut_tiled(Ctx* c)
{
    std::lock_guard<std::recursive_mutex> lk(g_lut_mu);
    if (c->d_grad_lut) return reinterpret_cast<const uint8_t*>(c->d_grad_lut);
    orientation_lut(c);
    std::vector<uint8_t> ol((size_t)511 * 511);
    HIP_CHECK(hipMemcpy(ol.data(), c->d_orient_lut, ol.size(), hipMemcpyDeviceToHost));
    std::vector<uint8_t> lut((size_t)64 * 64 * 64, 0);
    for (int Y = 0; Y < 511; ++Y)
        for (int X = 0; X < 511; ++X)
            lut[((size_t)(Y >> 3) << 12) | ((size_t)(X >> 3) << 6) | ((Y & 7) << 3) | (X & 7)] = ol[(size_t)Y * 511 + X];
    HIP_CHECK(hipMalloc((void**)&c->d_grad_lut, lut.size()));
    HIP_CHECK(hipMemcpy(c->d_grad_lut, lut.data(), lut.size(), hipMemcpyHostToDevice));
    return reinterpret_cast<const uint8_t*>(c->d_grad_lut);
}

__device__ __forceinline__ void pixel_grad(const uint8_t* __restrict__ row_u, const uint8_t* __restrict__ row_c,
                                           const uint8_t* __restrict__ row_d, int x3, const uint8_t* __restrict__ lut, float* v2, int* bo)
{
    // row_* point at the byte rows; x3 = 3*x (pixel x of the centre row); colour channel with the largest |g|^2, first wins
    int bx = (int)row_c[x3 + 3] - (int)row_c[x3 - 3], by = (int)row_d[x3] - (int)row_u[x3];
    int bv = bx * bx + by * by;
#pragma unroll
    for (int k = 1; k < 3; ++k) {
        const int cx = (int)row_c[x3 + 3 + k] - (int)row_c[x3 - 3 + k], cy = (int)row_d[x3 + k] - (int)row_u[x3 + k];
        const int cv = cx * cx + cy * cy;
        if (cv > bv) { bv = cv; bx = cx; by = cy; }
    }
    *v2 = (float)bv;
    *bo = lut[(by + 255) * 511 + (bx + 255)];
}

// The per-image FHOG kernels below run on 1-D grids over (image, row, column): the images they see in production are the
// trackers' chips (23 x 23 scale samples, 64 x 64 translation windows), whose rows would fill 2-25 % of a 256-lane block each.
__device__ __forceinline__ bool flat_index(int nx, int ny, int nb, int* x, int* y, int* b)
{
    const unsigned g = blockIdx.x * 256u + threadIdx.x;
    const unsigned t = g / (unsigned)nx;
    *x = (int)(g - t * (unsigned)nx);
    *b = (int)(t / (unsigned)ny);
    *y = (int)(t - (unsigned)*b * (unsigned)ny);
    return *b < nb;
}
static inline dim3 flat_grid(int nx, int ny, int nb)
{
    const size_t total = (size_t)nx * ny * nb;
    PVF_REQUIRE(total < ((size_t)1 << 31), "fhog: too many work items for one launch");
    return dim3((unsigned)((total + 255) / 256));
}

// Pass 1: per pixel (orientation bin, gradient magnitude) into planes shifted by 3C/2 so that histogram cell (hy,hx) owns rows
// yy in [C*hy, C*hy+2C) and columns xx in [C*hx, C*hx+2C)  (yy = y + 3C/2, xx = x + 3C/2; pitch = multiple of 16 floats).
// One lane = 4 consecutive pixels of one row; interior quads fetch their 3 x 18-byte neighbourhood with
// 11 (unaligned) dword loads and pick the bytes with constant shifts; results leave as one float4 + one packed dword.
__device__ __forceinline__ void grad_from_bytes(const int u[3], const int d[3], const int l[3], const int r[3],
                                                const uint8_t* __restrict__ lut, float* v, int* o)
{
    int bx = r[0] - l[0], by = d[0] - u[0];
    int bv = bx * bx + by * by;
#pragma unroll
    for (int k = 1; k < 3; ++k) {
        const int cx = r[k] - l[k], cy = d[k] - u[k];
        const int cv = cx * cx + cy * cy;
        if (cv > bv) { bv = cv; bx = cx; by = cy; }
    }
    *v = sqrtf((float)bv);
    *o = lut[(by + 255) * 511 + (bx + 255)];
}

template <int C>
__global__ void __launch_bounds__(256) fhog_grad4_k(const uint8_t* __restrict__ img, size_t img_stride, int ih, int iw, int visible_nr,
                                                    int visible_nc, float* __restrict__ mag, uint8_t* __restrict__ bin, size_t px_stride,
                                                    int rows_t, int pitch, const uint8_t* __restrict__ lut, int n_img)
{
    int q, yy, b;                                       // quad index within the row, plane row, image
    if (!flat_index(pitch / 4, rows_t, n_img, &q, &yy, &b)) return;
    const int xx = 4 * q;
    const int y = yy - 3 * C / 2, x0 = xx - 3 * C / 2;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    int o[4] = {0, 0, 0, 0};
    if (y >= 1 && y < visible_nr && x0 + 3 >= 1 && x0 < visible_nc) {
        const uint8_t* im = img + (size_t)b * img_stride;
        const int rb = iw * 3;
        const uint8_t* rc = im + (size_t)y * rb;
        const uint8_t* ru = rc - rb;
        const uint8_t* rd = rc + rb;
        if (x0 >= 1 && x0 + 4 <= visible_nc && x0 + 6 <= iw) {
            uint32_t wc[5], wu[3], wd[3];
            const uint8_t* pc = rc + 3 * x0 - 3;
#pragma unroll
            for (int k = 0; k < 5; ++k) wc[k] = *reinterpret_cast<const uint32_t*>(pc + 4 * k);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                wu[k] = *reinterpret_cast<const uint32_t*>(ru + 3 * x0 + 4 * k);
                wd[k] = *reinterpret_cast<const uint32_t*>(rd + 3 * x0 + 4 * k);
            }
#define BYTE_OF(w, i) (int)(((w)[(i) >> 2] >> (8 * ((i) & 3))) & 0xffu)
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                int u[3], d[3], l[3], r[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    u[k] = BYTE_OF(wu, 3 * p + k); d[k] = BYTE_OF(wd, 3 * p + k);
                    l[k] = BYTE_OF(wc, 3 * p + k); r[k] = BYTE_OF(wc, 3 * p + 6 + k);
                }
                grad_from_bytes(u, d, l, r, lut, &v[p], &o[p]);
            }
#undef BYTE_OF
        } else {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int x = x0 + p;
                if (x >= 1 && x < visible_nc) {
                    int u[3], d[3], l[3], r[3];
#pragma unroll
                    for (int k = 0; k < 3; ++k) { u[k] = ru[3 * x + k]; d[k] = rd[3 * x + k]; l[k] = rc[3 * x - 3 + k]; r[k] = rc[3 * x + 3 + k]; }
                    grad_from_bytes(u, d, l, r, lut, &v[p], &o[p]);
                }
            }
        }
    }
    const size_t idx = (size_t)b * px_stride + (size_t)yy * pitch + xx;
    *reinterpret_cast<float4*>(mag + idx) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<uint32_t*>(bin + idx) = (uint32_t)o[0] | ((uint32_t)o[1] << 8) | ((uint32_t)o[2] << 16) | ((uint32_t)o[3] << 24);
}

// Pass 2: one lane per histogram cell walks its 2C x 2C window in row-major order (== the order dlib's scatter loop adds in),
// adding each vote to the bin's running sum kept in LDS (acc[bin][lane]: conflict-free).  Rows are read with 16-byte loads.
template <int C>
__global__ void __launch_bounds__(256) fhog_hist_k(const float* __restrict__ mag, const uint8_t* __restrict__ bin, size_t px_stride, int pitch,
                                                   float* __restrict__ hist, size_t hist_stride, int hr, int hc,
                                                   float* __restrict__ norm, size_t norm_stride, int cells_nr, int cells_nc, int n_img)
{
    __shared__ float acc[18][256];
    int hx, hy, b;
    const bool valid = flat_index(hc, hr, n_img, &hx, &hy, &b);
    const int tid = threadIdx.x;
#pragma unroll
    for (int o = 0; o < 18; ++o) acc[o][tid] = 0.0f;
    if (valid) {
        const float* mg = mag + (size_t)b * px_stride + (size_t)C * hx;
        const uint8_t* bn = bin + (size_t)b * px_stride + (size_t)C * hx;
        constexpr int NV = 2 * C / 4;            // float4 loads per row
        float4 pv[2][NV];
        uint32_t pb[2][4];
        auto load_row = [&](int wy, float4* dv, uint32_t* db) {
            const size_t row = (size_t)(C * hy + wy) * pitch;
#pragma unroll
            for (int q = 0; q < NV; ++q) dv[q] = *reinterpret_cast<const float4*>(mg + row + 4 * q);
            if (C == 8) {
                const uint4 t = *reinterpret_cast<const uint4*>(bn + row);
                db[0] = t.x; db[1] = t.y; db[2] = t.z; db[3] = t.w;
            } else {
                const uint2 t = *reinterpret_cast<const uint2*>(bn + row);
                db[0] = t.x; db[1] = t.y; db[2] = 0; db[3] = 0;
            }
        };
        load_row(0, pv[0], pb[0]);
#pragma unroll
        for (int wy = 0; wy < 2 * C; ++wy) {
            const int cur = wy & 1;
            if (wy + 1 < 2 * C) load_row(wy + 1, pv[cur ^ 1], pb[cur ^ 1]);   // next row is in flight while this one is accumulated
            const int i = wy % C;
            const float fy = ((float)i + 0.5f) / (float)C;
            const float wyv = (wy < C) ? fy : 1.0f - fy;
            float v[2 * C];
#pragma unroll
            for (int q = 0; q < NV; ++q) { v[4 * q] = pv[cur][q].x; v[4 * q + 1] = pv[cur][q].y; v[4 * q + 2] = pv[cur][q].z; v[4 * q + 3] = pv[cur][q].w; }
#pragma unroll
            for (int wx = 0; wx < 2 * C; ++wx) {
                const int j = wx % C;
                const float fx = ((float)j + 0.5f) / (float)C;
                const float wxv = (wx < C) ? fx : 1.0f - fx;
                const int o = (int)((pb[cur][wx >> 2] >> (8 * (wx & 3))) & 0xffu);
                acc[o][tid] = acc[o][tid] + (wyv * wxv) * v[wx];
            }
        }
        float* h = hist + (size_t)b * hist_stride + ((size_t)hy * hc + hx) * 18;
        float e = 0.0f;
#pragma unroll
        for (int o = 0; o < 9; ++o) {
            const float a0 = acc[o][tid], a1 = acc[o + 9][tid];
            h[o] = a0; h[o + 9] = a1;
            const float s2 = a0 + a1;
            e = e + s2 * s2;
        }
        if (hy >= 1 && hy <= cells_nr && hx >= 1 && hx <= cells_nc)
            norm[(size_t)b * norm_stride + (size_t)(hy - 1) * cells_nc + (hx - 1)] = e;
    }
}

__global__ void __launch_bounds__(256) fhog_feat_k(const float* __restrict__ hist, size_t hist_stride, int hc, const float* __restrict__ norm,
                                                   size_t norm_stride, int cells_nc, float* __restrict__ feat, size_t feat_stride, int fw,
                                                   int hog_nr, int hog_nc, int oy, int ox, int fh, int n_img)
{
    int px, py, b;                                             // padded output coordinates, image
    if (!flat_index(fw, fh, n_img, &px, &py, &b)) return;
    const int x = px - ox, y = py - oy;
    if (x < 0 || y < 0 || x >= hog_nc || y >= hog_nr) {
        float4* z = reinterpret_cast<float4*>(feat + (size_t)b * feat_stride + ((size_t)py * fw + px) * PVF_FHOG_STRIDE);
#pragma unroll
        for (int k = 0; k < 8; ++k) z[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    float n[9], h[18], o[32];
    const float* nb = norm + (size_t)b * norm_stride;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) n[i * 3 + j] = nb[(size_t)(y + i) * cells_nc + (x + j)];
    const float* hp = hist + (size_t)b * hist_stride + ((size_t)(y + 2) * hc + (x + 2)) * 18;
#pragma unroll
    for (int k = 0; k < 18; ++k) h[k] = hp[k];
    cell_features(h, n, o);
    float4* dst = reinterpret_cast<float4*>(feat + (size_t)b * feat_stride + ((size_t)(y + oy) * fw + (x + ox)) * PVF_FHOG_STRIDE);
#pragma unroll
    for (int k = 0; k < 8; ++k) dst[k] = make_float4(o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]);
}

// cell size 1 (correlation tracker translation chip): every pixel is a cell
__global__ void __launch_bounds__(256) fhog1_grad_k(const uint8_t* __restrict__ img, size_t img_stride, int ih, int iw,
                                                    float* __restrict__ norm, uint8_t* __restrict__ angle, size_t px_stride, const uint8_t* __restrict__ lut,
                                                    int n_img)
{
    int x, y, b;
    if (!flat_index(iw, ih, n_img, &x, &y, &b)) return;
    float v = 0.0f;
    int o = 0;
    if (y >= 1 && y < ih - 1 && x >= 1 && x < iw - 1) {
        const uint8_t* im = img + (size_t)b * img_stride;
        pixel_grad(im + (size_t)(y - 1) * iw * 3, im + (size_t)y * iw * 3, im + (size_t)(y + 1) * iw * 3, x * 3, lut, &v, &o);
    }
    norm[(size_t)b * px_stride + (size_t)y * iw + x] = v;
    angle[(size_t)b * px_stride + (size_t)y * iw + x] = (uint8_t)o;
}

__global__ void __launch_bounds__(256) fhog1_feat_k(const float* __restrict__ norm, const uint8_t* __restrict__ angle, size_t px_stride,
                                                    int iw, float* __restrict__ feat, size_t feat_stride, int fw, int hog_nr, int hog_nc,
                                                    int oy, int ox, int fh, int n_img, int planes)
{
    int px, py, b;
    if (!flat_index(fw, fh, n_img, &px, &py, &b)) return;
    const int x = px - ox, y = py - oy;
    // planes: [image][31][fh * fw] (a plane is contiguous: the correlation tracker transforms plane by plane) instead of [image][cell][32]
    float* pl = feat + (size_t)b * feat_stride + (size_t)py * fw + px;
    const size_t plane = (size_t)fh * fw;
    if (x < 0 || y < 0 || x >= hog_nc || y >= hog_nr) {
        if (planes) {
#pragma unroll
            for (int k = 0; k < 31; ++k) pl[k * plane] = 0.f;
            return;
        }
        float4* z = reinterpret_cast<float4*>(feat + (size_t)b * feat_stride + ((size_t)py * fw + px) * PVF_FHOG_STRIDE);
#pragma unroll
        for (int k = 0; k < 8; ++k) z[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    float n[9], h[18], o[32];
    const float* nb = norm + (size_t)b * px_stride;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) n[i * 3 + j] = nb[(size_t)(y + i) * iw + (x + j)];
    const int a = angle[(size_t)b * px_stride + (size_t)(y + 1) * iw + (x + 1)];
    const float mag = sqrtf(n[4]);
#pragma unroll
    for (int k = 0; k < 18; ++k) h[k] = (k == a) ? mag : 0.0f;
    cell_features(h, n, o);
    if (planes) {
#pragma unroll
        for (int k = 0; k < 31; ++k) pl[k * plane] = o[k];
        return;
    }
    float4* dst = reinterpret_cast<float4*>(feat + (size_t)b * feat_stride + ((size_t)(y + oy) * fw + (x + ox)) * PVF_FHOG_STRIDE);
#pragma unroll
    for (int k = 0; k < 8; ++k) dst[k] = make_float4(o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]);
}

void fhog_dims(int ih, int iw, int cell, int pad_r, int pad_c, int* fh, int* fw)
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

void fhog_device(Ctx* c, const uint8_t* d_img, int n, int h, int w, int cell, int pad_r, int pad_c, float* d_feat, DevBuf& hist, DevBuf& norm,
                 size_t img_stride_in, bool planes)
{
    PVF_REQUIRE(!planes || cell == 1, "fhog: plane-major output exists for cell size 1 only");
    DevBuf& grad = c->s_grad;
    const uint8_t* lut = orientation_lut(c);
    int fh, fw;
    fhog_dims(h, w, cell, pad_r, pad_c, &fh, &fw);
    PVF_REQUIRE(fh > 0 && fw > 0, "fhog: image too small");
    const size_t feat_stride = (size_t)fh * fw * PVF_FHOG_STRIDE;
    const int oy = (pad_r - 1) / 2, ox = (pad_c - 1) / 2;
    const size_t img_stride = img_stride_in ? img_stride_in : (size_t)h * w * 3;
    if (cell == 1) {
        const size_t px = (size_t)h * w;
        norm.ensure(px * n * sizeof(float));
        hist.ensure(px * n);
        hipLaunchKernelGGL(fhog1_grad_k, flat_grid(w, h, n), dim3(256), 0, c->stream, d_img, img_stride, h, w, norm.as<float>(), hist.as<uint8_t>(), px, lut, n);
        hipLaunchKernelGGL(fhog1_feat_k, flat_grid(fw, fh, n), dim3(256), 0, c->stream, norm.as<float>(), hist.as<uint8_t>(), px, w, d_feat, feat_stride, fw,
                           h - 2, w - 2, oy, ox, fh, n, planes ? 1 : 0);
        return;
    }
    PVF_REQUIRE(cell == 8 || cell == 4, "fhog: cell size 1, 4 or 8");
    const int cells_nr = (int)((double)h / (double)cell + 0.5), cells_nc = (int)((double)w / (double)cell + 0.5);
    const int hr = cells_nr + 2, hc = cells_nc + 2;
    const int visible_nr = std::min(cells_nr * cell, h) - 1, visible_nc = std::min(cells_nc * cell, w) - 1;
    const size_t hist_stride = (size_t)hr * hc * 18, norm_stride = (size_t)cells_nr * cells_nc;
    hist.ensure(hist_stride * n * sizeof(float));
    norm.ensure(norm_stride * n * sizeof(float));
    // pass 1: (bin, magnitude) planes in the cell-blocked layout; pass 2: per-cell ordered accumulation + cell energy
    const int rows_t = cell * (hr + 1), pitch = (cell * (hc + 1) + 15) / 16 * 16;
    const size_t px_stride = (size_t)rows_t * pitch;            // multiple of 16 => every row / batch plane stays 16-byte aligned
    grad.ensure(px_stride * n * 5 + 256);
    float* d_mag = grad.as<float>();
    uint8_t* d_bin = grad.as<uint8_t>() + px_stride * n * 4;
    const dim3 g4 = flat_grid(pitch / 4, rows_t, n), gh = flat_grid(hc, hr, n);
    if (cell == 8) {
        hipLaunchKernelGGL((fhog_grad4_k<8>), g4, dim3(256), 0, c->stream, d_img, img_stride, h, w, visible_nr, visible_nc, d_mag, d_bin, px_stride, rows_t, pitch, lut, n);
        hipLaunchKernelGGL((fhog_hist_k<8>), gh, dim3(256), 0, c->stream, d_mag, d_bin, px_stride, pitch, hist.as<float>(), hist_stride, hr, hc,
                           norm.as<float>(), norm_stride, cells_nr, cells_nc, n);
    } else {
        hipLaunchKernelGGL((fhog_grad4_k<4>), g4, dim3(256), 0, c->stream, d_img, img_stride, h, w, visible_nr, visible_nc, d_mag, d_bin, px_stride, rows_t, pitch, lut, n);
        hipLaunchKernelGGL((fhog_hist_k<4>), gh, dim3(256), 0, c->stream, d_mag, d_bin, px_stride, pitch, hist.as<float>(), hist_stride, hr, hc,
                           norm.as<float>(), norm_stride, cells_nr, cells_nc, n);
    }
    const int hog_nr = cells_nr - 2, hog_nc = cells_nc - 2;
    hipLaunchKernelGGL(fhog_feat_k, flat_grid(fw, fh, n), dim3(256), 0, c->stream, hist.as<float>(), hist_stride, hc, norm.as<float>(), norm_stride, cells_nc,
                       d_feat, feat_stride, fw, hog_nr, hog_nc, oy, ox, fh, n);
}

void fhog_debug(Ctx* c, const uint8_t* himg, int h, int w, int cell, int pad_r, int pad_c, std::vector<float>& out, int* fh, int* fw)
{
    fhog_dims(h, w, cell, pad_r, pad_c, fh, fw);
    PVF_REQUIRE(*fh > 0 && *fw > 0, "fhog: image too small");
    c->s_pyr.ensure((size_t)h * w * 3);
    HIP_CHECK(hipMemcpyAsync(c->s_pyr.p, himg, (size_t)h * w * 3, hipMemcpyHostToDevice, c->stream));
    const size_t nf = (size_t)(*fh) * (*fw) * PVF_FHOG_STRIDE;
    c->s_feat.ensure(nf * sizeof(float));
    c->feat_ring_owner = nullptr;            // s_feat is shared with the detector's feature maps: their zero border is gone after this
    fhog_device(c, c->s_pyr.as<uint8_t>(), 1, h, w, cell, pad_r, pad_c, c->s_feat.as<float>(), c->s_hist, c->s_norm, 0);
    out.resize(nf);
    HIP_CHECK(hipMemcpyAsync(out.data(), c->s_feat.p, nf * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
}

