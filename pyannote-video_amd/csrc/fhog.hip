// fhog.hip -- FHOG outside the detector: the correlation tracker's translation window (cell 1, 64 x 64: one kernel, compact record;
// reference tracking.py:203,250-251), the orientation tables, and the generic three-pass form behind the stage-access entry of the
// parity tests (cell 4 / 8, any size).  The detector's pyramid has its own fused kernel (detect.hip), the tracker's scale samples
// theirs (dsst.hip scale_fhog_k).
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

// The table of the detector's gradient waves (detect.hip: fhog_split_ml_k): the winning channel's two differences sit in the halves of one
// register as 16-bit two's complement, and the entry's place is made of bits of THAT register -- X = bx mod 512, Y = by mod 512 (no bias
// add), entry at (X & 7) | Y << 3 | (X >> 3) << 12: a 64-byte line is an 8 x 8 tile of neighbouring gradients.  (Neighbouring pixels mostly
// have small, similar gradients, so a wave's 64 look-ups touch a handful of lines; the row-major table cost ~50 L1 accesses per gather.)
const uint8_t* orientation_lut_wrapped(Ctx* c)
{
    std::lock_guard<std::recursive_mutex> lk(g_lut_mu);
    if (c->d_wrap_lut) return c->d_wrap_lut;
    orientation_lut(c);
    std::vector<uint8_t> ol((size_t)511 * 511);
    HIP_CHECK(hipMemcpy(ol.data(), c->d_orient_lut, ol.size(), hipMemcpyDeviceToHost));
    std::vector<uint8_t> lut((size_t)1 << 18, 0);
    for (int by = -255; by <= 255; ++by)
        for (int bx = -255; bx <= 255; ++bx) {
            const unsigned X = (unsigned)bx & 511u, Y = (unsigned)by & 511u;
            lut[(X & 7u) | (Y << 3) | ((X >> 3) << 12)] = ol[(size_t)(by + 255) * 511 + (bx + 255)];
        }
    HIP_CHECK(hipMalloc((void**)&c->d_wrap_lut, lut.size()));
    HIP_CHECK(hipMemcpy(c->d_wrap_lut, lut.data(), lut.size(), hipMemcpyHostToDevice));
    return c->d_wrap_lut;
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

// ---- cell size 1, 64 x 64 chip, padding 3 (the correlation tracker's translation window): every pixel is a cell whose histogram is
// ONE bin holding |g| (oracle/pvo_fhog.c fhog_cell1).  cell_features() of such a histogram has at most 6 non-zero outputs:
//   o[a] and o[18 + a % 9]  = S = (m0 + m1) + (m2 + m3),  m_k = min(|g|, nn_k) * nv_k     (the other 25 orientation planes are sums of
//                                                                                          min(0, nn_k) * nv_k = +0)
//   o[27 + k]               = m_k * tscale                (t_k = 0 + ... + ((m_k + 0) + 0) + ... : adding +0 changes nothing)
// so the kernel writes the COMPACT record (S, the four textures, the bin a) -- 21 bytes per pixel instead of 124 -- and the tracker's
// plane loop rebuilds plane i as (a == i ? S : 0).  Bit for bit what the 31-plane form held (tests: test_fhog_bit_exact at (1, 3, 64 x 64)
// expands the record again; tracker PSR / positions / filter state against the oracle).  One block per chip: squared magnitudes of the
// 64 x 64 pixels in LDS, then the features; nothing but the record goes to HBM (round 4: |g|^2 + bin planes written, read back 9 times,
// 508 KB of planes per tracker).
// Record layout: pixel q = 64 y + x sits at slot TRKF_SLOT(q) = (q & 511) * 8 + (q >> 9) of each array, so that the eight pixels
// q0 + 512 k a thread of the tracker's 512-thread kernels owns (dsst.hip FUSED_*) are 32 consecutive bytes of S / T and 8 of the bins.
__global__ void __launch_bounds__(256) fhog1_compact_k(const uint8_t* __restrict__ img, size_t img_stride, const uint8_t* __restrict__ lut,
                                                       uint8_t* __restrict__ out)
{
    __shared__ float nrm[64 * 64];
    const int b = blockIdx.x, tid = threadIdx.x;
    const uint8_t* im = img + (size_t)b * img_stride;
    int ang[16];
    // thread <-> pixels: two groups g = tid, tid + 256 of eight pixels g + 512 k (in both passes: a pixel's bin stays in its thread)
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int q = tid + 256 * (k >> 3) + 512 * (k & 7), y = q >> 6, x = q & 63;
        float v = 0.0f;
        int o = 0;
        if (y >= 1 && y < 63 && x >= 1 && x < 63)
            pixel_grad(im + (size_t)(y - 1) * 64 * 3, im + (size_t)y * 64 * 3, im + (size_t)(y + 1) * 64 * 3, x * 3, lut, &v, &o);
        nrm[q] = v;
        ang[k] = o;
    }
    __syncthreads();
    float* S = reinterpret_cast<float*>(out + (size_t)b * TRKF_BYTES);
    uint8_t* A = out + (size_t)b * TRKF_BYTES + TRKF_A;
    const float eps = 0.0001f, tscale = (float)(2 * 0.2357);
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        float s[8], t[4][8];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const int q = tid + 256 * g + 512 * kk, y = q >> 6, x = q & 63;
            s[kk] = 0.0f;
#pragma unroll
            for (int c = 0; c < 4; ++c) t[c][kk] = 0.0f;
            if (y >= 1 && y <= 62 && x >= 1 && x <= 62) {
                float n[9];
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) n[i * 3 + j] = nrm[(y - 1 + i) * 64 + (x - 1 + j)];
                const float mag = sqrtf(n[4]);
                const float z1[4] = {n[4], n[1], n[3], n[0]};
                const float z2[4] = {n[5], n[2], n[4], n[1]};
                const float z3[4] = {n[7], n[4], n[6], n[3]};
                const float z4[4] = {n[8], n[5], n[7], n[4]};
                float m[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float nn = 0.2f * sqrtf((((z1[c] + z2[c]) + z3[c]) + z4[c]) + eps);
                    const float nv = 0.1f / nn;
                    m[c] = fminf(mag, nn) * nv;
                    t[c][kk] = m[c] * tscale;
                }
                s[kk] = (m[0] + m[1]) + (m[2] + m[3]);
            }
        }
        const int slot = (tid + 256 * g) * 8;
        float4* d = reinterpret_cast<float4*>(S + slot);
        d[0] = make_float4(s[0], s[1], s[2], s[3]);
        d[1] = make_float4(s[4], s[5], s[6], s[7]);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float4* dt = reinterpret_cast<float4*>(S + (size_t)(1 + c) * TRKF_PLANE + slot);
            dt[0] = make_float4(t[c][0], t[c][1], t[c][2], t[c][3]);
            dt[1] = make_float4(t[c][4], t[c][5], t[c][6], t[c][7]);
        }
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) { lo |= (uint32_t)ang[8 * g + kk] << (8 * kk); hi |= (uint32_t)ang[8 * g + 4 + kk] << (8 * kk); }
        *reinterpret_cast<uint2*>(A + slot) = make_uint2(lo, hi);
    }
}

// stage access (pvf_debug_fhog at the tracker's shape): the compact record back into [64][64][32]
__global__ void __launch_bounds__(256) fhog1_expand_k(const uint8_t* __restrict__ rec, float* __restrict__ feat)
{
    const int q = blockIdx.x * 256 + threadIdx.x, sl = TRKF_SLOT(q);
    const float* S = reinterpret_cast<const float*>(rec);
    const int a = rec[TRKF_A + sl];
    float o[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) o[k] = 0.0f;
#pragma unroll
    for (int k = 0; k < 18; ++k) if (k == a) o[k] = S[sl];
#pragma unroll
    for (int k = 0; k < 9; ++k) if (k == (a >= 9 ? a - 9 : a)) o[18 + k] = S[sl];
#pragma unroll
    for (int c = 0; c < 4; ++c) o[27 + c] = S[(size_t)(1 + c) * TRKF_PLANE + sl];
    float4* dst = reinterpret_cast<float4*>(feat + (size_t)q * PVF_FHOG_STRIDE);
#pragma unroll
    for (int k = 0; k < 8; ++k) dst[k] = make_float4(o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]);
}

void fhog1_compact(Ctx* c, const uint8_t* d_chips, int n, uint8_t* d_rec)
{
    hipLaunchKernelGGL(fhog1_compact_k, dim3(n), dim3(256), 0, c->stream, d_chips, (size_t)64 * 64 * 3, orientation_lut(c), d_rec);
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

// generic form (any size, cell 4 or 8): the stage-access entry of the parity tests.  Production shapes have kernels of their own: the
// detector's pyramid (detect.hip), the tracker's translation chip (fhog1_compact_k above) and its scale samples (dsst.hip scale_fhog_k).
void fhog_device(Ctx* c, const uint8_t* d_img, int n, int h, int w, int cell, int pad_r, int pad_c, float* d_feat, DevBuf& hist, DevBuf& norm,
                 size_t img_stride_in)
{
    DevBuf& grad = c->s_grad;
    const uint8_t* lut = orientation_lut(c);
    int fh, fw;
    fhog_dims(h, w, cell, pad_r, pad_c, &fh, &fw);
    PVF_REQUIRE(fh > 0 && fw > 0, "fhog: image too small");
    const size_t feat_stride = (size_t)fh * fw * PVF_FHOG_STRIDE;
    const int oy = (pad_r - 1) / 2, ox = (pad_c - 1) / 2;
    const size_t img_stride = img_stride_in ? img_stride_in : (size_t)h * w * 3;
    PVF_REQUIRE(cell == 8 || cell == 4, "fhog: cell size 4 or 8 (cell 1: the tracker's 64 x 64 chip only)");
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
    if (cell == 1) {
        // the tracker's translation window is the one cell-1 shape this library computes: its own kernel, record expanded again
        PVF_REQUIRE(h == 64 && w == 64 && pad_r == 3 && pad_c == 3, "fhog: cell size 1 exists for the tracker's 64 x 64 chip with padding 3 only");
        c->s_trkfeat.ensure(TRKF_BYTES);
        fhog1_compact(c, c->s_pyr.as<uint8_t>(), 1, c->s_trkfeat.as<uint8_t>());
        hipLaunchKernelGGL(fhog1_expand_k, dim3(64 * 64 / 256), dim3(256), 0, c->stream, c->s_trkfeat.as<uint8_t>(), c->s_feat.as<float>());
    } else if (cell == 4 && h == 23 && w == 23 && pad_r == 1 && pad_c == 1) {
        fhog_scale_chips(c, c->s_pyr.as<uint8_t>(), 1, c->s_feat.as<float>());          // the tracker's scale sample: its kernel (dsst.hip)
    } else {
        fhog_device(c, c->s_pyr.as<uint8_t>(), 1, h, w, cell, pad_r, pad_c, c->s_feat.as<float>(), c->s_hist, c->s_norm, 0);
    }
    HIP_CHECK(hipGetLastError());
    out.resize(nf);
    HIP_CHECK(hipMemcpyAsync(out.data(), c->s_feat.p, nf * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
}
