// detect.hip -- K1..K4: image pyramid, FHOG, filter scoring, NMS  (replaces dlib.get_frontal_face_detector()(rgb, 1);
// reference pyannote/video/face/face.py:54,66).  All arithmetic follows the orders stated in oracle/pvo_fhog.c and
// oracle/pvo_detect.c so that boxes are bit-identical; the code itself is written for gfx950 (wave64, LDS tiles).
#include "pvf_internal.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>

// =====================================================================================================
// K1: bilinear resize (pyramid_up / pyramid_down<6>), uint8 RGB HWC, double coordinates, (v + 0.5) truncation
// =====================================================================================================
// K1, lean form: one lane = one output column, walking RS consecutive output rows.  The horizontal blend of a source row,
//   H_s = (1 - lr) * S[s][left] + lr * S[s][right],
// depends only on (s, column), and consecutive output rows share source rows, so each H_s is evaluated once and kept in
// registers (two-entry cache; the hit test depends on the row only => wave-uniform).  The value written is the same expression
// as before:  v = (1 - tb) * H_top + tb * H_bottom ; out = (uint8)(v + 0.5).
template <int RS>
__global__ void __launch_bounds__(256) resize_strip_k(const uint8_t* const* __restrict__ in_ptrs, const uint8_t* __restrict__ in_base,
                                                      size_t in_stride, int ih, int iw, uint8_t* __restrict__ out, size_t out_stride,
                                                      int oh, int ow, double x_scale, double y_scale)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int r0 = blockIdx.y * RS, b = blockIdx.z;
    if (c >= ow) return;
    const uint8_t* in = in_ptrs ? in_ptrs[b] : in_base + (size_t)b * in_stride;
    uint8_t* ob = out + (size_t)b * out_stride;
    const double x = c * x_scale;
    const int left = (int)floor(x);
    const int right = min(left + 1, iw - 1);
    const double lr = x - left, lr1 = 1 - lr;
    const int ol = left * 3, orr = right * 3;
    int s0 = -1, s1 = -1;              // cached source rows
    double h0[3], h1[3];
    const int r_end = min(r0 + RS, oh);
    for (int r = r0; r < r_end; ++r) {
        const double y = r * y_scale;
        const int top = (int)floor(y);
        const int bottom = min(top + 1, ih - 1);
        const double tb = y - top, tb1 = 1 - tb;
        // make (s0 == top, s1 == bottom)
        if (s1 == top) { s0 = s1; h0[0] = h1[0]; h0[1] = h1[1]; h0[2] = h1[2]; s1 = -1; }
        if (s0 != top) {
            const uint8_t* p = in + (size_t)top * iw * 3;
#pragma unroll
            for (int k = 0; k < 3; ++k) { const double tl = p[ol + k], tr = p[orr + k]; h0[k] = lr1 * tl + lr * tr; }
            s0 = top;
        }
        if (s1 != bottom) {
            if (bottom == top) { h1[0] = h0[0]; h1[1] = h0[1]; h1[2] = h0[2]; }
            else {
                const uint8_t* p = in + (size_t)bottom * iw * 3;
#pragma unroll
                for (int k = 0; k < 3; ++k) { const double bl = p[ol + k], br = p[orr + k]; h1[k] = lr1 * bl + lr * br; }
            }
            s1 = bottom;
        }
        uint8_t* o = ob + ((size_t)r * ow + c) * 3;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double v = tb1 * h0[k] + tb * h1[k];
            o[k] = (uint8_t)(v + 0.5);
        }
    }
}

static void launch_resize(Ctx* c, const uint8_t* const* in_ptrs, const uint8_t* in_base, size_t in_stride, int ih, int iw,
                          uint8_t* out, size_t out_stride, int oh, int ow, int batch)
{
    const double x_scale = (iw - 1) / (double)std::max(ow - 1, 1);
    const double y_scale = (ih - 1) / (double)std::max(oh - 1, 1);
    PVF_REQUIRE(x_scale <= 2.0, "resize: more than 2x horizontal decimation is not used on this path");
    constexpr int RS = 16;
    dim3 grid((ow + 255) / 256, (oh + RS - 1) / RS, batch);
    hipLaunchKernelGGL((resize_strip_k<RS>), grid, dim3(256), 0, c->stream, in_ptrs, in_base, in_stride, ih, iw, out, out_stride, oh, ow,
                       x_scale, y_scale);
}

// K1, v3 (batched detector): same arithmetic, memory access reshaped.  The v2 form issued 6 byte loads per new source row and
// 3 byte stores per pixel -- 9 vector-memory instructions per pixel made it texture-addresser bound (TA busy 75 %), and every
// source row fetch exposed a full memory latency.  Here a wave first requests EVERY source row its RS output rows need (one
// coalesced dword per lane and row: the 64 columns of a wave span < 256 source bytes for scales <= 1.25) and parks them in LDS;
// after that single wait the strip runs on LDS + VALU only.  A lane picks its two neighbouring source pixels (6 bytes) from
// three LDS dwords with a byte funnel shift; the 64 result pixels of a wave leave as 48 aligned dwords: each lane packs its 3
// bytes, lanes 0..47 collect the two packed pixels their dword straddles through the LDS crossbar (ds_bpermute) and store once.
// Needs 4-byte aligned output rows (level images of the batched path have a padded row pitch).
#define RESIZE_MAXS 22
template <int RS>
__global__ void __launch_bounds__(256) resize_rows_k(const uint8_t* const* __restrict__ in_ptrs, const uint8_t* __restrict__ in_base,
                                                     size_t in_stride, int in_rb, int ih, int iw, uint8_t* __restrict__ out,
                                                     size_t out_stride, int out_rb, int oh, int ow, double x_scale, double y_scale)
{
    constexpr int MAXS = RESIZE_MAXS;
    __shared__ uint32_t s_rows[4][MAXS][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c0 = blockIdx.x * 256 + wave * 64;               // first column of this wave's segment
    const int r0 = blockIdx.y * RS, b = blockIdx.z;
    if (c0 >= ow) return;                                      // wave-uniform
    const int c = min(c0 + lane, ow - 1);                      // lanes past the row end recompute the last column (never stored)
    const uint8_t* in = in_ptrs ? in_ptrs[b] : in_base + (size_t)b * in_stride;
    const uint8_t* in_end = in + (size_t)ih * in_rb;
    uint8_t* ob = out + (size_t)b * out_stride + (size_t)c0 * 3;
    const double x = c * x_scale;
    const int left = (int)floor(x);
    const int left0 = __builtin_amdgcn_readfirstlane(left);    // lane 0 holds column c0: the smallest source column of the wave
    const bool has_right = (left + 1 <= iw - 1);
    const double lr = x - left, lr1 = 1 - lr;
    const int r_end = min(r0 + RS, oh);
    const int s_first = (int)floor(r0 * y_scale);
    const int s_last = min((int)floor((r_end - 1) * y_scale) + 1, ih - 1);
    // stage the source rows: dword `lane` of the 256-byte window that starts at the aligned address below the wave's first pixel.
    // Branch-free so that all loads are in flight together: a dword that lies wholly past the frame is redirected to the frame's
    // last word (never used; an aligned dword cannot straddle a page, so the word holding the last byte is always readable).
    {
        typedef const __attribute__((address_space(1))) uint32_t* gptr_t;
        const uintptr_t last_word = ((uintptr_t)in_end - 1) & ~(uintptr_t)3;
        const int nrows = s_last - s_first + 1;
        uint32_t t[MAXS];
#pragma unroll
        for (int k = 0; k < MAXS; ++k) {
            t[k] = 0;
            if (k < nrows) {                                   // wave-uniform
                const uintptr_t a = (uintptr_t)(in + (size_t)(s_first + k) * in_rb + 3 * left0);
                uintptr_t q = (a & ~(uintptr_t)3) + 4 * lane;
                q = q < last_word ? q : last_word;
                t[k] = *(gptr_t)q;
            }
        }
#pragma unroll
        for (int k = 0; k < MAXS; ++k) s_rows[wave][k][lane] = t[k];
    }
    // output dword d of the segment (lane d < 48) straddles packed pixels a = 4d / 3 and a + 1, starting at byte 4d - 3a of pixel a
    const int pa_lane = (4 * lane) / 3, phase = 4 * lane - 3 * pa_lane;
    const int bp0 = 4 * min(pa_lane, 63), bp1 = 4 * min(pa_lane + 1, 63);
    const int seg_bytes = min(ow - c0, 64) * 3;
    const bool stores = (4 * lane < seg_bytes);
    int s0 = -1, s1 = -1;              // cached source rows
    double h0[3], h1[3];
    auto hblend = [&](int srow, double* hh) {
        const uint8_t* a0 = in + (size_t)srow * in_rb + 3 * left0;
        const unsigned off = ((unsigned)(uintptr_t)a0 & 3u) + 3u * (unsigned)(left - left0);     // byte offset of this lane's pixel in the window
        const uint32_t* w = &s_rows[wave][srow - s_first][off >> 2];
        const uint32_t w0 = w[0], w1 = w[1], w2 = w[2];
        const uint32_t lo = __builtin_amdgcn_alignbyte(w1, w0, off & 3u), hi = __builtin_amdgcn_alignbyte(w2, w1, off & 3u);
        const double tl0 = (double)(lo & 0xffu), tl1 = (double)((lo >> 8) & 0xffu), tl2 = (double)((lo >> 16) & 0xffu);
        double tr0 = (double)(lo >> 24), tr1 = (double)(hi & 0xffu), tr2 = (double)((hi >> 8) & 0xffu);
        if (!has_right) { tr0 = tl0; tr1 = tl1; tr2 = tl2; }
        hh[0] = lr1 * tl0 + lr * tr0;
        hh[1] = lr1 * tl1 + lr * tr1;
        hh[2] = lr1 * tl2 + lr * tr2;
    };
    for (int r = r0; r < r_end; ++r) {
        const double y = r * y_scale;
        const int top = (int)floor(y);
        const int bottom = min(top + 1, ih - 1);
        const double tb = y - top, tb1 = 1 - tb;
        if (s1 == top) { s0 = s1; h0[0] = h1[0]; h0[1] = h1[1]; h0[2] = h1[2]; s1 = -1; }
        if (s0 != top) { hblend(top, h0); s0 = top; }
        if (s1 != bottom) {
            if (bottom == top) { h1[0] = h0[0]; h1[1] = h0[1]; h1[2] = h0[2]; }
            else hblend(bottom, h1);
            s1 = bottom;
        }
        uint32_t P = 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double v = tb1 * h0[k] + tb * h1[k];
            P |= (uint32_t)(uint8_t)(v + 0.5) << (8 * k);
        }
        const uint32_t Pa = (uint32_t)__builtin_amdgcn_ds_bpermute(bp0, (int)P), Pb = (uint32_t)__builtin_amdgcn_ds_bpermute(bp1, (int)P);
        const uint32_t dw = __builtin_amdgcn_alignbyte(Pb >> 8, Pa | (Pb << 24), (unsigned)phase);
        if (stores) *reinterpret_cast<uint32_t*>(ob + (size_t)r * out_rb + 4 * lane) = dw;
    }
}

static void launch_resize_rows(Ctx* c, const uint8_t* const* in_ptrs, const uint8_t* in_base, size_t in_stride, int in_rb, int ih, int iw,
                               uint8_t* out, size_t out_stride, int out_rb, int oh, int ow, int batch)
{
    const double x_scale = (iw - 1) / (double)std::max(ow - 1, 1);
    const double y_scale = (ih - 1) / (double)std::max(oh - 1, 1);
    constexpr int RS = 16;
    PVF_REQUIRE(x_scale <= 1.25 && (RS - 1) * y_scale + 3 <= RESIZE_MAXS, "resize_rows: scale outside the pyramid's range (2x up, 6/5 down)");
    PVF_REQUIRE(out_rb % 4 == 0 && out_stride % 4 == 0 && ((uintptr_t)out & 3) == 0 && out_rb >= (ow * 3 + 3) / 4 * 4, "resize: output rows must be 4-byte aligned");
    dim3 grid((ow + 255) / 256, (oh + RS - 1) / RS, batch);
    hipLaunchKernelGGL((resize_rows_k<RS>), grid, dim3(256), 0, c->stream, in_ptrs, in_base, in_stride, in_rb, ih, iw, out, out_stride, out_rb,
                       oh, ow, x_scale, y_scale);
}

static void pyramid_up_dims(int ih, int iw, int* oh, int* ow)
{
    const double right = ((iw - 1) + 1.25) * 2.0;
    const double bottom = ((ih - 1) + 0.75) * 2.0;
    *ow = (int)std::floor(right + 0.5) + 1;
    *oh = (int)std::floor(bottom + 0.5) + 1;
}

// =====================================================================================================
// K2: FHOG.  gradient -> (orientation bin, magnitude) per pixel; histogram cells gather their 2C x 2C window in
// row-major pixel order (== the order dlib's scatter loop adds in); 4-way block normalisation -> 31 features.
// =====================================================================================================

// Orientation snap of dlib's FHOG: arg-max over 9 directions of +-dot(direction, gradient), first maximum wins.
// The gradient of a uint8 image is a pair of integers in [-255,255]^2, so the bin is a pure function of 511x511 inputs:
// it is tabulated once on the host with exactly these float operations (mul, mul, add, strict compares; no contraction),
// which makes the table bit-identical to evaluating the chain per pixel.
static const float h_dirx[9] = {1.0000f, 0.9397f, 0.7660f, 0.500f, 0.1736f, -0.1736f, -0.5000f, -0.7660f, -0.9397f};
static const float h_diry[9] = {0.0000f, 0.3420f, 0.6428f, 0.8660f, 0.9848f, 0.9848f, 0.8660f, 0.6428f, 0.3420f};

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4u __attribute__((aligned(4)));

const uint8_t* orientation_lut(Ctx* c)
{
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

// correctly rounded sqrt of a non-negative integer-valued float < 2^24: the hardware estimate (<= 1 ulp) stepped to the
// neighbour the exact residuals ask for.  Same result as sqrtf(); skips its denormal scaling and class checks.
__device__ __forceinline__ float sqrt_exact_small(float x)
{
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sm = __uint_as_float(__float_as_uint(s) - 1u), sp = __uint_as_float(__float_as_uint(s) + 1u);
    const float rm = fmaf(-sm, s, x), rp = fmaf(-sp, s, x);
    float r = (rm <= 0.0f) ? sm : s;
    r = (rp > 0.0f) ? sp : r;
    return r;
}

// colour channel with the largest |g|^2 (first wins); magnitude by arithmetic, orientation bin from the tiled table
__device__ __forceinline__ void grad_lookup(const int u[3], const int d[3], const int l[3], const int r[3],
                                            const uint8_t* __restrict__ lut_t, float* v, int* o)
{
    int bx = r[0] - l[0], by = d[0] - u[0];
    int bv = bx * bx + by * by;
    int bi = by * 512 + bx;
#pragma unroll
    for (int k = 1; k < 3; ++k) {
        const int cx = r[k] - l[k], cy = d[k] - u[k];
        const int cv = cx * cx + cy * cy;
        const int ci = cy * 512 + cx;
        if (cv > bv) { bv = cv; bi = ci; }
    }
    const unsigned P = (unsigned)(bi + 255 * 512 + 255);           // Y << 9 | X
    const unsigned off = (P & 0x3F007u) | ((P & 0x1F8u) << 3) | ((P >> 6) & 0x38u);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)lut_t, 0, 64 * 64 * 64, 0x00020000);
    *o = (int)__builtin_amdgcn_raw_buffer_load_b8(rs, off, 0, 0);
    *v = sqrt_exact_small((float)bv);
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

__device__ __forceinline__ void cell_features(const float* h, const float* n, float* o)
{
    const float eps = 0.0001f;
    const float z1[4] = {n[4], n[1], n[3], n[0]};
    const float z2[4] = {n[5], n[2], n[4], n[1]};
    const float z3[4] = {n[7], n[4], n[6], n[3]};
    const float z4[4] = {n[8], n[5], n[7], n[4]};
    float nn[4], nv[4], t[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        nn[k] = 0.2f * sqrtf((((z1[k] + z2[k]) + z3[k]) + z4[k]) + eps);
        nv[k] = 0.1f / nn[k];
    }
#pragma unroll
    for (int g = 0; g < 18; g += 3) {
        float hh[3][4];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
#pragma unroll
            for (int k = 0; k < 4; ++k) hh[j][k] = fminf(h[g + j], nn[k]) * nv[k];
            o[g + j] = (hh[j][0] + hh[j][1]) + (hh[j][2] + hh[j][3]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] = t[k] + ((hh[0][k] + hh[1][k]) + hh[2][k]);
    }
    const float tscale = (float)(2 * 0.2357);
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] = t[k] * tscale;
#pragma unroll
    for (int g = 0; g < 9; ++g) {
        const float s = h[g] + h[g + 9];
        float hh[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) hh[k] = fminf(s, nn[k]) * nv[k];
        o[18 + g] = (hh[0] + hh[1]) + (hh[2] + hh[3]);
    }
    o[27] = t[0]; o[28] = t[1]; o[29] = t[2]; o[30] = t[3];
    o[31] = 0.0f;
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
                                                    int oy, int ox, int fh, int n_img)
{
    int px, py, b;
    if (!flat_index(fw, fh, n_img, &px, &py, &b)) return;
    const int x = px - ox, y = py - oy;
    if (x < 0 || y < 0 || x >= hog_nc || y >= hog_nr) {
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
    if (cell == 1) {
        const size_t px = (size_t)h * w;
        norm.ensure(px * n * sizeof(float));
        hist.ensure(px * n);
        hipLaunchKernelGGL(fhog1_grad_k, flat_grid(w, h, n), dim3(256), 0, c->stream, d_img, img_stride, h, w, norm.as<float>(), hist.as<uint8_t>(), px, lut, n);
        hipLaunchKernelGGL(fhog1_feat_k, flat_grid(fw, fh, n), dim3(256), 0, c->stream, norm.as<float>(), hist.as<uint8_t>(), px, w, d_feat, feat_stride, fw,
                           h - 2, w - 2, oy, ox, fh, n);
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
    fhog_device(c, c->s_pyr.as<uint8_t>(), 1, h, w, cell, pad_r, pad_c, c->s_feat.as<float>(), c->s_hist, c->s_norm, 0);
    out.resize(nf);
    HIP_CHECK(hipMemcpyAsync(out.data(), c->s_feat.p, nf * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
}

// =====================================================================================================
// K3: filter scoring.  score[f](r,c) = fmaf chain over (m, n, p) -- identical order to the oracle.
// v1: VALU, one output position per lane, features staged in LDS with a 36-float cell pitch (conflict-free b128 reads),
// weights come through the scalar unit (uniform addresses).
// =====================================================================================================
struct ScoreParams { float thresh[8]; int n_filters; int level; int cap; };
struct CandRec { float score; int32_t filter, level, r, c; };

template <int NF>
__global__ void __launch_bounds__(256) score_k(const float* __restrict__ feat, size_t feat_stride, int fh, int fw,
                                               const float* __restrict__ w, ScoreParams sp, int* __restrict__ counts,
                                               CandRec* __restrict__ cands)
{
    constexpr int TR = 8, TC = 32, FR = 10, FC = 10, PITCH = 36;
    constexpr int LR = TR + FR - 1, LC = TC + FC - 1;
    extern __shared__ __attribute__((aligned(16))) float s_f[]; // [LR][LC][PITCH]
    const int b = blockIdx.z;
    const int r_base = blockIdx.y * TR, c_base = blockIdx.x * TC; // top-left of the feature window of this tile
    const float* fb = feat + (size_t)b * feat_stride;
    for (int i = threadIdx.x; i < LR * LC * 8; i += blockDim.x) {
        const int cell = i >> 3, q = i & 7;
        const int ly = cell / LC, lx = cell % LC;
        const int y = r_base + ly, x = c_base + lx;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (y < fh && x < fw) v = reinterpret_cast<const float4*>(fb + ((size_t)y * fw + x) * PVF_FHOG_STRIDE)[q];
        reinterpret_cast<float4*>(s_f + (size_t)cell * PITCH)[q] = v;
    }
    __syncthreads();
    const int ty = threadIdx.x / TC, tx = threadIdx.x % TC;
    float acc[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) acc[f] = 0.0f;
    for (int m = 0; m < FR; ++m)
        for (int n = 0; n < FC; ++n) {
            const float* fp = s_f + ((size_t)(ty + m) * LC + (tx + n)) * PITCH;
            const float* wp = w + ((size_t)m * FC + n) * PVF_FHOG_STRIDE;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float4 v = reinterpret_cast<const float4*>(fp)[q];
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    const float* wf = wp + (size_t)f * FR * FC * PVF_FHOG_STRIDE + 4 * q;
                    acc[f] = fmaf(v.x, wf[0], acc[f]);
                    acc[f] = fmaf(v.y, wf[1], acc[f]);
                    acc[f] = fmaf(v.z, wf[2], acc[f]);
                    if (q < 7) acc[f] = fmaf(v.w, wf[3], acc[f]);
                }
            }
        }
    // output position (centre convention of spatially_filter_image): r = top + FR/2, c = left + FC/2
    const int r = r_base + ty + FR / 2, cc = c_base + tx + FC / 2;
    const int r1 = fh - (FR - FR / 2 - 1), c1 = fw - (FC - FC / 2 - 1);
    if (r < r1 && cc < c1) {
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            if (acc[f] >= sp.thresh[f]) {
                const int idx = atomicAdd(&counts[b], 1);
                if (idx < sp.cap) {
                    CandRec rec;
                    rec.score = acc[f] - sp.thresh[f];
                    rec.filter = f; rec.level = sp.level; rec.r = r; rec.c = cc;
                    cands[(size_t)b * sp.cap + idx] = rec;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// K3 on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32, same k-ordered fmaf chain as the VALU form).
// N = 5 filters would use 5 of 16 MFMA columns, so three neighbouring output columns share one tile:
//   column j = 5*s + f  (s = 0..2 shift, f = filter)  ->  15 of 16 columns carry work, K grows from 10 to 12 cells per row.
//   A[i][k]  = F[r + m][c0 + 3 i + n'][p]          i = 16 window positions spaced 3 cells apart, k = (n', p), n' in [0,12)
//   B[k][j]  = W[f][m][n' - s][p]  (0 outside the 10-cell filter row): zero terms are exact no-ops in the chain, the
//   non-zero ones arrive in (m, n, p) order  =>  bit-identical to the oracle's chain.  Useful MACs / issued = 75.7 %.
// One wave = one output row x 96 columns (two 16-position tiles); its feature row segment lives in a wave-private LDS slab
// (34-float cell pitch: conflict-free ds_read_b32 for lanes 3 cells apart), B fragments stream from L2.
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) score_mfma_k(const float* __restrict__ feat, size_t feat_stride, int fh, int fw,
                                                    const float* __restrict__ Bg, ScoreParams sp, int* __restrict__ counts,
                                                    CandRec* __restrict__ cands)
{
    constexpr int FR = 10, FC = 10, NK = 12, PITCH = 34, MT = 2, WCOLS = MT * 48, SEG = WCOLS + 11;
    extern __shared__ __attribute__((aligned(16))) float s_seg[]; // [4 waves][SEG][PITCH]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.z;
    const int r_top = blockIdx.y * 4 + wave, c_base = blockIdx.x * WCOLS;
    const int r1 = fh - (FR - FR / 2 - 1), c1 = fw - (FC - FC / 2 - 1);
    if (r_top + FR / 2 >= r1) return;               // wave-uniform
    float* seg = s_seg + (size_t)wave * SEG * PITCH;
    const float* fb = feat + (size_t)b * feat_stride;
    const int i = lane & 15, kq = lane >> 4;
    f32x4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool two_tiles = (c_base + 48 + FC / 2 < c1);   // wave-uniform: ragged row ends run one 48-column tile only
    for (int m = 0; m < FR; ++m) {
        const int fr = r_top + m;
        // issue every load of this filter row up front: 96 B fragments (L2) + the feature row segment, then fill the slab
        float bv[NK * 8];
        const float* bp = Bg + (size_t)m * NK * 8 * 64 + lane;
#pragma unroll
        for (int q = 0; q < NK * 8; ++q) bv[q] = bp[q * 64];
        constexpr int NST = (SEG * 8 + 63) / 64;
        float4 sv[NST];
#pragma unroll
        for (int u = 0; u < NST; ++u) {
            const int idx = lane + 64 * u;
            const int cell = idx >> 3, q = idx & 7;
            const int x = c_base + cell;
            sv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < SEG * 8 && fr < fh && x < fw) sv[u] = reinterpret_cast<const float4*>(fb + ((size_t)fr * fw + x) * PVF_FHOG_STRIDE)[q];
        }
#pragma unroll
        for (int u = 0; u < NST; ++u) {
            const int idx = lane + 64 * u;
            if (idx < SEG * 8) {
                const int cell = idx >> 3, q = idx & 7;
                float2* d = reinterpret_cast<float2*>(seg + cell * PITCH + 4 * q);
                d[0] = make_float2(sv[u].x, sv[u].y);
                d[1] = make_float2(sv[u].z, sv[u].w);
            }
        }
        const float* a0 = seg + (3 * i) * PITCH + kq;
        // A fragments are fetched one cell column (8 k-steps x MT tiles) ahead of the MFMAs that consume them
        float an[8 * MT];
#pragma unroll
        for (int pq = 0; pq < 8; ++pq)
#pragma unroll
            for (int t = 0; t < MT; ++t) an[pq * MT + t] = a0[(t * 48) * PITCH + 4 * pq];
#pragma unroll
        for (int n = 0; n < NK; ++n) {
            float ac[8 * MT];
#pragma unroll
            for (int q = 0; q < 8 * MT; ++q) ac[q] = an[q];
            if (n + 1 < NK) {
#pragma unroll
                for (int pq = 0; pq < 8; ++pq)
#pragma unroll
                    for (int t = 0; t < MT; ++t) an[pq * MT + t] = a0[(t * 48 + n + 1) * PITCH + 4 * pq];
            }
            __builtin_amdgcn_sched_barrier(0);
            if (two_tiles) {
#pragma unroll
                for (int pq = 0; pq < 8; ++pq)
#pragma unroll
                    for (int t = 0; t < MT; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[pq * MT + t], bv[n * 8 + pq], acc[t], 0, 0, 0);
            } else {
#pragma unroll
                for (int pq = 0; pq < 8; ++pq)
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[pq * MT], bv[n * 8 + pq], acc[0], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // C/D layout of the 16x16 MFMA: column j = lane & 15, row = 4 * (lane >> 4) + reg
    const int j = lane & 15;
    if (j < 15) {
        const int s = j / 5, f = j % 5;
        const float th = sp.thresh[f];
        const int r = r_top + FR / 2;
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int pos = 4 * (lane >> 4) + reg;
                const int cc = c_base + t * 48 + 3 * pos + s + FC / 2;
                const float v = acc[t][reg];
                if (cc < c1 && v >= th) {
                    const int idx = atomicAdd(&counts[b], 1);
                    if (idx < sp.cap) {
                        CandRec rec;
                        rec.score = v - th; rec.filter = f; rec.level = sp.level; rec.r = r; rec.c = cc;
                        cands[(size_t)b * sp.cap + idx] = rec;
                    }
                }
            }
    }
}

// =====================================================================================================
// host side: level schedule, rectangle mapping, canonical sort, NMS
// =====================================================================================================
static inline long iround(double v) { return (long)std::floor(v + 0.5); }
static void rect_down6(long r[4]) { const double ratio = (6 - 1.0) / 6; for (int i = 0; i < 4; ++i) r[i] = iround((r[i] - 0.3) * ratio + 0.3); }
static void rect_up6(long r[4]) { const double ratio = 6 / (6 - 1.0); for (int i = 0; i < 4; ++i) r[i] = iround((r[i] - 0.3) * ratio + 0.3); }
static void rect_down2i(long r[4])
{
    r[0] = iround(r[0] / 2.0 - 1.25); r[1] = iround(r[1] / 2.0 - 0.75);
    r[2] = iround(r[2] / 2.0 - 1.25); r[3] = iround(r[3] / 2.0 - 0.75);
}
static int detector_levels(int h, int w, const DetectorModel& m)
{
    long r[4] = {0, 0, w - 1, h - 1};
    int levels = 0;
    do { rect_down6(r); ++levels; } while ((r[2] - r[0] + 1) >= m.min_w && (r[3] - r[1] + 1) >= m.min_h && levels < m.max_levels);
    return levels;
}
static void fhog_to_image(long px, long py, int cell, int pad_r, int pad_c, long* ox, long* oy)
{
    long x = (px + 1 - (pad_c - 1) / 2) * cell + 1;
    long y = (py + 1 - (pad_r - 1) / 2) * cell + 1;
    x += (x >= 0) ? cell / 2 : -(cell / 2);
    y += (y >= 0) ? cell / 2 : -(cell / 2);
    *ox = x; *oy = y;
}

struct LevelDims { int h, w; };
static std::vector<LevelDims> level_schedule(int h, int w, int upsample, const DetectorModel& m, std::vector<LevelDims>* ups)
{
    int ch = h, cw = w;
    for (int u = 0; u < upsample; ++u) {
        int nh, nw;
        pyramid_up_dims(ch, cw, &nh, &nw);
        ch = nh; cw = nw;
        if (ups) ups->push_back({ch, cw});
    }
    const int levels = detector_levels(ch, cw, m);
    std::vector<LevelDims> out;
    out.push_back({ch, cw});
    for (int l = 1; l < levels; ++l) { ch = (5 * ch) / 6; cw = (5 * cw) / 6; out.push_back({ch, cw}); }
    return out;
}

// table of frame pointers for the kernels of one batch.  Two tables alternate (det_slot): the pinned host copy of a batch must
// survive until its asynchronous upload has run, and det_run_many has the next batch in flight before it collects this one.
static void upload_frame_ptrs(Ctx* c, const std::vector<Frame>& frames, const uint8_t*** d_ptrs)
{
    DevBuf& d = c->s_fptr[c->det_slot & 1];
    HostBuf& h = c->h_fptr[c->det_slot & 1];
    d.ensure(frames.size() * sizeof(void*));
    h.ensure(frames.size() * sizeof(void*));
    const uint8_t** hp = h.as<const uint8_t*>();
    for (size_t i = 0; i < frames.size(); ++i) hp[i] = frames[i].d;
    HIP_CHECK(hipMemcpyAsync(d.p, hp, frames.size() * sizeof(void*), hipMemcpyHostToDevice, c->stream));
    *d_ptrs = d.as<const uint8_t*>();
}

// builds level `want_level` (or all levels when want_level < 0, calling per_level after each one); images ping-pong in s_pyr
template <class F>
static void run_pyramid(Ctx* c, const std::vector<Frame>& frames, int upsample, int want_level, F&& per_level)
{
    const DetectorModel& m = c->det;
    const int B = (int)frames.size();
    const int h = frames[0].h, w = frames[0].w;
    for (auto& f : frames) PVF_REQUIRE(f.h == h && f.w == w, "batched frames must share one size");
    std::vector<LevelDims> ups;
    std::vector<LevelDims> lv = level_schedule(h, w, upsample, m, &ups);
    auto padded = [](int hh, int ww) { return ((size_t)hh * ww * 3 + 15) & ~(size_t)15; };
    const size_t max_img = padded(lv[0].h, lv[0].w);
    c->s_pyr.ensure(2 * max_img * B + 64);
    uint8_t* buf[2] = {c->s_pyr.as<uint8_t>(), c->s_pyr.as<uint8_t>() + max_img * B};
    const uint8_t** d_ptrs = nullptr;
    upload_frame_ptrs(c, frames, &d_ptrs);
    int cur = 0;
    const uint8_t* cur_img = nullptr; // null => original frames via pointers
    int ch = h, cw = w;
    {
        ProfScope ps(c, "pyramid");
        for (size_t u = 0; u < ups.size(); ++u) {
            launch_resize(c, cur_img ? nullptr : d_ptrs, cur_img, padded(ch, cw), ch, cw, buf[cur], padded(ups[u].h, ups[u].w),
                          ups[u].h, ups[u].w, B);
            cur_img = buf[cur]; cur ^= 1; ch = ups[u].h; cw = ups[u].w;
        }
    }
    if (!cur_img) {
        // no upsampling: copy frames into the ping-pong buffer so that every level has the same batched layout
        for (int b = 0; b < B; ++b)
            HIP_CHECK(hipMemcpyAsync(buf[cur] + (size_t)b * padded(h, w), frames[b].d, (size_t)h * w * 3, hipMemcpyDeviceToDevice, c->stream));
        cur_img = buf[cur]; cur ^= 1;
    }
    for (int l = 0; l < (int)lv.size(); ++l) {
        if (l > 0) {
            ProfScope ps(c, "pyramid");
            launch_resize(c, nullptr, cur_img, padded(ch, cw), ch, cw, buf[cur], padded(lv[l].h, lv[l].w), lv[l].h, lv[l].w, B);
            cur_img = buf[cur]; cur ^= 1; ch = lv[l].h; cw = lv[l].w;
        }
        if (want_level < 0 || want_level == l) per_level(l, cur_img, ch, cw);
        if (want_level == l) break;
    }
}

// =====================================================================================================
// All pyramid levels in one launch per stage.  HBM is large (288 GB): the whole pyramid of a batch stays resident
// (81 MB of images + 136 MB of gradient planes + 33 MB of histograms + 58 MB of features per 1080p frame), so after the
// (sequentially dependent) resize chain, FHOG pass 1, pass 2, the feature pass and the scoring each run as ONE grid that
// covers every level -- 4 launches instead of 80 per batch, and the small levels fill the CUs the big ones leave idle.
// A block finds its level with a short scan of block-start offsets passed by value (kernarg / scalar cache).
// =====================================================================================================
#define ML_MAX 32
struct LvDesc {
    int h, w, rb;                               // level image, row pitch in bytes (multiple of 64)
    int cells_nr, cells_nc, hr, hc, visible_nr, visible_nc;
    int rows_t, pitch;                          // gradient planes
    int fh, fw, hog_nr, hog_nc;                 // features
    int grad_bx, hist_bx, feat_bx, score_bx, score_by;
    int valid_score;
    long long img_off, img_stride;              // bytes
    long long px_off, px_stride;                // elements
    long long hist_off, hist_stride, norm_off, norm_stride, feat_off, feat_stride;   // floats
};
struct MlStarts { int nl; int b0[ML_MAX + 1]; };

// Logical block id of the multi-level kernels.  A contiguous-range-per-XCD order (so that vertically adjacent tiles meet in one
// L2) was measured 5-8 % SLOWER for the histogram / feature / scoring kernels (the ranges differ in cost per block, and the
// re-reads it saves are served by the MALL anyway), so blocks keep the hardware's round-robin order.
__device__ __forceinline__ int ml_block(const MlStarts& st) { return (int)blockIdx.x; }
static inline int ml_grid(int total) { return total; }

__device__ __forceinline__ int ml_level(const MlStarts& st, int g)
{
    int l = 0;
    while (l + 1 < st.nl && g >= st.b0[l + 1]) ++l;
    return l;
}

// v3 of the gradient pass: one lane owns 4 pixel columns x GR rows.  The 20-byte neighbourhood of a row is loaded once
// (5 unaligned dwords) and serves as the "down" row of the row above, the centre row, and the "up" row of the row below,
// so an interior row costs 5 loads instead of 11 and the level lookup / index arithmetic is paid once per GR rows.
#define GRAD_ROWS 8
// (magnitude, bin) of a pixel in ONE dword.  A magnitude is 0 or sqrt of an integer in [1, 130050], i.e. in [1, 361): scaled by
// 2^-126 (exact, still a normal number) its exponent field is 1..9, so the sign bit and the top four exponent bits are zero and
// hold the 5-bit orientation bin.  The histogram pass undoes it with one AND and one exact multiply.  4 bytes per pixel instead
// of 5 and one store / one load stream instead of two.
__device__ __forceinline__ uint32_t pack_mag_bin(float mag, int bin) { return __float_as_uint(mag * 0x1p-126f) | ((uint32_t)bin << 27); }
__device__ __forceinline__ float packed_mag(uint32_t w) { return __uint_as_float(w & 0x07FFFFFFu) * 0x1p+126f; }
__device__ __forceinline__ int packed_bin(uint32_t w) { return (int)(w >> 27); }

__global__ void __launch_bounds__(256) fhog_grad4r_ml_k(MlStarts st, const LvDesc* __restrict__ lv, int B, const uint8_t* __restrict__ img_base,
                                                        uint32_t* __restrict__ px_base,
                                                        const uint8_t* __restrict__ lut2)
{
    constexpr int C = 8, GR = GRAD_ROWS;
    const int g = ml_block(st);
    if (g >= st.b0[st.nl]) return;
    const int l = ml_level(st, g);
    const LvDesc d = lv[l];
    const int local = g - st.b0[l];
    const int nyb = (d.rows_t + GR - 1) / GR;
    const int qb = local % d.grad_bx;
    const int yb = (local / d.grad_bx) % nyb;
    const int b = local / (d.grad_bx * nyb);
    const int xx = 4 * (qb * 256 + threadIdx.x);
    if (xx >= d.pitch) return;
    const int x0 = xx - 3 * C / 2;
    const int yy0 = yb * GR;
    const int rb = d.rb;
    const uint8_t* im = img_base + d.img_off + (size_t)b * d.img_stride;
    uint32_t* pxo = px_base + d.px_off + (size_t)b * d.px_stride + xx;
    const bool col_any = (x0 + 3 >= 1 && x0 < d.visible_nc);
    const bool col_fast = (x0 >= 1 && x0 + 4 <= d.visible_nc && x0 + 6 <= d.w);
    if (col_fast) {
        // all GR + 2 image rows of this lane are requested up front (one latency per block, not one per row); a row outside
        // the image is never used.  The bin look-ups of row r are in flight while row r + 1 is computed; row r is stored then.
        uint32_t w[GR + 2][5];
        const uint8_t* pc = im + 3 * x0 - 3;
        const int yfirst = yy0 - 3 * C / 2;
#pragma unroll
        for (int k = 0; k < GR + 2; ++k) {
            const int y = yfirst - 1 + k;
            if (y >= 0 && y < d.h) {
                const uint8_t* p = pc + (size_t)y * rb;
                // aligned dwords + a byte funnel shift (the misalignment is the same for every lane of a row): unaligned
                // 16-byte loads measured 20 % slower for the whole kernel
                const unsigned sh = (unsigned)(uintptr_t)p & 3u;
                const uint8_t* pa = p - sh;
                const u32x4 q = *reinterpret_cast<const u32x4u*>(pa);
                const uint32_t a0 = q.x, a1 = q.y, a2 = q.z, a3 = q.w;
                const uint32_t a4 = *reinterpret_cast<const uint32_t*>(pa + 16), a5 = *reinterpret_cast<const uint32_t*>(pa + 20);
                w[k][0] = __builtin_amdgcn_alignbyte(a1, a0, sh); w[k][1] = __builtin_amdgcn_alignbyte(a2, a1, sh);
                w[k][2] = __builtin_amdgcn_alignbyte(a3, a2, sh); w[k][3] = __builtin_amdgcn_alignbyte(a4, a3, sh);
                w[k][4] = __builtin_amdgcn_alignbyte(a5, a4, sh);
            } else {
#pragma unroll
                for (int j = 0; j < 5; ++j) w[k][j] = 0;
            }
        }
#define BYTE_OF(w, i) (int)(((w)[(i) >> 2] >> (8 * ((i) & 3))) & 0xffu)
        float pv[4] = {0.f, 0.f, 0.f, 0.f};
        int po[4] = {0, 0, 0, 0};
#pragma unroll
        for (int r = 0; r <= GR; ++r) {
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            int o[4] = {0, 0, 0, 0};
            if (r < GR) {
                const int y = yfirst + r;
                if (yy0 + r < d.rows_t && y >= 1 && y < d.visible_nr) {            // block-uniform
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        int u[3], dd[3], ll[3], rr[3];
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            u[k] = BYTE_OF(w[r], 3 * p + 3 + k); dd[k] = BYTE_OF(w[r + 2], 3 * p + 3 + k);
                            ll[k] = BYTE_OF(w[r + 1], 3 * p + k); rr[k] = BYTE_OF(w[r + 1], 3 * p + 6 + k);
                        }
                        grad_lookup(u, dd, ll, rr, lut2, &v[p], &o[p]);
                    }
                }
            }
            if (r >= 1 && yy0 + r - 1 < d.rows_t) {
                const size_t idx = (size_t)(yy0 + r - 1) * d.pitch;
                *reinterpret_cast<uint4*>(pxo + idx) = make_uint4(pack_mag_bin(pv[0], po[0]), pack_mag_bin(pv[1], po[1]), pack_mag_bin(pv[2], po[2]),
                                                                  pack_mag_bin(pv[3], po[3]));
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) { pv[p] = v[p]; po[p] = o[p]; }
        }
#undef BYTE_OF
        return;
    }
    for (int r = 0; r < GR; ++r) {
        const int yy = yy0 + r, y = yy - 3 * C / 2;
        if (yy >= d.rows_t) break;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        int o[4] = {0, 0, 0, 0};
        if (col_any && y >= 1 && y < d.visible_nr) {
            const uint8_t* rc = im + (size_t)y * rb;
            const uint8_t* ru = rc - rb;
            const uint8_t* rd = rc + rb;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int x = x0 + p;
                if (x >= 1 && x < d.visible_nc) {
                    int u[3], dd[3], ll[3], rr[3];
#pragma unroll
                    for (int k = 0; k < 3; ++k) { u[k] = ru[3 * x + k]; dd[k] = rd[3 * x + k]; ll[k] = rc[3 * x - 3 + k]; rr[k] = rc[3 * x + 3 + k]; }
                    grad_lookup(u, dd, ll, rr, lut2, &v[p], &o[p]);
                }
            }
        }
        const size_t idx = (size_t)yy * d.pitch;
        *reinterpret_cast<uint4*>(pxo + idx) = make_uint4(pack_mag_bin(v[0], o[0]), pack_mag_bin(v[1], o[1]), pack_mag_bin(v[2], o[2]), pack_mag_bin(v[3], o[3]));
    }
}

// v3 of the histogram pass: one lane owns HK vertically consecutive cells of one cell column and walks the 8 (HK + 1) pixel rows
// they cover ONCE, top to bottom.  A row in band g feeds the lower half of cell g - 1 and the upper half of cell g, so every cell
// still receives its votes in row-major order of its own 16 x 16 window (the order dlib's scatter loop produces), while the
// (magnitude, bin) planes are read (HK + 1) / HK times instead of twice.  Two accumulator sets in LDS alternate between cells.
#define HIST_CELLS 4
__global__ void __launch_bounds__(256) fhog_hist4_ml_k(MlStarts st, const LvDesc* __restrict__ lv, int B, const uint32_t* __restrict__ px_base,
                                                       float* __restrict__ hist_base,
                                                       float* __restrict__ norm_base)
{
    constexpr int C = 8, HK = HIST_CELLS, NV = 2 * C / 4;
    __shared__ float acc[2][18][256];
    const int g0 = ml_block(st);
    if (g0 >= st.b0[st.nl]) return;
    const int l = ml_level(st, g0);
    const LvDesc d = lv[l];
    const int local = g0 - st.b0[l];
    const int nyb = (d.hr + HK - 1) / HK;
    const int xb = local % d.hist_bx;
    const int yb = (local / d.hist_bx) % nyb;
    const int b = local / (d.hist_bx * nyb);
    const int hx = xb * 256 + threadIdx.x;
    const int tid = threadIdx.x;
#pragma unroll
    for (int o = 0; o < 18; ++o) { acc[0][o][tid] = 0.0f; acc[1][o][tid] = 0.0f; }
    if (hx >= d.hc) return;
    const int hy0 = yb * HK;
    const int ncell = (d.hr - hy0 < HK) ? d.hr - hy0 : HK;       // cells of this lane that exist (block-uniform)
    const uint32_t* pxi = px_base + d.px_off + (size_t)b * d.px_stride + (size_t)C * hx + (size_t)(C * hy0) * d.pitch;
    uint4 pw[2][NV];
    auto load_row = [&](int r, uint4* dw) {
        const size_t row = (size_t)r * d.pitch;
#pragma unroll
        for (int q = 0; q < NV; ++q) dw[q] = *reinterpret_cast<const uint4*>(pxi + row + 4 * q);
    };
    auto finish = [&](int j) {                                     // cell j of this lane is complete: write it out, clear its set
        const int hy = hy0 + j, set = j & 1;
        float* h = hist_base + d.hist_off + (size_t)b * d.hist_stride + ((size_t)hy * d.hc + hx) * 18;
        float e = 0.0f;
#pragma unroll
        for (int o = 0; o < 9; ++o) {
            const float a0 = acc[set][o][tid], a1 = acc[set][o + 9][tid];
            h[o] = a0; h[o + 9] = a1;
            const float s2 = a0 + a1;
            e = e + s2 * s2;
            acc[set][o][tid] = 0.0f; acc[set][o + 9][tid] = 0.0f;
        }
        if (hy >= 1 && hy <= d.cells_nr && hx >= 1 && hx <= d.cells_nc)
            norm_base[d.norm_off + (size_t)b * d.norm_stride + (size_t)(hy - 1) * d.cells_nc + (hx - 1)] = e;
    };
    const int nrows = C * (ncell + 1);
    load_row(0, pw[0]);
    for (int g = 0; g <= ncell; ++g) {
        const bool lower = (g >= 1);            // rows of this band are the lower half of cell g - 1
        const bool upper = (g < ncell);         // ... and the upper half of cell g
        float* accl = &acc[(g + 1) & 1][0][tid];
        float* accu = &acc[g & 1][0][tid];
#pragma unroll
        for (int i = 0; i < C; ++i) {
            const int cur = i & 1;
            const int r = C * g + i;
            if (r + 1 < nrows) load_row(r + 1, pw[cur ^ 1]);
            const float fy = ((float)i + 0.5f) / (float)C;
            float v[2 * C];
            int ob[2 * C];
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                const uint32_t w4[4] = {pw[cur][q].x, pw[cur][q].y, pw[cur][q].z, pw[cur][q].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[4 * q + e] = packed_mag(w4[e]); ob[4 * q + e] = packed_bin(w4[e]); }
            }
            if (lower) {
#pragma unroll
                for (int wx = 0; wx < 2 * C; ++wx) {
                    const int j = wx % C;
                    const float fx = ((float)j + 0.5f) / (float)C;
                    const float wxv = (wx < C) ? fx : 1.0f - fx;
                    const int o = ob[wx];
                    accl[o * 256] = accl[o * 256] + ((1.0f - fy) * wxv) * v[wx];
                }
            }
            if (upper) {
#pragma unroll
                for (int wx = 0; wx < 2 * C; ++wx) {
                    const int j = wx % C;
                    const float fx = ((float)j + 0.5f) / (float)C;
                    const float wxv = (wx < C) ? fx : 1.0f - fx;
                    const int o = ob[wx];
                    accu[o * 256] = accu[o * 256] + (fy * wxv) * v[wx];
                }
            }
        }
        if (lower) finish(g - 1);
    }
}

__global__ void __launch_bounds__(256) fhog_feat_ml_k(MlStarts st, const LvDesc* __restrict__ lv, int B, const float* __restrict__ hist_base,
                                                      const float* __restrict__ norm_base, float* __restrict__ feat_base, int oy, int ox)
{
    const int g = ml_block(st);
    if (g >= st.b0[st.nl]) return;
    const int l = ml_level(st, g);
    const LvDesc d = lv[l];
    const int local = g - st.b0[l];
    const int xb = local % d.feat_bx;
    const int py = (local / d.feat_bx) % d.fh;
    const int b = local / (d.feat_bx * d.fh);
    const int px = xb * 256 + threadIdx.x;
    if (px >= d.fw) return;
    float4* dst = reinterpret_cast<float4*>(feat_base + d.feat_off + (size_t)b * d.feat_stride + ((size_t)py * d.fw + px) * PVF_FHOG_STRIDE);
    const int x = px - ox, y = py - oy;
    if (x < 0 || y < 0 || x >= d.hog_nc || y >= d.hog_nr) {
#pragma unroll
        for (int k = 0; k < 8; ++k) dst[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    float n[9], h[18], o[32];
    const float* nb = norm_base + d.norm_off + (size_t)b * d.norm_stride;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) n[i * 3 + j] = nb[(size_t)(y + i) * d.cells_nc + (x + j)];
    const float* hp = hist_base + d.hist_off + (size_t)b * d.hist_stride + ((size_t)(y + 2) * d.hc + (x + 2)) * 18;
#pragma unroll
    for (int k = 0; k < 18; ++k) h[k] = hp[k];
    cell_features(h, n, o);
#pragma unroll
    for (int k = 0; k < 8; ++k) dst[k] = make_float4(o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]);
}

// ---------------------------------------------------------------------------------------------------
// K3 v3: one wave owns R consecutive output rows x 96 columns and walks the R + 9 feature rows it needs ONCE.
// Staged feature row t feeds output row j through filter row m = t - j, so every A fragment read from the slab is used for
// up to R MFMAs and the slab is filled (R+9)/R times per output row instead of 10 times.  B fragments (packed four k-steps
// per lane: [m][n][2][64 lanes][4]) and A fragments are fetched one cell column ahead of the MFMAs that consume them; the
// next feature row is loaded into registers while the current one is multiplied.  Each accumulator still receives its
// terms in (m, n, p) order  =>  bit-identical to the oracle's chain.
template <int R>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
score_mfma_rows_ml_k(MlStarts st, const LvDesc* __restrict__ lv, int B, const float* __restrict__ feat_base,
                     const float4* __restrict__ Bg4, ScoreParams sp, int* __restrict__ counts, CandRec* __restrict__ cands)
{
    constexpr int FR = 10, FC = 10, NK = 12, PITCH = 34, MT = 2, WCOLS = MT * 48, SEG = WCOLS + 11, NT = FR + R - 1;
    constexpr int NST = (SEG * 8 + 63) / 64;
    constexpr int RSRC_FLAGS = 0x00020000;          // raw buffer, 32-bit data format (out-of-range lanes read 0)
    extern __shared__ __attribute__((aligned(16))) float s_seg[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = ml_block(st);
    if (g >= st.b0[st.nl]) return;
    const int l = ml_level(st, g);
    const LvDesc d = lv[l];
    const int local = g - st.b0[l];
    const int bx = local % d.score_bx;
    const int by = (local / d.score_bx) % d.score_by;
    const int b = __builtin_amdgcn_readfirstlane(local / (d.score_bx * d.score_by));
    const int fh = d.fh, fw = d.fw;
    const int r_top = (by * 4 + wave) * R, c_base = bx * WCOLS;
    const int r1 = fh - (FR - FR / 2 - 1), c1 = fw - (FC - FC / 2 - 1);
    if (r_top + FR / 2 >= r1) return;               // wave-uniform
    float* seg = s_seg + (size_t)wave * SEG * PITCH;
    const float* fb = feat_base + d.feat_off + (size_t)b * d.feat_stride + (size_t)c_base * PVF_FHOG_STRIDE;
    const int seg_cells = (fw - c_base < SEG) ? fw - c_base : SEG;
    const int i = lane & 15, kq = lane >> 4;
    const int lane16 = lane * 16;
    f32x4 acc[R][MT];
#pragma unroll
    for (int j = 0; j < R; ++j)
#pragma unroll
        for (int t = 0; t < MT; ++t) acc[j][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool two_tiles = (c_base + 48 + FC / 2 < c1);
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc((void*)Bg4, 0, FR * NK * 2 * 64 * 16, RSRC_FLAGS);
    u32x4 sv[NST];
    // the segment of one feature row is one linear run of seg_cells * 128 bytes: lane + 64 u -> 16 bytes at 16 * (lane + 64 u)
    auto load_row = [&](int fr) {
        const int bytes = (fr < fh) ? seg_cells * PVF_FHOG_STRIDE * 4 : 0;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(fb + (size_t)fr * fw * PVF_FHOG_STRIDE), 0, bytes, RSRC_FLAGS);
#pragma unroll
        for (int u = 0; u < NST; ++u) sv[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane16, u * 1024, 0);
    };
    auto fill_slab = [&]() {
#pragma unroll
        for (int u = 0; u < NST; ++u) {
            const int idx = lane + 64 * u;
            if (idx < SEG * 8) {
                const int cell = idx >> 3, q = idx & 7;
                uint2* dd = reinterpret_cast<uint2*>(seg + cell * PITCH + 4 * q);
                dd[0] = make_uint2(sv[u].x, sv[u].y);
                dd[1] = make_uint2(sv[u].z, sv[u].w);
            }
        }
    };
    load_row(r_top);
    fill_slab();
    const float* a0 = seg + (3 * i) * PITCH + kq;
    for (int t = 0; t < NT; ++t) {
        if (t + 1 < NT) load_row(r_top + t + 1);
        // filter row of output row j at this step (clamped: the fragments of an inactive row are loaded but never used)
        int bo[R];
        bool on[R];
        bool all_on = two_tiles;
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int m = t - j;
            on[j] = (m >= 0 && m < FR);
            all_on = all_on && on[j];
            const int mc = m < 0 ? 0 : (m >= FR ? FR - 1 : m);
            bo[j] = mc * NK * 2048;
        }
        u32x4 bn[R][2];
        float an[8 * MT];
#pragma unroll
        for (int j = 0; j < R; ++j) {
            bn[j][0] = __builtin_amdgcn_raw_buffer_load_b128(brs, lane16, bo[j], 0);
            bn[j][1] = __builtin_amdgcn_raw_buffer_load_b128(brs, lane16, bo[j] + 1024, 0);
        }
#pragma unroll
        for (int pq = 0; pq < 8; ++pq)
#pragma unroll
            for (int tt = 0; tt < MT; ++tt) an[pq * MT + tt] = a0[(tt * 48) * PITCH + 4 * pq];
#pragma unroll
        for (int n = 0; n < NK; ++n) {
            float ac[8 * MT];
            float bc[R][8];
#pragma unroll
            for (int q = 0; q < 8 * MT; ++q) ac[q] = an[q];
#pragma unroll
            for (int j = 0; j < R; ++j)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const uint32_t b0 = bn[j][h].x, b1 = bn[j][h].y, b2 = bn[j][h].z, b3 = bn[j][h].w;   // scalars first: bit_cast of a vector element lvalue reads element 0
                    bc[j][4 * h] = __uint_as_float(b0); bc[j][4 * h + 1] = __uint_as_float(b1);
                    bc[j][4 * h + 2] = __uint_as_float(b2); bc[j][4 * h + 3] = __uint_as_float(b3);
                }
            if (n + 1 < NK) {
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    bn[j][0] = __builtin_amdgcn_raw_buffer_load_b128(brs, lane16, bo[j] + (n + 1) * 2048, 0);
                    bn[j][1] = __builtin_amdgcn_raw_buffer_load_b128(brs, lane16, bo[j] + (n + 1) * 2048 + 1024, 0);
                }
#pragma unroll
                for (int pq = 0; pq < 8; ++pq)
#pragma unroll
                    for (int tt = 0; tt < MT; ++tt) an[pq * MT + tt] = a0[(tt * 48 + n + 1) * PITCH + 4 * pq];
            }
            __builtin_amdgcn_sched_barrier(0);
            if (all_on) {
                // steady state: 2R independent accumulator chains interleaved
#pragma unroll
                for (int pq = 0; pq < 8; ++pq)
#pragma unroll
                    for (int j = 0; j < R; ++j)
#pragma unroll
                        for (int tt = 0; tt < MT; ++tt)
                            acc[j][tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[pq * MT + tt], bc[j][pq], acc[j][tt], 0, 0, 0);
            } else {
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    if (!on[j]) continue;
                    if (two_tiles) {
#pragma unroll
                        for (int pq = 0; pq < 8; ++pq)
#pragma unroll
                            for (int tt = 0; tt < MT; ++tt)
                                acc[j][tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[pq * MT + tt], bc[j][pq], acc[j][tt], 0, 0, 0);
                    } else {
#pragma unroll
                        for (int pq = 0; pq < 8; ++pq)
                            acc[j][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[pq * MT], bc[j][pq], acc[j][0], 0, 0, 0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (t + 1 < NT) fill_slab();
    }
    const int jc = lane & 15;
    if (jc < 15) {
        const int s = jc / 5, f = jc % 5;
        const float th = sp.thresh[f];
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int r = r_top + j + FR / 2;
            if (r >= r1) continue;
#pragma unroll
            for (int tt = 0; tt < MT; ++tt)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int pos = 4 * (lane >> 4) + reg;
                    const int cc = c_base + tt * 48 + 3 * pos + s + FC / 2;
                    const float v = acc[j][tt][reg];
                    if (cc < c1 && v >= th) {
                        const int idx = atomicAdd(&counts[b], 1);
                        if (idx < sp.cap) {
                            CandRec rec;
                            rec.score = v - th; rec.filter = f; rec.level = l; rec.r = r; rec.c = cc;
                            cands[(size_t)b * sp.cap + idx] = rec;
                        }
                    }
                }
        }
    }
}

// output rows per wave of the scoring kernel (PVF_SCORE_ROWS = 2 or 4)
static int score_rows_per_wave()
{
    static int r = -1;
    if (r < 0) {
        const char* e = getenv("PVF_SCORE_ROWS");
        r = e ? atoi(e) : 4;
        if (r != 2 && r != 4) r = 4;
    }
    return r;
}

struct MlPlan {
    int h = 0, w = 0, upsample = -1, B = 0;
    std::vector<LvDesc> lv;
    std::vector<LevelDims> ups;
    MlStarts grad, hist, feat, score;
    int grad_blocks = 0, hist_blocks = 0, feat_blocks = 0, score_blocks = 0;
    size_t img_bytes = 0, px_elems = 0, hist_floats = 0, norm_floats = 0, feat_floats = 0, up_bytes = 0;
    LvDesc* d_lv = nullptr;
};

static MlPlan* ml_plan(Ctx* c, int h, int w, int upsample, int B)
{
    static std::map<std::pair<Ctx*, std::vector<int>>, MlPlan> cache;
    auto key = std::make_pair(c, std::vector<int>{h, w, upsample, B});
    auto it = cache.find(key);
    if (it != cache.end()) return &it->second;
    const DetectorModel& m = c->det;
    MlPlan p;
    p.h = h; p.w = w; p.upsample = upsample; p.B = B;
    std::vector<LevelDims> dims = level_schedule(h, w, upsample, m, &p.ups);
    PVF_REQUIRE((int)dims.size() <= ML_MAX, "too many pyramid levels");
    auto al = [](size_t v, size_t a) { return (v + a - 1) / a * a; };
    for (size_t u = 0; u + 1 < p.ups.size(); ++u) p.up_bytes = std::max(p.up_bytes, (size_t)p.ups[u].h * al((size_t)p.ups[u].w * 3, 64) * B);
    p.grad.nl = p.hist.nl = p.feat.nl = p.score.nl = (int)dims.size();
    for (size_t l = 0; l < dims.size(); ++l) {
        LvDesc d;
        memset(&d, 0, sizeof d);
        d.h = dims[l].h; d.w = dims[l].w;
        d.cells_nr = (int)((double)d.h / 8.0 + 0.5); d.cells_nc = (int)((double)d.w / 8.0 + 0.5);
        d.hr = d.cells_nr + 2; d.hc = d.cells_nc + 2;
        d.visible_nr = std::min(d.cells_nr * 8, d.h) - 1; d.visible_nc = std::min(d.cells_nc * 8, d.w) - 1;
        d.rows_t = 8 * (d.hr + 1); d.pitch = (8 * (d.hc + 1) + 15) / 16 * 16;
        d.hog_nr = d.cells_nr - 2; d.hog_nc = d.cells_nc - 2;
        const bool feat_ok = d.hog_nr > 0 && d.hog_nc > 0;
        d.fh = feat_ok ? d.hog_nr + m.frows - 1 : 0; d.fw = feat_ok ? d.hog_nc + m.fcols - 1 : 0;
        d.valid_score = (d.fh >= m.frows && d.fw >= m.fcols) ? 1 : 0;
        d.rb = (int)al((size_t)d.w * 3, 64);
        d.img_off = (long long)p.img_bytes; d.img_stride = (long long)d.h * d.rb;
        p.img_bytes += (size_t)d.img_stride * B;
        d.px_off = (long long)p.px_elems; d.px_stride = (long long)d.rows_t * d.pitch;
        p.px_elems += (size_t)d.px_stride * B;
        d.hist_off = (long long)p.hist_floats; d.hist_stride = (long long)d.hr * d.hc * 18;
        p.hist_floats += al((size_t)d.hist_stride * B, 4);
        d.norm_off = (long long)p.norm_floats; d.norm_stride = (long long)d.cells_nr * d.cells_nc;
        p.norm_floats += al((size_t)d.norm_stride * B, 4);
        d.feat_off = (long long)p.feat_floats; d.feat_stride = (long long)d.fh * d.fw * PVF_FHOG_STRIDE;
        p.feat_floats += (size_t)d.feat_stride * B;
        d.grad_bx = (d.pitch / 4 + 255) / 256; d.hist_bx = (d.hc + 255) / 256; d.feat_bx = std::max((d.fw + 255) / 256, 0);
        const int out_r = d.fh - 9, out_c = d.fw - 9;
        const int rows_per_block = 4 * score_rows_per_wave();
        d.score_bx = d.valid_score ? (out_c + 95) / 96 : 0; d.score_by = d.valid_score ? (out_r + rows_per_block - 1) / rows_per_block : 0;
        p.grad.b0[l] = p.grad_blocks; p.grad_blocks += d.grad_bx * ((d.rows_t + GRAD_ROWS - 1) / GRAD_ROWS) * B;
        p.hist.b0[l] = p.hist_blocks; p.hist_blocks += d.hist_bx * ((d.hr + HIST_CELLS - 1) / HIST_CELLS) * B;
        p.feat.b0[l] = p.feat_blocks; p.feat_blocks += d.feat_bx * d.fh * B;
        p.score.b0[l] = p.score_blocks; p.score_blocks += d.score_bx * d.score_by * B;
        p.lv.push_back(d);
    }
    const int nl = (int)dims.size();
    p.grad.b0[nl] = p.grad_blocks; p.hist.b0[nl] = p.hist_blocks; p.feat.b0[nl] = p.feat_blocks; p.score.b0[nl] = p.score_blocks;
    HIP_CHECK(hipMalloc((void**)&p.d_lv, sizeof(LvDesc) * nl));
    HIP_CHECK(hipMemcpy(p.d_lv, p.lv.data(), sizeof(LvDesc) * nl, hipMemcpyHostToDevice));
    auto res = cache.emplace(key, std::move(p));
    return &res.first->second;
}

// builds the whole pyramid of the batch into s_pyr (level images at plan->lv[l].img_off); returns the plan
static MlPlan* ml_build_pyramid(Ctx* c, const std::vector<Frame>& frames, int upsample)
{
    const int B = (int)frames.size();
    const int h = frames[0].h, w = frames[0].w;
    for (auto& f : frames) PVF_REQUIRE(f.h == h && f.w == w, "batched frames must share one size");
    MlPlan* p = ml_plan(c, h, w, upsample, B);
    c->s_pyr.ensure(p->img_bytes + p->up_bytes + 256);
    uint8_t* base = c->s_pyr.as<uint8_t>();
    uint8_t* up_tmp = base + ((p->img_bytes + 63) / 64) * 64;
    const uint8_t** d_ptrs = nullptr;
    upload_frame_ptrs(c, frames, &d_ptrs);
    ProfScope ps(c, "pyramid");
    auto al = [](size_t v, size_t a) { return (v + a - 1) / a * a; };
    const uint8_t* cur = nullptr;
    int ch = h, cw = w, crb = w * 3;
    size_t cstride = 0;
    for (size_t u = 0; u < p->ups.size(); ++u) {
        const bool last = (u + 1 == p->ups.size());
        uint8_t* dst = last ? base + p->lv[0].img_off : up_tmp;
        const int drb = (int)al((size_t)p->ups[u].w * 3, 64);
        const size_t dstride = (size_t)p->ups[u].h * drb;
        launch_resize_rows(c, cur ? nullptr : d_ptrs, cur, cstride, crb, ch, cw, dst, dstride, drb, p->ups[u].h, p->ups[u].w, B);
        cur = dst; cstride = dstride; crb = drb; ch = p->ups[u].h; cw = p->ups[u].w;
    }
    if (!cur) {
        for (int b = 0; b < B; ++b)
            HIP_CHECK(hipMemcpy2DAsync(base + p->lv[0].img_off + (size_t)b * p->lv[0].img_stride, (size_t)p->lv[0].rb, frames[b].d, (size_t)w * 3,
                                       (size_t)w * 3, (size_t)h, hipMemcpyDeviceToDevice, c->stream));
    }
    for (size_t l = 1; l < p->lv.size(); ++l)
        launch_resize_rows(c, nullptr, base + p->lv[l - 1].img_off, (size_t)p->lv[l - 1].img_stride, p->lv[l - 1].rb, p->lv[l - 1].h, p->lv[l - 1].w,
                           base + p->lv[l].img_off, (size_t)p->lv[l].img_stride, p->lv[l].rb, p->lv[l].h, p->lv[l].w, B);
    return p;
}

// pyramid + FHOG features of every level of the batch (s_feat at plan->lv[l].feat_off); returns the plan
static MlPlan* ml_features(Ctx* c, const std::vector<Frame>& frames, int upsample)
{
    const DetectorModel& m = c->det;
    const int B = (int)frames.size();
    MlPlan* p = ml_build_pyramid(c, frames, upsample);
    const uint8_t* lut2 = orientation_lut_tiled(c);
    c->s_grad.ensure(p->px_elems * 4 + 256);
    c->s_hist.ensure(p->hist_floats * sizeof(float) + 64);
    c->s_norm.ensure(p->norm_floats * sizeof(float) + 64);
    c->s_feat.ensure(p->feat_floats * sizeof(float) + 64);
    uint32_t* d_px = c->s_grad.as<uint32_t>();
    {
        ProfScope ps(c, "fhog");
        {
            ProfScope p1(c, "fhog_grad");
            hipLaunchKernelGGL(fhog_grad4r_ml_k, dim3(ml_grid(p->grad_blocks)), dim3(256), 0, c->stream, p->grad, p->d_lv, B, c->s_pyr.as<uint8_t>(), d_px, lut2);
        }
        {
            ProfScope p2(c, "fhog_hist");
            hipLaunchKernelGGL(fhog_hist4_ml_k, dim3(ml_grid(p->hist_blocks)), dim3(256), 0, c->stream, p->hist, p->d_lv, B, d_px, c->s_hist.as<float>(),
                               c->s_norm.as<float>());
        }
        if (p->feat_blocks > 0) {
            ProfScope p3(c, "fhog_feat");
            hipLaunchKernelGGL(fhog_feat_ml_k, dim3(ml_grid(p->feat_blocks)), dim3(256), 0, c->stream, p->feat, p->d_lv, B, c->s_hist.as<float>(),
                               c->s_norm.as<float>(), c->s_feat.as<float>(), (m.frows - 1) / 2, (m.fcols - 1) / 2);
        }
    }
    return p;
}

static void det_run_batch_ml(Ctx* c, const std::vector<Frame>& frames, int upsample, const ScoreParams& sp0, int* d_counts, CandRec* d_cands)
{
    const DetectorModel& m = c->det;
    const int B = (int)frames.size();
    MlPlan* p = ml_features(c, frames, upsample);
    if (p->score_blocks > 0) {
        ProfScope ps(c, "score");
        const size_t lds = (size_t)4 * (2 * 48 + 11) * 34 * sizeof(float);
        const float4* b4 = reinterpret_cast<const float4*>(m.d_bmfma4);
        if (score_rows_per_wave() == 2)
            hipLaunchKernelGGL(score_mfma_rows_ml_k<2>, dim3(ml_grid(p->score_blocks)), dim3(256), lds, c->stream, p->score, p->d_lv, B, c->s_feat.as<float>(), b4,
                               sp0, d_counts, d_cands);
        else
            hipLaunchKernelGGL(score_mfma_rows_ml_k<4>, dim3(ml_grid(p->score_blocks)), dim3(256), lds, c->stream, p->score, p->d_lv, B, c->s_feat.as<float>(), b4,
                               sp0, d_counts, d_cands);
    }
}

void det_pyramid_level(Ctx* c, const Frame& f, int upsample, int level, std::vector<uint8_t>* out, int* oh, int* ow)
{
    PVF_REQUIRE(c->det.loaded, "detector not loaded");
    std::vector<Frame> fr{f};
    MlPlan* p = ml_build_pyramid(c, fr, upsample);
    PVF_REQUIRE(level >= 0 && level < (int)p->lv.size(), "pyramid level out of range");
    *oh = p->lv[level].h; *ow = p->lv[level].w;
    if (out) {
        out->resize((size_t)(*oh) * (*ow) * 3);
        HIP_CHECK(hipMemcpy2DAsync(out->data(), (size_t)(*ow) * 3, c->s_pyr.as<uint8_t>() + p->lv[level].img_off, (size_t)p->lv[level].rb,
                                   (size_t)(*ow) * 3, (size_t)(*oh), hipMemcpyDeviceToHost, c->stream));
    }
    HIP_CHECK(hipStreamSynchronize(c->stream));
}

// features of one pyramid level as the batched detector computes them (parity tests of the multi-level FHOG kernels)
void det_level_features(Ctx* c, const Frame& f, int upsample, int level, std::vector<float>* out, int* fh, int* fw)
{
    PVF_REQUIRE(c->det.loaded, "detector not loaded");
    std::vector<Frame> fr{f};
    MlPlan* p = ml_features(c, fr, upsample);
    PVF_REQUIRE(level >= 0 && level < (int)p->lv.size(), "pyramid level out of range");
    const LvDesc& d = p->lv[level];
    *fh = d.fh; *fw = d.fw;
    if (out) {
        out->resize((size_t)d.fh * d.fw * PVF_FHOG_STRIDE);
        if (!out->empty())
            HIP_CHECK(hipMemcpyAsync(out->data(), c->s_feat.as<float>() + d.feat_off, out->size() * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    }
    HIP_CHECK(hipStreamSynchronize(c->stream));
}

static bool raw_less(const RawDet& x, const RawDet& y)
{
    if (x.score != y.score) return x.score > y.score;
    if (x.filter != y.filter) return x.filter < y.filter;
    if (x.level != y.level) return x.level < y.level;
    if (x.r != y.r) return x.r < y.r;
    return x.c < y.c;
}

static void decode_candidates(const DetectorModel& m, int upsample, const CandRec* q, int n, std::vector<RawDet>& v);

void det_run_batch(Ctx* c, const std::vector<Frame>& frames, int upsample, double adjust, std::vector<std::vector<RawDet>>& raw_sorted)
{
    const DetectorModel& m = c->det;
    PVF_REQUIRE(m.loaded, "detector not loaded");
    PVF_REQUIRE(!frames.empty(), "no frames");
    const int B = (int)frames.size();
    const int cap = 8192;
    c->s_cand.ensure((size_t)B * cap * sizeof(CandRec) + (size_t)B * sizeof(int) + 64);
    int* d_counts = c->s_cand.as<int>();
    CandRec* d_cands = reinterpret_cast<CandRec*>(c->s_cand.as<uint8_t>() + (((size_t)B * sizeof(int) + 63) / 64) * 64);
    HIP_CHECK(hipMemsetAsync(d_counts, 0, (size_t)B * sizeof(int), c->stream));
    ScoreParams sp;
    for (int f = 0; f < 8; ++f) sp.thresh[f] = f < m.n_filters ? (float)((double)m.thresh[f] + adjust) : 3.0e38f;
    sp.n_filters = m.n_filters; sp.cap = cap;
    static const bool per_level = getenv("PVF_PER_LEVEL") != nullptr;
    if (m.n_filters == 5 && m.d_bmfma && !per_level) {
        det_run_batch_ml(c, frames, upsample, sp, d_counts, d_cands);
    } else
    run_pyramid(c, frames, upsample, -1, [&](int l, const uint8_t* img, int h, int w) {
        int fh, fw;
        fhog_dims(h, w, m.cell, m.frows, m.fcols, &fh, &fw);
        if (fh < m.frows || fw < m.fcols) return;
        const size_t feat_stride = (size_t)fh * fw * PVF_FHOG_STRIDE;
        c->s_feat.ensure(feat_stride * B * sizeof(float));
        {
            ProfScope ps(c, "fhog");
            fhog_device(c, img, B, h, w, m.cell, m.frows, m.fcols, c->s_feat.as<float>(), c->s_hist, c->s_norm, (((size_t)h * w * 3 + 15) & ~(size_t)15));
        }
        {
            ProfScope ps(c, "score");
            sp.level = l;
            const int out_r = fh - 9, out_c = fw - 9;
            dim3 grid((out_c + 31) / 32, (out_r + 7) / 8, B);
            const size_t lds = (size_t)(8 + 9) * (32 + 9) * 36 * sizeof(float);
#define LAUNCH_SCORE(NF)                                                                                                       \
    {                                                                                                                          \
        static bool attr_set = false;                                                                                          \
        if (!attr_set) {                                                                                                       \
            HIP_CHECK(hipFuncSetAttribute((const void*)score_k<NF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));    \
            attr_set = true;                                                                                                   \
        }                                                                                                                      \
    }                                                                                                                        \
    hipLaunchKernelGGL((score_k<NF>), grid, dim3(256), lds, c->stream, c->s_feat.as<float>(), feat_stride, fh, fw, m.d_w, sp, \
                       d_counts, d_cands)
            if (false) {
            } else if (m.n_filters == 5 && m.d_bmfma) {
                const size_t lds2 = (size_t)4 * (2 * 48 + 11) * 34 * sizeof(float);
                dim3 g2((out_c + 95) / 96, (out_r + 3) / 4, B);
                hipLaunchKernelGGL(score_mfma_k, g2, dim3(256), lds2, c->stream, c->s_feat.as<float>(), feat_stride, fh, fw, m.d_bmfma, sp,
                                   d_counts, d_cands);
            } else
            switch (m.n_filters) {
                case 1: LAUNCH_SCORE(1); break;
                case 2: LAUNCH_SCORE(2); break;
                case 3: LAUNCH_SCORE(3); break;
                case 4: LAUNCH_SCORE(4); break;
                case 5: LAUNCH_SCORE(5); break;
                case 6: LAUNCH_SCORE(6); break;
                case 7: LAUNCH_SCORE(7); break;
                default: LAUNCH_SCORE(8); break;
            }
#undef LAUNCH_SCORE
        }
    });
    HIP_CHECK(hipGetLastError());
    c->h_cand.ensure((size_t)B * cap * sizeof(CandRec) + (size_t)B * sizeof(int) + 64);
    int* h_counts = c->h_cand.as<int>();
    CandRec* h_cands = reinterpret_cast<CandRec*>(c->h_cand.as<uint8_t>() + (((size_t)B * sizeof(int) + 63) / 64) * 64);
    HIP_CHECK(hipMemcpyAsync(h_counts, d_counts, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    raw_sorted.assign(B, {});
    for (int b = 0; b < B; ++b) {
        const int n = h_counts[b];
        if (n > cap) throw PvfError("detector: candidate buffer overflow (threshold far too low for this input)");
        if (n == 0) continue;
        HIP_CHECK(hipMemcpyAsync(h_cands + (size_t)b * cap, d_cands + (size_t)b * cap, (size_t)n * sizeof(CandRec), hipMemcpyDeviceToHost,
                                 c->stream));
    }
    HIP_CHECK(hipStreamSynchronize(c->stream));
    for (int b = 0; b < B; ++b) decode_candidates(m, upsample, h_cands + (size_t)b * cap, h_counts[b], raw_sorted[b]);
}

// ---- many batches with the host work of batch j hidden behind the kernels of batch j + 1 ------------------------------------
// Per batch the host has to wait for the candidate counts, copy the candidates, map them to image rectangles and sort them;
// done batch by batch that leaves the GPU idle for ~1 ms out of every ~13.  Here the kernels of the next batch are queued
// before the results of the current one are collected.  Scratch (pyramid, planes, features) is shared: stream order keeps
// batch j + 1 from touching it before batch j is done; only the candidate buffers and the frame-pointer tables alternate.
static const int DET_PREFETCH = 512;      // candidates per frame copied back unconditionally (more are fetched on demand)

static void decode_candidates(const DetectorModel& m, int upsample, const CandRec* q, int n, std::vector<RawDet>& v)
{
    const int bw = m.fcols - 2 * m.padding, bh = m.frows - 2 * m.padding;
    v.resize(n);
    for (int i = 0; i < n; ++i) {
        long rect[4];
        const long cl = q[i].c - bw / 2, ct = q[i].r - bh / 2;
        fhog_to_image(cl, ct, m.cell, m.frows, m.fcols, &rect[0], &rect[1]);
        fhog_to_image(cl + bw - 1, ct + bh - 1, m.cell, m.frows, m.fcols, &rect[2], &rect[3]);
        for (int k = 0; k < q[i].level; ++k) rect_up6(rect);
        for (int u = 0; u < upsample; ++u) rect_down2i(rect);
        v[i] = RawDet{q[i].score, q[i].filter, q[i].level, q[i].r, q[i].c, (int32_t)rect[0], (int32_t)rect[1], (int32_t)rect[2], (int32_t)rect[3]};
    }
    std::sort(v.begin(), v.end(), raw_less);
}

void det_run_many(Ctx* c, const std::vector<Frame>& frames, int batch, int upsample, double adjust,
                  std::vector<std::vector<RawDet>>& raw_sorted)
{
    const DetectorModel& m = c->det;
    PVF_REQUIRE(m.loaded, "detector not loaded");
    PVF_REQUIRE(!frames.empty() && batch > 0, "no frames");
    const int N = (int)frames.size();
    raw_sorted.assign(N, {});
    static const bool per_level = getenv("PVF_PER_LEVEL") != nullptr;
    if (!(m.n_filters == 5 && m.d_bmfma && !per_level) || N <= batch) {
        for (int o = 0; o < N; o += batch) {                   // plain batch-by-batch form
            std::vector<Frame> fr(frames.begin() + o, frames.begin() + std::min(N, o + batch));
            std::vector<std::vector<RawDet>> part;
            det_run_batch(c, fr, upsample, adjust, part);
            for (size_t i = 0; i < part.size(); ++i) raw_sorted[o + i] = std::move(part[i]);
        }
        return;
    }
    const int cap = 8192, PF = DET_PREFETCH;
    ScoreParams sp;
    for (int f = 0; f < 8; ++f) sp.thresh[f] = f < m.n_filters ? (float)((double)m.thresh[f] + adjust) : 3.0e38f;
    sp.n_filters = m.n_filters; sp.cap = cap;
    const size_t cnt_bytes = (((size_t)batch * sizeof(int) + 63) / 64) * 64;
    for (int k = 0; k < 2; ++k) {
        c->s_cand2[k].ensure(cnt_bytes + (size_t)batch * cap * sizeof(CandRec));
        c->h_cand2[k].ensure(cnt_bytes + (size_t)batch * PF * sizeof(CandRec));
        if (!c->det_ev[k]) HIP_CHECK(hipEventCreateWithFlags(&c->det_ev[k], hipEventDisableTiming));
    }
    auto submit = [&](int o, int slot) {
        std::vector<Frame> fr(frames.begin() + o, frames.begin() + std::min(N, o + batch));
        const int B = (int)fr.size();
        int* d_counts = c->s_cand2[slot].as<int>();
        CandRec* d_cands = reinterpret_cast<CandRec*>(c->s_cand2[slot].as<uint8_t>() + cnt_bytes);
        HIP_CHECK(hipMemsetAsync(d_counts, 0, (size_t)B * sizeof(int), c->stream));
        c->det_slot = slot;
        det_run_batch_ml(c, fr, upsample, sp, d_counts, d_cands);
        HIP_CHECK(hipGetLastError());
        uint8_t* hb = c->h_cand2[slot].as<uint8_t>();
        HIP_CHECK(hipMemcpyAsync(hb, d_counts, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIP_CHECK(hipMemcpy2DAsync(hb + cnt_bytes, (size_t)PF * sizeof(CandRec), d_cands, (size_t)cap * sizeof(CandRec),
                                   (size_t)PF * sizeof(CandRec), (size_t)B, hipMemcpyDeviceToHost, c->stream));
        HIP_CHECK(hipEventRecord(c->det_ev[slot], c->stream));
    };
    auto collect = [&](int o, int slot) {
        const int B = std::min(N, o + batch) - o;
        HIP_CHECK(hipEventSynchronize(c->det_ev[slot]));
        const uint8_t* hb = c->h_cand2[slot].as<uint8_t>();
        const int* h_counts = reinterpret_cast<const int*>(hb);
        const CandRec* h_cands = reinterpret_cast<const CandRec*>(hb + cnt_bytes);
        const CandRec* d_cands = reinterpret_cast<const CandRec*>(c->s_cand2[slot].as<uint8_t>() + cnt_bytes);
        std::vector<CandRec> big;
        for (int b = 0; b < B; ++b) {
            const int n = h_counts[b];
            if (n > cap) throw PvfError("detector: candidate buffer overflow (threshold far too low for this input)");
            if (n <= PF) { decode_candidates(m, upsample, h_cands + (size_t)b * PF, n, raw_sorted[o + b]); continue; }
            big.resize(n);                                      // rare: more candidates than were copied back ahead
            HIP_CHECK(hipMemcpyAsync(big.data(), d_cands + (size_t)b * cap, (size_t)n * sizeof(CandRec), hipMemcpyDeviceToHost, c->stream));
            HIP_CHECK(hipStreamSynchronize(c->stream));
            decode_candidates(m, upsample, big.data(), n, raw_sorted[o + b]);
        }
    };
    submit(0, 0);
    int slot = 0;
    for (int o = 0; o < N; o += batch) {
        if (o + batch < N) submit(o + batch, slot ^ 1);
        collect(o, slot);
        slot ^= 1;
    }
    c->det_slot = 0;
}

static bool boxes_overlap(const RawDet& a, const RawDet& b, double iou, double covered)
{
    const long il = std::max(a.l, b.l), it = std::max(a.t, b.t), ir = std::min(a.rr, b.rr), ib = std::min(a.b, b.b);
    if (il > ir || it > ib) return false;
    const double inner = (double)(ir - il + 1) * (double)(ib - it + 1);
    const long ol = std::min(a.l, b.l), ot = std::min(a.t, b.t), orr = std::max(a.rr, b.rr), ob = std::max(a.b, b.b);
    const double outer = (double)(orr - ol + 1) * (double)(ob - ot + 1);
    const double aa = (double)(a.rr - a.l + 1) * (double)(a.b - a.t + 1);
    const double ab = (double)(b.rr - b.l + 1) * (double)(b.b - b.t + 1);
    return inner / outer > iou || inner / aa > covered || inner / ab > covered;
}

void det_nms(const DetectorModel& m, const std::vector<RawDet>& sorted, std::vector<RawDet>& out)
{
    out.clear();
    for (const RawDet& d : sorted) {
        bool hit = false;
        for (const RawDet& k : out) if (boxes_overlap(k, d, m.nms_iou, m.nms_covered)) { hit = true; break; }
        if (!hit) out.push_back(d);
    }
}
