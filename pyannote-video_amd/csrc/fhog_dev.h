// fhog_dev.h -- device helpers shared by the FHOG kernels (detector: detect.hip, tracker chips: fhog.hip).
// All arithmetic follows the orders stated in oracle/pvo_fhog.c.
#pragma once
#include "pvf_internal.h"

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4u __attribute__((aligned(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// orientation-bin tables (fhog.hip): row-major 511 x 511 ...
const uint8_t* orientation_lut(Ctx* c);
// ... and addressed by the two differences mod 512 (X = bx & 511, Y = by & 511): (X & 7) | Y << 3 | (X >> 3) << 12 -- 8 x 8 tiles again
const uint8_t* orientation_lut_wrapped(Ctx* c);

// gradient of the colour channel with the largest |g|^2 (first wins): squared magnitude and orientation bin (row-major table)
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

// correctly rounded sqrt of an integer-valued float in 0 .. 2 * 255^2 (a gradient's squared magnitude).  On this domain v_sqrt_f32 is
// never above the correctly rounded root and at most one step below it (20 919 of the 130 051 inputs: tools/probes/sqrt_probe.hip;
// checked for every input once per context, detect.hip: check_device_assumptions -- a mismatch is an error), so ONE exact residual
// decides: the next float up is the answer when x lies above s * s_up, the square of the midpoint up to a quarter step squared.
// Same result as sqrtf(); 5 instructions instead of its denormal scaling, class checks and two-sided correction (9).
__device__ __forceinline__ float sqrt_exact_small(float x)
{
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sp = __uint_as_float(__float_as_uint(s) + 1u);
    return fmaf(-sp, s, x) > 0.0f ? sp : s;
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

