// Dev-time probe (not part of the library): how fast does one CU issue v_mfma_f32_16x16x4_f32 under the scoring kernel's
// instruction mix?  mode bit 0: 8 x buffer_load_dwordx4 of B fragments per 64 MFMAs (245 KB table, L2-resident);
// bit 1: 8 x ds_read2_b32 of A fragments per 64 MFMAs; bit 2: fragments consumed one step later (software pipeline) instead of constants.
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_probe tools/probes/mfma_probe.hip ; run: /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int MODE, int CHAINS>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
probe_k(const float4* __restrict__ Bg, float* __restrict__ out, int iters)
{
    extern __shared__ __attribute__((aligned(16))) float slab[];
    const int lane = threadIdx.x;
    for (int i = lane; i < 107 * 34; i += 64) slab[i] = (float)(i & 15) * 0.001f;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc((void*)Bg, 0, 10 * 12 * 2 * 64 * 16, 0x00020000);
    const int lane16 = lane * 16;
    const float* a0 = slab + (3 * (lane & 15)) * 34 + (lane >> 4);
    f32x4 acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    u32x4 bn[4][2];
    float an[16];
#pragma unroll
    for (int j = 0; j < 4; ++j) { bn[j][0] = (u32x4){1, 2, 3, 4}; bn[j][1] = (u32x4){5, 6, 7, 8}; }
#pragma unroll
    for (int q = 0; q < 16; ++q) an[q] = (float)lane * 0.01f + q;
    int off = (blockIdx.x * 7) % 120;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < 12; ++n) {
            float ac[16], bc[4][8];
#pragma unroll
            for (int q = 0; q < 16; ++q) ac[q] = an[q];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const uint32_t b0 = bn[j][h].x, b1 = bn[j][h].y, b2 = bn[j][h].z, b3 = bn[j][h].w;
                    bc[j][4 * h] = __uint_as_float(b0); bc[j][4 * h + 1] = __uint_as_float(b1);
                    bc[j][4 * h + 2] = __uint_as_float(b2); bc[j][4 * h + 3] = __uint_as_float(b3);
                }
            if (MODE & 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int o = (((off + j * 12 + n) % 120) * 2048);
                    bn[j][0] = __builtin_amdgcn_raw_buffer_load_b128(brs, lane16, o, 0);
                    bn[j][1] = __builtin_amdgcn_raw_buffer_load_b128(brs, lane16, o + 1024, 0);
                }
            }
            if (MODE & 2) {
#pragma unroll
                for (int pq = 0; pq < 8; ++pq)
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) an[pq * 2 + tt] = a0[(tt * 48 + n) * 34 + 4 * pq];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int pq = 0; pq < 8; ++pq)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) {
                        const int k = (j * 2 + tt) % CHAINS;
                        acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[pq * 2 + tt], bc[j][pq], acc[k], 0, 0, 0);
                    }
            __builtin_amdgcn_sched_barrier(0);
        }
        off = (off + 12) % 120;
    }
    float r = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) r += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    out[blockIdx.x * 64 + lane] = r;
}

template <int MODE, int CHAINS>
static void run(const char* name, const float4* dB, float* dout, int waves_per_cu, size_t lds)
{
    const int grid = 256 * waves_per_cu, iters = 40;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((probe_k<MODE, CHAINS>), dim3(grid), dim3(64), lds, 0, dB, dout, 2);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((probe_k<MODE, CHAINS>), dim3(grid), dim3(64), lds, 0, dB, dout, iters);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double flop = (double)grid * iters * 12 * 64 * 2048.0;
    printf("%-44s waves/CU %2d  %.3f ms  %.1f TFLOP/s issued\n", name, waves_per_cu, ms, flop / (ms * 1e-3) / 1e12);
}

int main()
{
    float4* dB; float* dout;
    const size_t nb = (size_t)10 * 12 * 2 * 64;
    std::vector<float4> hb(nb);
    for (size_t i = 0; i < nb; ++i) hb[i] = make_float4(0.001f * (i % 7), 0.002f, 0.0f, 0.003f);
    CK(hipMalloc((void**)&dB, nb * sizeof(float4)));
    CK(hipMemcpy(dB, hb.data(), nb * sizeof(float4), hipMemcpyHostToDevice));
    CK(hipMalloc((void**)&dout, (size_t)256 * 16 * 64 * sizeof(float)));
    const size_t lds = 107 * 34 * 4;
    for (int w : {8, 4}) {
        run<0, 8>("MFMA only, 8 chains", dB, dout, w, lds);
        run<0, 2>("MFMA only, 2 chains", dB, dout, w, lds);
        run<0, 1>("MFMA only, 1 chain", dB, dout, w, lds);
        run<1, 8>("+ 8 buffer_load_dwordx4 / 64 MFMA", dB, dout, w, lds);
        run<2, 8>("+ 16 ds_read_b32 / 64 MFMA", dB, dout, w, lds);
        run<3, 8>("+ both (scoring kernel's steady state)", dB, dout, w, lds);
    }
    // 16 resident waves per CU would need < 128 VGPRs: not this kernel; 8 = the scoring kernel's occupancy
    return 0;
}
