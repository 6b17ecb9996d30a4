// Issue cost of the instructions the resize and FHOG kernels are made of, measured on the device: 1, 2 and 4 waves per SIMD (one block of 256 / 512 / 1024 threads), each timing REP x 64 independent copies of one instruction with s_memtime.  Prints cycles per wave64 instruction per SIMD.
//   hipcc --offload-arch=gfx950 -O2 tools/microbench_isa.hip -o /tmp/microbench_isa && /tmp/microbench_isa
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R64(x) R4(R16(x))

#define BENCH(name, setup, body)                                                                                   \
    __global__ void __launch_bounds__(1024) k_##name(uint64_t* out, int reps)                                      \
    {                                                                                                              \
        __shared__ uint32_t lds[4096];                                                                             \
        lds[threadIdx.x] = threadIdx.x; lds[threadIdx.x + 1024] = 1;                                                \
        __syncthreads();                                                                                           \
        uint32_t a3 = (threadIdx.x * 3) & 4095, a4 = (threadIdx.x * 4) & 4095, a36 = (threadIdx.x * 18 / 5) & 4095;\
        (void)a3; (void)a4; (void)a36;                                                                             \
        setup;                                                                                                     \
        uint64_t t0 = __builtin_readcyclecounter();                                                                \
        for (int r = 0; r < reps; ++r) { R64(body) }                                                               \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");                                                             \
        uint64_t t1 = __builtin_readcyclecounter();                                                                \
        if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;                                              \
    }

BENCH(v_mul_f32, float x = threadIdx.x; float y, asm volatile("v_mul_f32 %0, %1, %1" : "=v"(y) : "v"(x));)
BENCH(v_pk_mul_f32, double x = threadIdx.x; double y, asm volatile("v_pk_mul_f32 %0, %1, %1" : "=v"(y) : "v"(x));)
BENCH(v_mul_f64, double x = threadIdx.x; double y, asm volatile("v_mul_f64 %0, %1, %1" : "=v"(y) : "v"(x));)
BENCH(v_add_f64, double x = threadIdx.x; double y, asm volatile("v_add_f64 %0, %1, %1" : "=v"(y) : "v"(x));)
BENCH(v_fma_f64, double x = threadIdx.x; double y, asm volatile("v_fma_f64 %0, %1, %1, %1" : "=v"(y) : "v"(x));)
BENCH(v_cvt_f64_u32, uint32_t x = threadIdx.x; double y, asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(y) : "v"(x));)
BENCH(v_cvt_i32_f64, double x = threadIdx.x; uint32_t y, asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(y) : "v"(x));)
BENCH(v_cvt_f32_ubyte0, uint32_t x = threadIdx.x; float y, asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(y) : "v"(x));)
BENCH(v_cvt_f32_u32, uint32_t x = threadIdx.x; float y, asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(y) : "v"(x));)
BENCH(v_cvt_u32_f32, float x = threadIdx.x; uint32_t y, asm volatile("v_cvt_u32_f32 %0, %1" : "=v"(y) : "v"(x));)
BENCH(v_mov_b64, double x = threadIdx.x; double y, asm volatile("v_mov_b64 %0, %1" : "=v"(y) : "v"(x));)
BENCH(v_lshl_add_u64, uint64_t x = threadIdx.x; uint64_t y, asm volatile("v_lshl_add_u64 %0, %1, 0, %1" : "=v"(y) : "v"(x));)
BENCH(v_and_or_b32, uint32_t x = threadIdx.x; uint32_t y, asm volatile("v_and_or_b32 %0, %1, %1, %1" : "=v"(y) : "v"(x));)
BENCH(v_rsq_f32, float x = threadIdx.x + 1.f; float y, asm volatile("v_rsq_f32 %0, %1" : "=v"(y) : "v"(x));)
BENCH(v_sqrt_f32, float x = threadIdx.x + 1.f; float y, asm volatile("v_sqrt_f32 %0, %1" : "=v"(y) : "v"(x));)
BENCH(v_mul_lo_u32, uint32_t x = threadIdx.x; uint32_t y, asm volatile("v_mul_lo_u32 %0, %1, %1" : "=v"(y) : "v"(x));)
BENCH(v_mad_u32_u24, uint32_t x = threadIdx.x; uint32_t y, asm volatile("v_mad_u32_u24 %0, %1, %1, %1" : "=v"(y) : "v"(x));)
BENCH(ds_read_u8_s36, uint32_t y, asm volatile("ds_read_u8 %0, %1" : "=v"(y) : "v"(a36));)
BENCH(ds_read_b32, uint32_t y, asm volatile("ds_read_b32 %0, %1" : "=v"(y) : "v"(a4));)
BENCH(ds_read_b64, uint64_t y, asm volatile("ds_read_b64 %0, %1" : "=v"(y) : "v"(a4 * 2));)
BENCH(ds_read_b128_s16, __uint128_t y, asm volatile("ds_read_b128 %0, %1" : "=v"(y) : "v"((a4 * 4) & 4095 * 4));)
BENCH(ds_write_b8_s3, uint32_t x = threadIdx.x, asm volatile("ds_write_b8 %0, %1" : : "v"(a3), "v"(x));)
BENCH(ds_write_b32, uint32_t x = threadIdx.x, asm volatile("ds_write_b32 %0, %1" : : "v"(a4), "v"(x));)
BENCH(ds_add_u32_s4, uint32_t x = threadIdx.x, asm volatile("ds_add_u32 %0, %1" : : "v"(a4), "v"(x));)
BENCH(ds_add_f32_s4, float x = threadIdx.x, asm volatile("ds_add_f32 %0, %1" : : "v"(a4), "v"(x));)

BENCH(v_add_f32, float x = threadIdx.x; float y, asm volatile("v_add_f32 %0, %1, %1" : "=v"(y) : "v"(x));)
BENCH(v_fma_f32, float x = threadIdx.x; float y, asm volatile("v_fma_f32 %0, %1, %1, %1" : "=v"(y) : "v"(x));)
BENCH(v_fmac_f32, float x = threadIdx.x; float y = 1.f, asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(y) : "v"(x));)
BENCH(v_min_f32, float x = threadIdx.x; float y, asm volatile("v_min_f32 %0, %1, %1" : "=v"(y) : "v"(x));)
BENCH(v_pk_add_f32, double x = threadIdx.x; double y, asm volatile("v_pk_add_f32 %0, %1, %1" : "=v"(y) : "v"(x));)
BENCH(v_pk_fma_f32, double x = threadIdx.x; double y, asm volatile("v_pk_fma_f32 %0, %1, %1, %1" : "=v"(y) : "v"(x));)
BENCH(v_add_u32, uint32_t x = threadIdx.x; uint32_t y, asm volatile("v_add_u32 %0, %1, %1" : "=v"(y) : "v"(x));)
BENCH(v_and_b32, uint32_t x = threadIdx.x; uint32_t y, asm volatile("v_and_b32 %0, %1, %1" : "=v"(y) : "v"(x));)
BENCH(v_mov_b32, uint32_t x = threadIdx.x; uint32_t y, asm volatile("v_mov_b32 %0, %1" : "=v"(y) : "v"(x));)
BENCH(v_lshlrev_b32, uint32_t x = threadIdx.x; uint32_t y, asm volatile("v_lshlrev_b32 %0, 3, %1" : "=v"(y) : "v"(x));)
BENCH(v_lshl_add_u32, uint32_t x = threadIdx.x; uint32_t y, asm volatile("v_lshl_add_u32 %0, %1, 2, %1" : "=v"(y) : "v"(x));)
BENCH(v_max_i32, uint32_t x = threadIdx.x; uint32_t y, asm volatile("v_max_i32 %0, %1, %1" : "=v"(y) : "v"(x));)
BENCH(v_max3_i32, uint32_t x = threadIdx.x; uint32_t y, asm volatile("v_max3_i32 %0, %1, %1, %1" : "=v"(y) : "v"(x));)
BENCH(v_cmp_lt_f32, float x = threadIdx.x, asm volatile("v_cmp_lt_f32 vcc, %0, %0" : : "v"(x) : "vcc");)
BENCH(v_sub_u32_sdwa, uint32_t x = threadIdx.x; uint32_t y, asm volatile("v_sub_u32_sdwa %0, %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2" : "=v"(y) : "v"(x));)
BENCH(v_mul_i32_i24, uint32_t x = threadIdx.x; uint32_t y, asm volatile("v_mul_i32_i24 %0, %1, %1" : "=v"(y) : "v"(x));)
BENCH(v_mad_i32_i24, uint32_t x = threadIdx.x; uint32_t y, asm volatile("v_mad_i32_i24 %0, %1, %1, %1" : "=v"(y) : "v"(x));)
BENCH(v_mov_b32_dpp, uint32_t x = threadIdx.x; uint32_t y = 0, asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(y) : "v"(x));)
BENCH(v_perm_b32, uint32_t x = threadIdx.x; uint32_t y, asm volatile("v_perm_b32 %0, %1, %1, %1" : "=v"(y) : "v"(x));)
BENCH(v_or3_b32, uint32_t x = threadIdx.x; uint32_t y, asm volatile("v_or3_b32 %0, %1, %1, %1" : "=v"(y) : "v"(x));)
BENCH(v_cvt_f32_f16, uint32_t x = threadIdx.x; float y, asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(y) : "v"(x));)
BENCH(v_dot2_f32_f16, uint32_t x = threadIdx.x; float y, asm volatile("v_dot2_f32_f16 %0, %1, %1, %1" : "=v"(y) : "v"(x));)
BENCH(v_dot4_i32_i8, uint32_t x = threadIdx.x; uint32_t y, asm volatile("v_dot4_i32_i8 %0, %1, %1, %1" : "=v"(y) : "v"(x));)
BENCH(v_sad_u8, uint32_t x = threadIdx.x; uint32_t y, asm volatile("v_sad_u8 %0, %1, %1, %1" : "=v"(y) : "v"(x));)
BENCH(ds_read2st64_b32, uint64_t y, asm volatile("ds_read2st64_b32 %0, %1 offset1:4" : "=v"(y) : "v"(a4));)
BENCH(ds_write2st64_b32, uint32_t x = threadIdx.x, asm volatile("ds_write2st64_b32 %0, %1, %1 offset1:4" : : "v"(a4), "v"(x));)
BENCH(ds_write_b64, uint64_t x = threadIdx.x, asm volatile("ds_write_b64 %0, %1" : : "v"(a4 * 2), "v"(x));)
BENCH(ds_write_b128, __uint128_t x = threadIdx.x, asm volatile("ds_write_b128 %0, %1" : : "v"((a4 * 4) & 16368), "v"(x));)
BENCH(buffer_load_ubyte_l1, uint32_t y, asm volatile("global_load_ubyte %0, %1, %2" : "=v"(y) : "v"(a36), "s"(out));)

struct Entry { const char* name; void (*k)(uint64_t*, int); };
#define E(name) { #name, k_##name }

int main()
{
    Entry es[] = { E(v_mul_f32), E(v_pk_mul_f32), E(v_mul_f64), E(v_add_f64), E(v_fma_f64), E(v_cvt_f64_u32), E(v_cvt_i32_f64), E(v_cvt_f32_ubyte0),
                   E(v_cvt_f32_u32), E(v_cvt_u32_f32), E(v_mov_b64), E(v_lshl_add_u64), E(v_and_or_b32), E(v_rsq_f32), E(v_sqrt_f32), E(v_mul_lo_u32),
                   E(v_mad_u32_u24), E(ds_read_u8_s36), E(ds_read_b32), E(ds_read_b64), E(ds_read_b128_s16), E(ds_write_b8_s3), E(ds_write_b32),
                   E(ds_add_u32_s4), E(ds_add_f32_s4),
                   E(v_add_f32), E(v_fma_f32), E(v_fmac_f32), E(v_min_f32), E(v_pk_add_f32), E(v_pk_fma_f32), E(v_add_u32), E(v_and_b32), E(v_mov_b32), E(v_lshlrev_b32), E(v_lshl_add_u32), E(v_max_i32), E(v_max3_i32), E(v_cmp_lt_f32), E(v_sub_u32_sdwa), E(v_mul_i32_i24), E(v_mad_i32_i24), E(v_mov_b32_dpp), E(v_perm_b32), E(v_or3_b32), E(v_cvt_f32_f16), E(v_dot2_f32_f16), E(v_dot4_i32_i8), E(v_sad_u8), E(ds_read2st64_b32), E(ds_write2st64_b32), E(ds_write_b64), E(ds_write_b128), E(buffer_load_ubyte_l1) };
    uint64_t* d;
    if (hipMalloc(&d, 128) != hipSuccess) { printf("no device\n"); return 1; }
    const int reps = 64;
    printf("# s_memtime ticks per wave64 instruction per SIMD, with 1 / 2 / 4 waves per SIMD (one block on one CU) | whole device, 8 waves per SIMD: wall ns x 2.4 per instruction per SIMD\n");
    for (auto& e : es) {
        printf("%-20s", e.name);
        for (int threads : {256, 512, 1024}) {                                  // 1, 2, 4 waves per SIMD
            uint64_t h[16];
            hipLaunchKernelGGL(e.k, dim3(1), dim3(threads), 0, 0, d, reps);     // warm
            hipLaunchKernelGGL(e.k, dim3(1), dim3(threads), 0, 0, d, reps);
            (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            const int waves = threads / 64;
            double c = 0;
            for (int i = 0; i < waves; ++i) c += (double)h[i];
            printf(" %7.2f", c / waves / (reps * 64.0) / (waves / 4));          // per instruction per SIMD
        }
        // the whole device, 8 waves per SIMD (2 blocks of 1024 per CU), by the wall clock: nanoseconds x 2.4 = cycles at the nominal clock
        {
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            const int blocks = 512, big = 1024;
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(1024), 0, 0, d, big);
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(1024), 0, 0, d, big);
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            const double per_simd = (double)blocks / 256 * 16 / 4 * big * 64.0;
            printf("   | %7.2f", ms * 1e6 * 2.4 / per_simd);
        }
        printf("\n");
    }
    int clk = 0;
    (void)hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("# device clock rate attribute: %d kHz\n", clk);
    return 0;
}
