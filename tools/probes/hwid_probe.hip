// where do the waves of a 512-thread block land?  prints SIMD_ID (HW_REG_HW_ID bits 5:4), CU_ID, WAVE_ID per wave for a few blocks
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void __launch_bounds__(512) k(unsigned* out, int spin)
{
    extern __shared__ unsigned lds[];
    const int wave = threadIdx.x >> 6;
    unsigned hw = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);      // HW_ID bits 15:0
    unsigned acc = 0;
    for (int i = 0; i < spin; ++i) acc += __builtin_amdgcn_s_memtime() & 1;  // keep the block resident for a while
    lds[threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = hw | (acc & 0x10000u);
}
int main()
{
    const int nb = 2048;
    unsigned* d; hipMalloc(&d, nb * 8 * 4);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 74240);
    hipLaunchKernelGGL(k, dim3(nb), dim3(512), 74240, 0, d, 2000);
    std::vector<unsigned> h(nb * 8);
    hipMemcpy(h.data(), d, nb * 8 * 4, hipMemcpyDeviceToHost);
    int hist[8][4] = {};
    int per_simd_bad = 0;
    for (int b = 0; b < nb; ++b) {
        int cnt[4] = {};
        for (int w = 0; w < 8; ++w) { const int simd = (h[b * 8 + w] >> 4) & 3; hist[w][simd]++; cnt[simd]++; }
        if (cnt[0] != 2 || cnt[1] != 2 || cnt[2] != 2 || cnt[3] != 2) ++per_simd_bad;
        if (b < 6 || b == 1000) { printf("block %d:", b); for (int w = 0; w < 8; ++w) printf(" w%d simd%u cu%u wv%u |", w, (h[b*8+w] >> 4) & 3, (h[b*8+w] >> 8) & 15, h[b*8+w] & 15); printf("\n"); }
    }
    for (int w = 0; w < 8; ++w) printf("wave %d: simd histogram %d %d %d %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    printf("blocks whose 8 waves are not 2 per SIMD: %d of %d\n", per_simd_bad, nb);
    return 0;
}
