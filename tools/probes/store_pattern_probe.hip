// the feature stores of the FHOG kernels: a lane owns a 128-byte cell.  (A) eight 16-byte stores per lane, lanes 128 bytes apart (what a
// lane-per-cell kernel does) against (B) the same bytes with lane i of store s writing bytes 1024 s + 16 i of the wave's 8 KB run.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void __launch_bounds__(256) k(f32x4* out, int runs_per_wave, float v)
{
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    f32x4 x = {v, v + 1, v + 2, v + (float)lane};
    for (int r = 0; r < runs_per_wave; ++r) {
        f32x4* run = out + ((size_t)r * gridDim.x * 4 + wave) * 512;      // 8 KB per wave and step
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (MODE == 0) run[lane * 8 + s] = x;
            else run[s * 64 + lane] = x;
        }
    }
}
int main()
{
    const int blocks = 2048, runs = 108;                                      // 2048 * 4 * 108 * 8 KB = 7.25 GB
    const size_t bytes = (size_t)blocks * 4 * runs * 8192;
    f32x4* d; if (hipMalloc(&d, bytes) != hipSuccess) return 1;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int mode = 0; mode < 2; ++mode)
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(a);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, runs, 1.0f);
            else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, runs, 1.0f);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("%s: %.3f ms for %.2f GB = %.2f TB/s\n", mode == 0 ? "A lane-per-cell (16 B at a 128-byte stride)" : "B coalesced runs", ms, bytes / 1e9, bytes / ms / 1e9);
        }
    return 0;
}
