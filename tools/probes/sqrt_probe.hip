// Is v_sqrt_f32 correctly rounded for every integer-valued input the FHOG gradient magnitude can take (0 .. 2 * 255^2)?
// fhog_dev.h steps the hardware estimate to the correctly rounded neighbour with two residual checks (8 VALU instructions per pixel);
// if the estimate is already exact on this domain those steps can go.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O2 tools/probes/sqrt_probe.hip -o /tmp/sqrt_probe && /tmp/sqrt_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void probe(int n, float* hw)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) hw[i] = __builtin_amdgcn_sqrtf((float)i);
}
int main()
{
    const int n = 2 * 255 * 255 + 1;
    float* d;
    hipMalloc(&d, n * sizeof(float));
    hipLaunchKernelGGL(probe, dim3((n + 255) / 256), dim3(256), 0, 0, n, d);
    std::vector<float> h(n);
    hipMemcpy(h.data(), d, n * sizeof(float), hipMemcpyDeviceToHost);
    int bad = 0, first = -1, above = 0, below = 0, far = 0;
    for (int i = 0; i < n; ++i) {
        const float want = (float)std::sqrt((double)i);          // correctly rounded: sqrt in double, then one rounding (exact for these inputs)
        if (h[i] != want) {
            if (first < 0) first = i;
            ++bad;
            if (h[i] > want) ++above; else ++below;
            if (h[i] != std::nextafterf(want, 0.0f) && h[i] != std::nextafterf(want, 1e30f)) ++far;
        }
    }
    printf("v_sqrt_f32 on integers 0..%d: %d of %d differ from the correctly rounded value (first at %d): %d above it, %d below it, %d by more than one step\n",
           n - 1, bad, n, first, above, below, far);
    return 0;
}
