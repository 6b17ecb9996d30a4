#include <hip/hip_runtime.h>
#include <cstdio>
extern __shared__ float sm[];
__global__ void __launch_bounds__(256) k240(float* o) { o[threadIdx.x] = sm[threadIdx.x]; }
int main() {
    for (int lds : {40000, 43956, 53000, 65536, 80820, 81920, 82000, 87912}) {
        hipFuncSetAttribute((const void*)k240, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        int n = -1; hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k240, 256, lds);
        printf("lds %d -> %d blocks/CU (err %d)\n", lds, n, (int)e);
    }
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("sharedMemPerMultiprocessor %zu maxSharedMemoryPerMultiProcessor %zu perBlock %zu optin %zu\n", p.sharedMemPerMultiprocessor, p.maxSharedMemoryPerMultiProcessor, p.sharedMemPerBlock, p.sharedMemPerBlockOptin);
    return 0;
}
