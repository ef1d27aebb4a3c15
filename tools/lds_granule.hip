// GPU-box probe: workgroups per CU by dynamic LDS size (how the hardware rounds an LDS allocation).
//   hipcc --offload-arch=gfx950 -o gpurun_out/lds_granule tools/lds_granule.hip && gpurun_out/lds_granule
#include <hip/hip_runtime.h>
#include <cstdio>
extern "C" __global__ void probe(double *p) { extern __shared__ double s[]; s[threadIdx.x] = 1.0; p[threadIdx.x] = s[63 - threadIdx.x]; }
int main()
{
    hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int last = -1;
    for (int bytes = 12288; bytes <= 16384; bytes += 16) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void *)probe, 64, bytes) != hipSuccess) { printf("error at %d\n", bytes); return 1; }
        if (n != last) { printf("%d bytes of LDS: %d workgroups of 64 threads per CU\n", bytes, n); last = n; }
    }
    return 0;
}
