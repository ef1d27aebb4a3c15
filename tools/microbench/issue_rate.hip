// GPU box: how fast does ONE wave issue fp64 VALU instructions on gfx950 -- dependent chain vs 2 / 4 independent
// chains, plain and with a second wave on the same SIMD?  (shader clock via s_memtime)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/issue_rate tools/microbench/issue_rate.hip && /tmp/issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CHAINS>
__global__ void k(double *out, long long *cyc, int n, double a, double b)
{
    double x0 = out[threadIdx.x], x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    __syncthreads();
    long long t0 = clock64();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            x0 = __builtin_fma(x0, a, b);
            if (CHAINS > 1) x1 = __builtin_fma(x1, a, b);
            if (CHAINS > 2) { x2 = __builtin_fma(x2, a, b); x3 = __builtin_fma(x3, a, b); }
        }
    }
    long long t1 = clock64();
    out[threadIdx.x + blockDim.x * blockIdx.x] = x0 + x1 + x2 + x3;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int CHAINS>
__global__ void ks(double *out, long long *cyc, int n, int a, int b)     // scalar ALU chain
{
    int x0 = (int)out[0], x1 = x0 + 1;
    long long t0 = clock64();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            x0 = __builtin_amdgcn_readfirstlane(x0) * a + b;
            if (CHAINS > 1) x1 = __builtin_amdgcn_readfirstlane(x1) * a + b;
        }
    }
    long long t1 = clock64();
    out[threadIdx.x] = x0 + x1;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main()
{
    double *out; long long *cyc, h[8];
    hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 1024); hipMemset(out, 0, 1 << 20);
    const int n = 2000;
    for (int threads : {64, 128, 256, 512}) {
#define RUN(C) do { hipLaunchKernelGGL(k<C>, dim3(1), dim3(threads), 0, 0, out, cyc, n, 1.0000001, 1e-9); hipDeviceSynchronize(); \
        hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost); printf("threads %3d  fp64 fma, %d independent chain(s): %.2f clock64 ticks per instruction\n", threads, C, (double)h[0] / (n * 16.0 * C)); } while (0)
        RUN(1); RUN(2); RUN(4);
    }
    hipLaunchKernelGGL(ks<1>, dim3(1), dim3(64), 0, 0, out, cyc, n, 3, 1); hipDeviceSynchronize(); hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost);
    printf("scalar chain (readfirstlane + s_mul + s_add): %.2f ticks per iteration\n", (double)h[0] / (n * 16.0));
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    int wc = 0; hipDeviceGetAttribute(&wc, hipDeviceAttributeWallClockRate, 0);
    printf("clock rate %d kHz, wall clock rate %d kHz (clock64 = s_memtime counts at a constant rate)\n", clk, wc);
    return 0;
}
