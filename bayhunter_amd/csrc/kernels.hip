// kernels.hip -- gfx950 kernels of libbayhunter_amd (hand-written HIP, wave64).
//
//   swd_kernel : one lane per (model, dispersion target); one wave per workgroup; the fp32 layer
//                stacks of the workgroup's 64 models are staged once into LDS, layer-major
//                ([array][layer][lane], conflict-free), from coalesced reads of the batched fp64
//                model arrays; the search itself is swd_core.h (exact replay of surfdisp96.f).
//   rf_kernel  : one workgroup per M models; phases P1..P4 of rf_core.h with __syncthreads between.
//
// Both are fp64 scalar recurrences: bound by FP64 VALU issue + transcendental latency, not by HBM
// and not MFMA-shaped ("Roofline").  HBM traffic is the algorithmic minimum: each model
// byte is read once, each output written once; everything else lives in LDS/registers.
#include <hip/hip_runtime.h>
#include "kernels.h"
#include "rf_core.h"
#include "swd_core.h"

namespace bh {

// ------------------------------------------------------------------------------------------- SWD
struct LdsLay {
    float *base;  // lds + lane
    int L;        // layers per array
    __device__ __forceinline__ float d(int i) const { return base[(0 * L + i) * SWD_T]; }
    __device__ __forceinline__ float a(int i) const { return base[(1 * L + i) * SWD_T]; }
    __device__ __forceinline__ float b(int i) const { return base[(2 * L + i) * SWD_T]; }
    __device__ __forceinline__ float rho(int i) const { return base[(3 * L + i) * SWD_T]; }
    __device__ __forceinline__ void set_d(int i, float v) { base[(0 * L + i) * SWD_T] = v; }
    __device__ __forceinline__ void set_a(int i, float v) { base[(1 * L + i) * SWD_T] = v; }
    __device__ __forceinline__ void set_b(int i, float v) { base[(2 * L + i) * SWD_T] = v; }
    __device__ __forceinline__ void set_rho(int i, float v) { base[(3 * L + i) * SWD_T] = v; }
};

// 2 waves per SIMD: the search state + one Dunkin layer need ~250 VGPRs; pin the allocator there
__global__ __launch_bounds__(SWD_T) __attribute__((amdgpu_waves_per_eu(2, 2))) void swd_kernel(SwdArgs A)
{
    extern __shared__ float lds[];
    const int tid = threadIdx.x;
    const int b0 = blockIdx.x * SWD_T;
    const int t = blockIdx.y;
    const int L = A.Lmax;
    const int nrows = min(SWD_T, A.B - b0);

    // stage: rows b0..b0+nrows-1 of the [B][Lmax] fp64 arrays are one contiguous chunk each;
    // read it coalesced, round to fp32 like f2py (surf96_modsw.py:68-82), store layer-major.
    const int nelem = nrows * L;
    const long g0 = (long)b0 * L;
    for (int idx = tid; idx < nelem; idx += SWD_T) {
        int r = idx / L, l = idx - r * L;
        lds[(0 * L + l) * SWD_T + r] = (float)A.h[g0 + idx];
        lds[(1 * L + l) * SWD_T + r] = (float)A.vp[g0 + idx];
        lds[(2 * L + l) * SWD_T + r] = (float)A.vs[g0 + idx];
        lds[(3 * L + l) * SWD_T + r] = (float)A.rho[g0 + idx];
    }
    __syncthreads();

    const int b = b0 + tid;
    if (b >= A.B) return;
    const SwdTargetDev tg = A.tg[t];
    LdsLay lay{lds + tid, L};
    double *cws = nullptr, *cbws = nullptr;
    if (tg.mode > 1) {
        cws = A.ws + ((long)t * 2 * BH_NP) * A.B + b;
        cbws = cws + (long)BH_NP * A.B;
    }
    int nl = A.nlay[b];
    nl = nl < 1 ? 1 : (nl > L ? L : nl);
    double *out = A.out + (long)b * A.out_stride + tg.out_off;
    int err = swd_lane(lay, nl, tg, A.periods + tg.per_off, out, cws, cbws, A.B, nullptr);
    A.err[(long)b * A.ntargets + t] = err;
}

// -------------------------------------------------------------------------------------------- RF
__global__ __launch_bounds__(RF_T) void rf_kernel(RfArgs A)
{
    extern __shared__ double S[];
    const RfLaunch &P = A.P;
    const int tid = threadIdx.x;
    const int b0 = blockIdx.x * P.M;
    const int Mb = min(P.M, A.B - b0);
    const int L = P.Lmax;
    const RfLayout lo = rf_layout(L, P.nsamp);
    const int n = P.nsamp;

    // P1: flatten layers
    for (int idx = tid; idx < Mb * L; idx += RF_T) {
        int m = idx / L, i = idx - m * L;
        long b = b0 + m;
        int nl = A.nlay[b];
        nl = nl < 1 ? 1 : (nl > L ? L : nl);
        if (i < nl)
            rf_phase1_layer(S + (long)m * lo.per_model, lo, nl, i, A.h + b * L, A.vp + b * L,
                            A.vs + b * L, A.rho + b * L, A.qp ? A.qp + b * L : nullptr,
                            A.qs ? A.qs + b * L : nullptr, P.depth_input);
    }
    __syncthreads();
    // P2: interface coefficients (+ per-model scalars on interface 0)
    for (int idx = tid; idx < Mb * L; idx += RF_T) {
        int m = idx / L, i = idx - m * L;
        long b = b0 + m;
        int nl = A.nlay[b];
        nl = nl < 1 ? 1 : (nl > L ? L : nl);
        if (i < nl)
            rf_phase2_interface(S + (long)m * lo.per_model, lo, P, nl, i, A.vp[b * L], A.vs[b * L]);
    }
    __syncthreads();
    // P3: (model, frequency) tasks, model-major
    const int ntask = Mb * P.nfreq;
    for (int task = tid; task < ntask; task += RF_T) {
        int m = task / P.nfreq, j = task - m * P.nfreq;
        int nl = A.nlay[b0 + m];
        nl = nl < 1 ? 1 : (nl > L ? L : nl);
        double *Sm = S + (long)m * lo.per_model;
        cd crf = rf_phase3_task(Sm, lo, P, nl, j);
        st_cd(Sm + 2 * j, crf);
    }
    __syncthreads();
    // P4: inverse FFT in LDS
    const int nh = n / 2 - 1;
    for (int idx = tid; idx < Mb * nh; idx += RF_T) {
        int m = idx / nh, i = n / 2 + 1 + (idx - m * nh);
        rf_fft_hermitian(S + (long)m * lo.per_model, n, i);
    }
    __syncthreads();
    for (int idx = tid; idx < Mb * n; idx += RF_T) {
        int m = idx / n, i = idx - m * n;
        rf_fft_bitrev_scale(S + (long)m * lo.per_model, n, P.log2n, P.sc, i);
    }
    __syncthreads();
    for (int l = 1; l < n; l <<= 1) {
        for (int idx = tid; idx < Mb * (n / 2); idx += RF_T) {
            int m = idx / (n / 2), bf = idx - m * (n / 2);
            rf_fft_butterfly(S + (long)m * lo.per_model, A.tw, l, bf);
        }
        __syncthreads();
    }
    for (int idx = tid; idx < Mb * P.nout; idx += RF_T) {
        int m = idx / P.nout, i = idx - m * P.nout;
        A.out[(long)(b0 + m) * P.out_stride + P.out_off + i] = P.qn * S[(long)m * lo.per_model + 2 * i];
    }
}

// ---------------------------------------------------------------------------------------- launch
hipError_t launch_swd(const SwdArgs &A, hipStream_t stream)
{
    size_t lds = (size_t)4 * A.Lmax * SWD_T * sizeof(float);
    static thread_local size_t lds_set = 0;
    if (lds > 48 * 1024 && lds > lds_set) {
        hipError_t e = hipFuncSetAttribute((const void *)swd_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        lds_set = lds;
    }
    dim3 grid((A.B + SWD_T - 1) / SWD_T, A.ntargets);
    hipLaunchKernelGGL(swd_kernel, grid, dim3(SWD_T), lds, stream, A);
    return hipGetLastError();
}

size_t rf_lds_bytes(int Lmax, int nsamp, int M)
{
    RfLayout lo = rf_layout(Lmax, nsamp);
    return (size_t)M * lo.per_model * sizeof(double);
}

hipError_t launch_rf(const RfArgs &A, hipStream_t stream)
{
    size_t lds = rf_lds_bytes(A.P.Lmax, A.P.nsamp, A.P.M);
    static thread_local size_t lds_set = 0;
    if (lds > 48 * 1024 && lds > lds_set) {
        hipError_t e = hipFuncSetAttribute((const void *)rf_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        lds_set = lds;
    }
    dim3 grid((A.B + A.P.M - 1) / A.P.M);
    hipLaunchKernelGGL(rf_kernel, grid, dim3(RF_T), lds, stream, A);
    return hipGetLastError();
}

}  // namespace bh
