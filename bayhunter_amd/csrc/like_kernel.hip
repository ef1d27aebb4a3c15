// like_kernel.hip -- fused misfit + Gaussian log-likelihood for a batch of models.
//
// Reference: JointTarget.evaluate, src/Targets.py:322-347, with the covariance models of
// Valuation (src/Targets.py:105-173).  The reference materialises a dense n x n inverse covariance
// for every evaluation and does two numpy dots; here the diagonal and exponential (tridiagonal)
// models are closed-form single passes and only the fixed Gaussian model multiplies by a dense
// matrix: R^-1 (n x n, fixed for the whole run) is streamed once per workgroup from L2 while LIKE_M
// models' residual vectors sit in LDS, i.e. every matrix element fetched is used LIKE_M times.
//
// One workgroup = LIKE_M models, LIKE_T threads.  HBM traffic: the model's output row is read once
// (it is what swd_kernel/rf_kernel just wrote), 8*(ntargets+2) bytes are written per model.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "kernels.h"

namespace bh {

__device__ __forceinline__ double wave_sum(double v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---------------------------------------------------------------------------------------------
// Dense Gaussian model on the FP64 matrix cores.  Y = D * R^-1 with D = residuals [models x n] and
// the fixed R^-1 [n x n] is the one GEMM-shaped piece of the hot path (2*n^2 flops per model,
// n = 201 for a receiver function); q_m = sum_i Y[m][i] * D[m][i].
// One wave = 16 models.  v_mfma_f64_16x16x4_f64: A = D[16 models][4 k] (lane l: model l&15,
// k = l>>4), B = R^-1[4 k][16 cols] (lane l: k = l>>4, col = l&15), C/D: col = l&15,
// row = (l>>4) + 4*reg.  All ceil(n/16) column tiles are accumulated per k-step, so every
// residual fragment is loaded once; R^-1 (323 KB at n = 201) streams from L2.
// Writes partial q and sum(d^2) per (model, column-tile group) into `gq`; like_kernel adds them up.
typedef double double4_t __attribute__((ext_vector_type(4)));

// One workgroup = 4 waves = 64 models (16 per wave).  R^-1 is streamed through LDS in chunks of GQ_KC rows that
// the four waves share (round 1 had every wave fetch its own fragments from L2 with 256 VGPRs and one wave per
// SIMD: latency-bound at 13 TFLOP/s); the next chunk is loaded into the other buffer while the MFMAs of the
// current one issue.
// The column tiles of a target form GROUPS of GQ_NTG; every (model, group) gets its partial q and sum(d^2) in
// `gq`, and like_kernel adds the partials in group order.  Two decompositions compute the very same partials:
//   SPLIT   a workgroup takes ONE group (blockIdx.y) -- small batches.  Until round 4 one workgroup took all 13
//           tiles of a 201-point receiver function, 64 models at a time: a sampler's batch of 2 048 proposals was
//           32 workgroups on 256 CUs, each streaming the whole 323 KB matrix, on the critical path of every batch
//           (0.074 -> 0.036 ms alone, 0.24 ms beside the other chain group's dispersion searches)
//   fused   a workgroup takes NT tiles per pass (all 13 of a receiver function: residual fragments are loaded once
//           per model instead of once per group) -- large batches (131 072 models: 0.58 against 0.84 ms)
// so a model's likelihood does not depend on the size of the batch it is in.
enum { GQ_KC = 16, GQ_WAVES = 4 };
template <int NT, bool SPLIT>
__global__ __launch_bounds__(64 * GQ_WAVES) __attribute__((amdgpu_waves_per_eu(2, 2))) void gauss_q_kernel(LikeArgs A, int t, double *gq)
{
    extern __shared__ double rbuf[];                  // [2][GQ_KC][NT*16]
    constexpr int NP = NT * 16;
    constexpr int NG = (NT + GQ_NTG - 1) / GQ_NTG;    // groups per pass
    const LikeTargetDev tg = A.tg[t];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long b0 = ((long)blockIdx.x * GQ_WAVES + wave) * 16;
    const int n = tg.n;
    const double *R = A.aux + tg.aux_off;
    const int mrow = lane & 15, kq = lane >> 4;
    const long bm = b0 + mrow;
    const bool mvalid = bm < A.B;
    const double *drow = A.out + (mvalid ? bm : 0) * (long)A.out_stride + tg.off;
    const double *yobs = A.yobs + tg.off;
    const int ntiles = (n + 15) / 16;
    const int nchunk = (n + GQ_KC - 1) / GQ_KC;
    const int tb0 = SPLIT ? (int)blockIdx.y * NT : 0, tb1 = SPLIT ? tb0 + NT : ntiles;
    for (int tb = tb0; tb < tb1; tb += NT) {
        double4_t acc[NT];
#pragma unroll
        for (int i = 0; i < NT; i++) acc[i] = double4_t{0.0, 0.0, 0.0, 0.0};
        const int col0 = tb * 16;
        // chunk c of R^-1 (rows c*KC .., columns col0 .. col0+NP) -> rbuf[c & 1], zero padded
        auto load_chunk = [&](int c) {
            double *dst = rbuf + (c & 1) * (GQ_KC * NP);
#pragma unroll 4
            for (int e = tid; e < GQ_KC * NP; e += 64 * GQ_WAVES) {
                const int r = e / NP, cc = e - r * NP;
                const int k = c * GQ_KC + r, col = col0 + cc;
                dst[e] = (k < n && col < n) ? R[(long)k * n + col] : 0.0;
            }
        };
        load_chunk(0);
        __syncthreads();
        for (int c = 0; c < nchunk; c++) {
            if (c + 1 < nchunk) load_chunk(c + 1);       // other buffer: free since the last barrier
            const double *src = rbuf + (c & 1) * (GQ_KC * NP);
            double av[GQ_KC / 4];
#pragma unroll
            for (int ks = 0; ks < GQ_KC / 4; ks++) {
                const int k = c * GQ_KC + ks * 4 + kq;
                av[ks] = (mvalid && k < n) ? drow[k] - yobs[k] : 0.0;
            }
#pragma unroll 1
            for (int ks = 0; ks < GQ_KC / 4; ks++) {
                const double *row = src + (ks * 4 + kq) * NP + mrow;
#pragma unroll
                for (int i = 0; i < NT; i++)
                    acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ks], row[i * 16], acc[i], 0, 0, 0);
            }
            __syncthreads();
        }
        // lane holds Y[model kq + 4r][col (tb+i)*16 + mrow]; dot with D over the columns of each group
#pragma unroll
        for (int g = 0; g < NG; g++) {
            const int grp = tb / GQ_NTG + g;             // (passes start at a multiple of GQ_NTG tiles)
            if (grp * GQ_NTG >= ntiles) break;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const long bmod = b0 + kq + 4 * r;
                const bool ok = bmod < A.B;
                const double *dr = A.out + (ok ? bmod : 0) * (long)A.out_stride + tg.off;
                double q = 0.0, s2 = 0.0;
#pragma unroll
                for (int i = g * GQ_NTG; i < (g + 1) * GQ_NTG && i < NT; i++) {
                    const int col = (tb + i) * 16 + mrow;
                    const double d = (ok && col < n) ? dr[col] - yobs[col] : 0.0;
                    q += acc[i][r] * d;
                    s2 += d * d;
                }
                for (int o = 8; o > 0; o >>= 1) {       // reduce over the 16 lanes sharing kq
                    q += __shfl_xor(q, o, 64);
                    s2 += __shfl_xor(s2, o, 64);
                }
                if (mrow == 0 && ok) {
                    double *w = gq + (bmod * A.gq_groups + grp) * 2;
                    w[0] = q; w[1] = s2;
                }
            }
        }
    }
}

__global__ __launch_bounds__(LIKE_T) void like_kernel(LikeArgs A)
{
    extern __shared__ double sm[];          // [LIKE_M][nmax] residuals, then [LIKE_M][4] partials
    __shared__ double red[LIKE_M][LIKE_T / 64][2];
    __shared__ double acc_logl[LIKE_M];
    __shared__ double acc_mis[LIKE_M];
    __shared__ int bad[LIKE_M];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int b0 = blockIdx.x * LIKE_M;
    const int Mb = min(LIKE_M, A.B - b0);
    const int NW = LIKE_T / 64;

    if (tid < LIKE_M) {
        acc_logl[tid] = 0.0;
        acc_mis[tid] = 0.0;
        int flag = 0;
        if (tid < Mb)
            for (int f = 0; f < A.nflags; f++) flag |= A.err[(long)(b0 + tid) * A.nflags + f];
        bad[tid] = flag;
    }
    __syncthreads();

    for (int t = 0; t < A.ntargets; t++) {
        const LikeTargetDev tg = A.tg[t];
        const int n = tg.n;
        // residuals d = ymod - yobs of the workgroup's models -> LDS (coalesced row reads)
        for (int idx = tid; idx < Mb * n; idx += LIKE_T) {
            int m = idx / n, i = idx - m * n;
            sm[m * n + i] = A.out[(long)(b0 + m) * A.out_stride + tg.off + i] - A.yobs[tg.off + i];
        }
        __syncthreads();

        if (tg.cov == 3 && A.gq) {
            // precomputed on the matrix cores (gauss_q_kernel)
            if (tid < LIKE_M) {
                for (int w = 0; w < NW; w++) { red[tid][w][0] = 0.0; red[tid][w][1] = 0.0; }
                if (tid < Mb) {                     // the partials of the target's column-tile groups, in group order
                    const double *g = A.gq + ((long)t * A.B + (b0 + tid)) * 2 * A.gq_groups;
                    double qq = 0.0, ss = 0.0;
                    for (int k = 0; k < gq_groups_of(n); k++) { qq += g[2 * k]; ss += g[2 * k + 1]; }
                    red[tid][0][1] = qq;
                    red[tid][0][0] = ss;
                }
            }
        } else if (tg.cov == 3) {
            // q_m = d_m^T R^-1 d_m : thread i owns row i for all LIKE_M models
            const double *R = A.aux + tg.aux_off;
            double s2[LIKE_M], q[LIKE_M];
#pragma unroll
            for (int m = 0; m < LIKE_M; m++) { s2[m] = 0.0; q[m] = 0.0; }
            for (int i = tid; i < n; i += LIKE_T) {
                double rd[LIKE_M];
#pragma unroll
                for (int m = 0; m < LIKE_M; m++) rd[m] = 0.0;
                const double *Ri = R + (long)i * n;
                for (int j = 0; j < n; j++) {
                    double r = Ri[j];
#pragma unroll
                    for (int m = 0; m < LIKE_M; m++) rd[m] += r * sm[m * n + j];
                }
#pragma unroll
                for (int m = 0; m < LIKE_M; m++) {
                    double d = sm[m * n + i];
                    q[m] += d * rd[m];
                    s2[m] += d * d;
                }
            }
#pragma unroll
            for (int m = 0; m < LIKE_M; m++) {
                double a = wave_sum(s2[m]), c = wave_sum(q[m]);
                if (lane == 0) { red[m][wv][0] = a; red[m][wv][1] = c; }
            }
        } else {
            // closed forms: wave w handles models w, w+NW, ...
            for (int m = wv; m < LIKE_M; m += NW) {
                double s2 = 0.0, q = 0.0;
                if (m < Mb) {
                    const double *d = sm + m * n;
                    if (tg.cov == 0) {
                        for (int i = lane; i < n; i += 64) s2 += d[i] * d[i];
                        q = s2;
                    } else if (tg.cov == 1) {
                        const double *se = A.aux + tg.aux_off;
                        for (int i = lane; i < n; i += 64) {
                            double dd = d[i] * d[i];
                            s2 += dd;
                            q += dd / se[i];
                        }
                    } else {   // exponential law: R^-1 tridiagonal, src/Targets.py:130-137
                        double r = A.noise[(long)(b0 + m) * 2 * A.ntargets + 2 * t];
                        for (int i = lane; i < n; i += 64) {
                            double dd = d[i] * d[i];
                            s2 += dd;
                            double diag = (i == 0 || i == n - 1) ? 1.0 : 1.0 + r * r;
                            q += diag * dd;
                            if (i + 1 < n) q -= 2.0 * r * d[i] * d[i + 1];
                        }
                    }
                }
                s2 = wave_sum(s2);
                q = wave_sum(q);
                if (lane == 0) {
                    for (int w = 0; w < NW; w++) { red[m][w][0] = 0.0; red[m][w][1] = 0.0; }
                    red[m][0][0] = s2;
                    red[m][0][1] = q;
                }
            }
        }
        __syncthreads();
        if (tid < Mb) {
            const int m = tid;
            double s2 = 0.0, q = 0.0;
            for (int w = 0; w < NW; w++) { s2 += red[m][w][0]; q += red[m][w][1]; }
            const double corr = A.noise[(long)(b0 + m) * 2 * A.ntargets + 2 * t];
            const double sigma = A.noise[(long)(b0 + m) * 2 * A.ntargets + 2 * t + 1];
            double madist, logdet = (2.0 * n) * log(sigma);
            if (tg.cov == 2) {
                madist = q / (sigma * sigma * (1.0 - corr * corr));
                logdet += (n - 1) * log(1.0 - corr * corr);
            } else {
                madist = q / (sigma * sigma);
                logdet += tg.logdet_extra;
            }
            const double logl_part = -0.5 * (n * log(2.0 * 3.141592653589793) + logdet);
            acc_logl[m] += logl_part - madist / 2.0;
            const double rms = sqrt(s2 / n);
            acc_mis[m] += rms;
            A.misfits[(long)(b0 + m) * (A.ntargets + 1) + t] = bad[m] ? 1e15 : rms;
        }
        __syncthreads();
    }
    if (tid < Mb) {
        const int m = tid;
        if (bad[m]) {   // src/Targets.py:325-328
            A.logL[b0 + m] = -1e15;
            for (int t = 0; t <= A.ntargets; t++) A.misfits[(long)(b0 + m) * (A.ntargets + 1) + t] = 1e15;
        } else {
            A.logL[b0 + m] = acc_logl[m];
            A.misfits[(long)(b0 + m) * (A.ntargets + 1) + A.ntargets] = acc_mis[m];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Voronoi nuclei -> layers (src/Models.py:26-52) + prior checks (src/SingleChain.py:330-392).
// One lane per proposal; every operation is the reference's IEEE operation, so h, vp, rho are
// bit-identical to numpy's.  Byte-bound and tiny: 16*L bytes in, 32*L out per model.
__global__ __launch_bounds__(256) void voronoi_kernel(VoronoiArgs A)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= A.B) return;
    const int L = A.Lmax;
    const ModelPriorsDev &P = A.pri;
    int n = A.nlay[b];
    n = n < 1 ? 1 : (n > L ? L : n);
    const double *vs = A.vs + (long)b * L, *z = A.z + (long)b * L;
    double *mh = A.model + (long)b * 4 * L, *mvp = mh + L, *mvs = mh + 2 * L, *mrho = mh + 3 * L;
    const double vpvs = A.vpvs[b];
    const bool mantle = P.mantle_vs == P.mantle_vs;
    bool ok = true, in_mantle = false;
    const int layermodel = n - 1;
    if (!(layermodel >= P.layers_min && layermodel <= P.layers_max)) ok = false;
    double zprev = 0.0, zsum = 0.0, vsprev = 0.0;
    for (int i = 0; i < n; i++) {
        const double v = vs[i];
        double h = 0.0;
        if (i < n - 1) {
            const double zd = (z[i] + z[i + 1]) / 2.;       // interface midway between nuclei
            h = zd - zprev;
            zprev = zd;
            if (h < P.thickmin) ok = false;
        }
        zsum = zsum + h;                                     // np.cumsum(h)
        if (zsum < P.z_min || zsum > P.z_max) ok = false;
        if (v < P.vs_min || v > P.vs_max) ok = false;
        if (mantle && v >= P.mantle_vs) in_mantle = true;    // from the first mantle layer downwards
        const double vp = in_mantle ? v * P.mantle_vpvs : v * vpvs;
        if (i > 0) {
            if (P.lowvelperc == P.lowvelperc && !((v - (vsprev * (1 - P.lowvelperc))) > 0)) ok = false;
            if (P.highvelperc == P.highvelperc && !(((vsprev * (1 + P.highvelperc)) - v) > 0)) ok = false;
        }
        vsprev = v;
        mh[i] = h; mvp[i] = vp; mvs[i] = v; mrho[i] = vp * 0.32 + 0.77;
    }
    for (int i = n; i < L; i++) { mh[i] = 0.0; mvp[i] = 0.0; mvs[i] = 0.0; mrho[i] = 0.0; }
    A.valid[b] = ok ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------
// Self-test of the shared-reciprocal division (bh_common.h) against the `/` operator.
__device__ __forceinline__ unsigned long long xs64(unsigned long long &s)
{
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    return s;
}
__device__ __forceinline__ double rnd_double(unsigned long long &s, int max_exp)
{
    unsigned long long m = xs64(s) & 0x000fffffffffffffULL;
    int e = (int)(xs64(s) % (unsigned)(2 * max_exp + 1)) - max_exp;
    unsigned long long sign = (xs64(s) & 1ULL) << 63;
    unsigned long long bits = sign | ((unsigned long long)(e + 1023) << 52) | m;
    return __longlong_as_double((long long)bits);
}
__global__ void division_selftest_kernel(long n, unsigned seed, int max_exp, unsigned long long *bad)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    unsigned long long s = 0x9E3779B97F4A7C15ULL ^ ((unsigned long long)seed << 32) ^ (unsigned long long)(i * 2654435761UL + 1);
    unsigned long long nbad = 0;
    for (; i < n; i += stride) {
        const double a = rnd_double(s, max_exp), b = rnd_double(s, max_exp), c = rnd_double(s, max_exp);
        const Recip R = recip_of(b);
        const double q1 = qdiv(a, R), q2 = qdiv(c, R);
        const double r1 = a / b, r2 = c / b;
        if (__double_as_longlong(q1) != __double_as_longlong(r1)) nbad++;
        if (__double_as_longlong(q2) != __double_as_longlong(r2)) nbad++;
        const double sa = fabs(a);
        if (__double_as_longlong(xsqrt(sa)) != __double_as_longlong(sqrt(sa))) nbad++;
    }
    if (i == 0 && (xsqrt(0.0) != 0.0)) nbad++;
    if (nbad) atomicAdd(bad, nbad);
}
hipError_t launch_division_selftest(long n, unsigned seed, int max_exp, unsigned long long *bad, hipStream_t stream)
{
    hipLaunchKernelGGL(division_selftest_kernel, dim3(2048), dim3(256), 0, stream, n, seed, max_exp, bad);
    return hipGetLastError();
}

// Sort key of the dispersion searches' processing order (engine.reorder): one thread per model.
//   bits 24..30  127 - nlay          deepest models first (a wave's layer loop runs to its deepest model)
//   bits 18..23  class of the predicted search length, longest first: a launch ends with lanes waiting
//                for the last searches, so those should be short ones.  A search costs ~11 evaluations
//                per period plus one per 0.005 km/s between its start value (0.855 c_R of the slowest
//                layer) and the phase velocity at the longest period; predictor: a depth-kernel average
//                of vs at that period (weights exp(-z/reach) h, reach = 0.35 * 3.5 km/s * period; the
//                half-space counts with a thickness of one reach) minus 0.79 min(vs), classes of 0.15
//   bits  0..17  S-wave travel time through the stack in 1/256 s: neighbours in a wave should be alike
// Loads: a wave stages the h and vs rows of its 64 models in LDS with lane-consecutive addresses (the rows of a
// model are Lmax doubles long and mstride apart: per-thread loads of one model's row each fetched a 64-byte sector
// for every 8 bytes used -- 1.18 GB for the 84 MB of layer values of 524 288 ten-layer models, round 3).
__global__ void order_key_kernel(int B, int Lmax, int mstride, const int *nlay, const double *h,
                                 const double *vs, float reach, int by_length, int *keys)
{
    extern __shared__ float okl[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int Lp = Lmax | 1;                                // odd row pitch: thread m reads row m without bank conflicts
    float *hl = okl + (long)wave * 2 * 64 * Lp, *vl = hl + 64 * Lp;
    const long b0 = ((long)blockIdx.x * (blockDim.x >> 6) + wave) * 64;
    const int nm = b0 >= B ? 0 : (int)(B - b0 < 64 ? B - b0 : 64);
    {
        int m = lane / Lmax, i = lane - m * Lmax;           // element e = lane + 64 k of the [64][Lmax] tile
        const int dm = 64 / Lmax, di = 64 - dm * Lmax;
        while (m < nm) {
            const long g = (b0 + m) * (long)mstride + i;
            hl[m * Lp + i] = (float)h[g];
            vl[m * Lp + i] = (float)vs[g];
            m += dm; i += di;
            if (i >= Lmax) { i -= Lmax; m++; }
        }
    }
    __syncthreads();                                        // (a wave reads only what it wrote itself)
    const long b = b0 + lane;
    if (lane >= nm) return;
    int nl = nlay[b];
    nl = nl < 1 ? 1 : (nl > Lmax ? Lmax : nl);
    const float *hb = hl + lane * Lp, *vb = vl + lane * Lp;
    float tt = 0.f, z = 0.f, sw = 0.f, svw = 0.f, vmin = 1e9f;
    for (int i = 0; i < nl; i++) {
        const float hi = hb[i], vi = vb[i];
        if (!(vi > 0.f)) { z += hi; continue; }           // water layer: no S wave
        tt += hi / vi;
        const bool hs = i == nl - 1;
        const float zmid = hs ? z + 10.f : z + 0.5f * hi;
        const float w = __expf(-zmid / reach) * (hs ? reach : hi);
        sw += w; svw += vi * w;
        vmin = fminf(vmin, vi);
        z += hi;
    }
    int cls = 0;
    if (by_length && sw > 0.f) {
        const float span = 0.92f * svw / sw - 0.79f * vmin;
        cls = (int)fminf(fmaxf(floorf((6.0f - span) / 0.15f), 0.f), 63.f);
    }
    const int depth = 127 - (nl > 127 ? 127 : nl);
    const int t18 = (int)fminf(fmaxf(tt * 256.f, 0.f), 262143.f);
    keys[b] = (depth << 24) | (cls << 18) | t18;
}

hipError_t launch_order_keys(int B, int Lmax, int mstride, const int *nlay, const double *h, const double *vs,
                             double reach, int by_length, int *keys, hipStream_t stream)
{
    // waves per workgroup: as many as 64 KiB of LDS hold (2 x 64 x (Lmax | 1) floats each), at most four
    const size_t per_wave = (size_t)2 * 64 * (Lmax | 1) * sizeof(float);
    int W = (int)((64 * 1024) / per_wave);
    W = W < 1 ? 1 : W > 4 ? 4 : W;
    const long waves = ((long)B + 63) / 64;
    hipLaunchKernelGGL(order_key_kernel, dim3((unsigned)((waves + W - 1) / W)), dim3(64 * W), per_wave * W, stream, B, Lmax,
                       mstride, nlay, h, vs, (float)reach, by_length, keys);
    return hipGetLastError();
}

hipError_t launch_voronoi(const VoronoiArgs &A, hipStream_t stream)
{
    hipLaunchKernelGGL(voronoi_kernel, dim3((A.B + 255) / 256), dim3(256), 0, stream, A);
    return hipGetLastError();
}

// stages: 1 = the dense Gaussian products (they need only the rows of their own targets: an evaluation plan runs
// them behind the receiver-function kernel on its side stream, beside the dispersion searches), 2 = like_kernel
hipError_t launch_like(const LikeArgs &A, int nmax, hipStream_t stream, int stages)
{
    if (A.gq && (stages & 1)) {   // dense Gaussian targets first, on the FP64 MFMA
        for (int t = 0; t < A.ntargets; t++)
            if (A.tg[t].cov == 3)
            {
                const int nt = (A.tg[t].n + 15) / 16;
                const dim3 b(64 * GQ_WAVES);
                const unsigned gx = (unsigned)((A.B + 16 * GQ_WAVES - 1) / (16 * GQ_WAVES));
                double *gq = A.gq + (long)t * A.B * 2 * A.gq_groups;
                static const char *force = std::getenv("BH_GQ_FORM");        // "split" | "fused" (A/B)
                const bool split = force ? force[0] == 's' : A.B <= 32768;
                if (split) {
                    const size_t lds = (size_t)2 * GQ_KC * GQ_NTG * 16 * sizeof(double);
                    hipLaunchKernelGGL((gauss_q_kernel<GQ_NTG, true>), dim3(gx, gq_groups_of(A.tg[t].n)), b, lds, stream, A, t, gq);
                } else {
                    // a pass must start at a multiple of GQ_NTG tiles: 13 tiles only when one pass takes them all
                    const int NTsel = nt <= 4 ? 4 : nt <= 8 ? 8 : nt <= 13 ? 13 : 12;
                    const size_t lds = (size_t)2 * GQ_KC * NTsel * 16 * sizeof(double);   // <= 64 KiB
                    if (NTsel == 4) hipLaunchKernelGGL((gauss_q_kernel<4, false>), dim3(gx), b, lds, stream, A, t, gq);
                    else if (NTsel == 8) hipLaunchKernelGGL((gauss_q_kernel<8, false>), dim3(gx), b, lds, stream, A, t, gq);
                    else if (NTsel == 13) hipLaunchKernelGGL((gauss_q_kernel<13, false>), dim3(gx), b, lds, stream, A, t, gq);
                    else hipLaunchKernelGGL((gauss_q_kernel<12, false>), dim3(gx), b, lds, stream, A, t, gq);
                }
            }
    }
    if (!(stages & 2)) return hipGetLastError();
    size_t lds = (size_t)LIKE_M * nmax * sizeof(double);
    static size_t lds_set[16] = {0};
    if (lds > 48 * 1024) {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        dev &= 15;
        if (lds > lds_set[dev]) {
            e = hipFuncSetAttribute((const void *)like_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            lds_set[dev] = lds;
        }
    }
    dim3 grid((A.B + LIKE_M - 1) / LIKE_M);
    hipLaunchKernelGGL(like_kernel, grid, dim3(LIKE_T), lds, stream, A);
    return hipGetLastError();
}

}  // namespace bh
