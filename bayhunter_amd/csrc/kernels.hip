// kernels.hip -- gfx950 kernels of libbayhunter_amd (hand-written HIP, wave64).
//
//   swd_kernel : persistent lanes, one (model, dispersion target) search at a time per lane, pulled
//                from a per-target atomic work queue; one wave per workgroup; each lane keeps its
//                current model's fp32 layer stack in its column of a layer-major LDS image
//                ([array][layer][lane], conflict-free); the search itself is swd_core.h (exact
//                replay of surfdisp96.f).
//   rf_kernel  : one workgroup per M models; phases P1..P4 of rf_core.h with __syncthreads between.
//
// Both are fp64 scalar recurrences: bound by FP64 VALU issue + transcendental latency, not by HBM
// and not MFMA-shaped ("Roofline").  HBM traffic is the algorithmic minimum: each model
// byte is read once, each output written once; everything else lives in LDS/registers.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "kernels.h"
#include "rf_core.h"
#include "swd_core.h"
#include "swd_team.h"

namespace bh {

// hipFuncSetAttribute is per device: remember the largest dynamic-LDS size set per (kernel, device)
static hipError_t ensure_dyn_lds(const void *func, size_t lds, size_t (&set)[16])
{
    if (lds <= 48 * 1024) return hipSuccess;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    dev &= 15;
    if (lds > set[dev]) {
        e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        set[dev] = lds;
    }
    return hipSuccess;
}

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

// nlay outside 1..Lmax (a stale depth hint, an nlay array changed behind the engine's back): the
// search is not run on a truncated model; the row is NaN and the flag BH_MODEL_BAD_DEPTH.
__device__ __forceinline__ void swd_bad_depth(const SwdArgs &A, const SwdTargetDev &tg, int t, long b)
{
    double *out = A.out + b * A.out_stride + tg.out_off;
    for (int k = 0; k < tg.nper; k++) out[k] = __builtin_nan("");
    A.err[b * A.ntargets + t] = 2;
}

// Per-target work queue: lanes pull model indices from an atomic counter (zeroed by the launcher).
// A lane that pulls a task copies that model's fp64 row (contiguous, 32*L bytes) into ITS column of
// the layer-major LDS image, rounding to fp32 like f2py does (surf96_modsw.py:68-82).  At kernel
// start all 64 lanes pull together (consecutive indices -> the wave reads one contiguous chunk per
// array); later pulls are single lanes refilling while the rest of the wave keeps searching.
struct QueueSrc {
    const SwdArgs &A;
    const SwdTargetDev &tg;
    unsigned int *counter;
    int t;
    long cur;
    __device__ __forceinline__ int next(LdsLay &lay, double *&out, double *&cws, double *&cbws)
    {
        unsigned int b;
        int nl;
        const int L = A.Lmax;
        for (;;) {
            // (Handing the queue out in chunks of 64 consecutive slots per wave, so that a wave holds
            // neighbours of the processing order, was measured: 5.7 % slower -- slots reserved by a
            // wave whose lanes are still busy wait while other waves idle at the end.  DESIGN section 4.1.)
            // (Taking the slot a dozen evaluations before the search ends and walking these three dependent
            // steps one per loop trip -- no wait for the atomic, the order and the depth -- was tried:
            // no difference, 49.8 vs 49.2 ms on different boxes, driver share 6.6 % either way.)
            b = atomicAdd(counter, 1u);
            if (b >= (unsigned int)A.B) return 0;
            if (A.order) b = (unsigned int)A.order[b];
            nl = A.nlay[b];
            if (nl >= 1 && nl <= L) break;
            swd_bad_depth(A, tg, t, b);       // not a model of this batch's depth: flagged, never truncated
        }
        cur = b;
        const long g = (long)b * A.mstride;
        if (A.vec2) {   // rows 16-byte aligned: two layers per load (half the memory requests)
            // All loads of a chunk (12 layers of two arrays) are issued before the first value is used: a
            // fetch stalls the whole wave for the memory latency of these rows -- they come from HBM, every
            // model is read once -- and the plain loop (4 loads, wait, store, next pair of layers) paid it
            // once per pair of layers: 5 round trips for 10 layers, now 2.
            constexpr int CH = 6;
            for (int l0 = 0; l0 < nl; l0 += 2 * CH) {
                double2 p[CH], q[CH];
#pragma unroll
                for (int u = 0; u < CH; u++)
                    if (l0 + 2 * u < nl) {
                        p[u] = *(const double2 *)(A.h + g + l0 + 2 * u);
                        q[u] = *(const double2 *)(A.vp + g + l0 + 2 * u);
                    }
#pragma unroll
                for (int u = 0; u < CH; u++) {
                    const int l = l0 + 2 * u;
                    if (l < nl) { lay.set_d(l, (float)p[u].x); lay.set_a(l, (float)q[u].x); }
                    if (l + 1 < nl) { lay.set_d(l + 1, (float)p[u].y); lay.set_a(l + 1, (float)q[u].y); }
                }
#pragma unroll
                for (int u = 0; u < CH; u++)
                    if (l0 + 2 * u < nl) {
                        p[u] = *(const double2 *)(A.vs + g + l0 + 2 * u);
                        q[u] = *(const double2 *)(A.rho + g + l0 + 2 * u);
                    }
#pragma unroll
                for (int u = 0; u < CH; u++) {
                    const int l = l0 + 2 * u;
                    if (l < nl) { lay.set_b(l, (float)p[u].x); lay.set_rho(l, (float)q[u].x); }
                    if (l + 1 < nl) { lay.set_b(l + 1, (float)p[u].y); lay.set_rho(l + 1, (float)q[u].y); }
                }
            }
        } else {
            for (int l = 0; l < nl; l++) {
                lay.set_d(l, (float)A.h[g + l]);
                lay.set_a(l, (float)A.vp[g + l]);
                lay.set_b(l, (float)A.vs[g + l]);
                lay.set_rho(l, (float)A.rho[g + l]);
            }
        }
        out = A.out + (long)b * A.out_stride + tg.out_off;
        if (tg.mode > 1) {
            cws = A.ws + ((long)t * 2 * BH_NP) * A.B + b;
            cbws = cws + (long)BH_NP * A.B;
        }
        return nl;
    }
    // Results are real*4 values (surfdisp96.f:298-310).  With room in LDS a search's values wait in
    // this lane's column of a [period][lane] fp32 image and the row is written when the search is
    // over: back-to-back 16-byte stores of one 8*nper-byte run, instead of one scattered store every
    // other period -- which the memory side tallies at 32-64 bytes each (2.7x the row, round 1).
    float *stage;                 // lds + lane, or nullptr: store as the periods converge
    __device__ __forceinline__ void put(SwdState &S, int k, int kmax, float v)
    {
        if (stage) stage[(k - 1) * SWD_T] = v;
        else swd_put_direct(S, k, kmax, v);
    }
    __device__ __forceinline__ void fill_zero(SwdState &S, int k, int kmax)
    {
        if (stage) { for (int i = k; i <= kmax; i++) stage[(i - 1) * SWD_T] = 0.f; }
        else swd_zero_direct(S, k, kmax);
    }
    __device__ __forceinline__ void done(int err)
    {
        A.err[cur * A.ntargets + t] = err;
        if (stage) {
            const int n = tg.nper;
            double *o = A.out + cur * A.out_stride + tg.out_off;
            int k = 0;
            if (((unsigned long long)o & 15) && n > 0) { o[0] = (double)stage[0]; k = 1; }   // to a 16-byte boundary
            for (; k + 1 < n; k += 2) swd_store_pair(o + k, stage[k * SWD_T], stage[(k + 1) * SWD_T]);
            if (k < n) o[k] = (double)stage[k * SWD_T];
        }
    }
    __device__ __forceinline__ void sphere(LdsLay &lay, int mmax, int ifunc) { swd_sphere(lay, mmax, ifunc); }
};

// 2 waves per SIMD: the search state + one Dunkin layer need 189 VGPRs (250 while polynomial coefficients
// were kept in VGPR pairs, bh_math.h); the allocator's budget is pinned to two waves
#ifndef BH_SWD_WAVES
#define BH_SWD_WAVES 2            // waves per SIMD the register budget of swd_kernel is set for
#endif
// Diagnostic build only (-DBH_LANE_PROFILE, tools/lane_phase_profile.py): shader-clock cycles of the three
// parts of swd_lane's loop -- driver (events, task fetch), period equation, control -- summed over all
// lanes (a lane also counts the cycles it sits masked off while other lanes of its wave run their
// control code); read back with bh_debug_lane_profile.
#if defined(BH_LANE_PROFILE)
__device__ unsigned long long g_lane_prof[8];
template <class Lay, class Src>
__device__ __forceinline__ void swd_lane_prof(Lay &lay, Src &src, const SwdTargetDev &tg, const double *BH_RESTRICT per, int wss)
{
    SwdState S;
    swd_state_init(S);
    NevRegs nv;
    swd_nev_init(nv);
    nv.cycles = 0;
    unsigned long long td = 0, te = 0, tc = 0, n = 0, t0 = clock64(), t1;
    for (;;) {
        swd_events(S, lay, src, tg, per, wss);
        t1 = clock64(); td += t1 - t0; t0 = t1;
        if (S.st == SWD_ST_DONE) break;
        const double wvno = S.omega / S.ceval;
        const double del = (tg.iwave == 1) ? swd_dltar1(lay, S.mmax, S.llw, wvno, S.omega)
                                           : swd_dltar4(lay, S.mmax, S.llw, wvno, S.omega);
        t1 = clock64(); te += t1 - t0; t0 = t1;
        swd_control(S, del, nv);
        t1 = clock64(); tc += t1 - t0; t0 = t1;
        n++;
    }
    atomicAdd(&g_lane_prof[0], td);
    atomicAdd(&g_lane_prof[1], te);
    atomicAdd(&g_lane_prof[2], tc);
    atomicAdd(&g_lane_prof[3], n);
    atomicAdd(&g_lane_prof[4], nv.cycles);
}
extern "C" int bh_debug_lane_profile(unsigned long long *out, int reset)
{
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lane_prof), sizeof(g_lane_prof)) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[8] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_lane_prof), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}
#endif
__global__ __launch_bounds__(SWD_T) __attribute__((amdgpu_waves_per_eu(BH_SWD_WAVES, BH_SWD_WAVES))) void swd_kernel(SwdArgs A)
{
    extern __shared__ float lds[];
    const int t = blockIdx.y;
    if (!((A.tmask >> t) & 1u)) return;              // not a target of this launch (kernels.h)
    const SwdTargetDev tg = A.tg[t];
    LdsLay lay{lds + threadIdx.x, A.Lmax};
    QueueSrc src{A, tg, A.counters + t, t, 0, A.stage ? lds + 4 * A.Lmax * SWD_T + threadIdx.x : nullptr};
    // the target's periods in LDS, behind the model and result images: a lane reads one whenever it
    // starts a period, and in most loop trips some lane of the wave does
    double *perl = (double *)(lds + (4 * A.Lmax + A.stage) * SWD_T);
    for (int k = threadIdx.x; k < tg.nper; k += SWD_T) perl[k] = A.periods[tg.per_off + k];
    __syncthreads();
#if defined(BH_LANE_PROFILE)
    swd_lane_prof(lay, src, tg, perl, A.B);
#else
    swd_lane(lay, src, tg, perl, A.B, nullptr);
#endif
}

// ---------------------------------------------------------------------------------- SWD, team form
// One wave per (model, target): swd_team.h.  The model's fp32 layer stack is shared by the wave
// ([array][layer] in LDS); every lane runs the (wave-uniform) driver/control code on its own copy of
// the search state, so the only communication is the per-round matrix/trial/value exchange in LDS.
struct TeamLay {
    float *base;
    int L;
    __device__ __forceinline__ float d(int i) const { return base[i]; }
    __device__ __forceinline__ float a(int i) const { return base[L + i]; }
    __device__ __forceinline__ float b(int i) const { return base[2 * L + i]; }
    __device__ __forceinline__ float rho(int i) const { return base[3 * L + i]; }
    __device__ __forceinline__ void set_d(int i, float v) { base[i] = v; }
    __device__ __forceinline__ void set_a(int i, float v) { base[L + i] = v; }
    __device__ __forceinline__ void set_b(int i, float v) { base[2 * L + i] = v; }
    __device__ __forceinline__ void set_rho(int i, float v) { base[3 * L + i] = v; }
};

struct TeamSrc {
    const SwdArgs &A;
    const SwdTargetDev &tg;
    int t, lane, nlanes, taken;
    long b;
    unsigned int *queue;     // nullptr: this team runs search `b` only; else teams pull searches
                             // from the per-target counter until it runs past the batch
    float *res;              // wide teams: the search's results wait here (LDS, [nper]) and are written
                             // as one coalesced row at the end -- no global store inside the loop, so
                             // the barriers of a round never wait for one; nullptr: store directly
    __device__ __forceinline__ int next(TeamLay &lay, double *&out, double *&cws, double *&cbws)
    {
        const int L = A.Lmax;
        int nl;
        for (;;) {
            if (queue) {
                int nb = 0;
                if (lane == 0) nb = (int)atomicAdd(queue, 1u);
                nb = __shfl(nb, (int)(threadIdx.x & 63) - lane, 64);      // the team's lane 0
                if (nb >= A.B) return 0;
                b = A.order ? A.order[nb] : nb;
            } else {
                if (taken) return 0;
                taken = 1;
                if (A.order) b = A.order[b];
            }
            nl = A.nlay[b];
            if (nl >= 1 && nl <= L) break;
            if (lane == 0) swd_bad_depth(A, tg, t, b);
        }
        const long g = b * A.mstride;
        for (int l = lane; l < nl; l += nlanes) {
            lay.set_d(l, (float)A.h[g + l]);
            lay.set_a(l, (float)A.vp[g + l]);
            lay.set_b(l, (float)A.vs[g + l]);
            lay.set_rho(l, (float)A.rho[g + l]);
        }
        __syncthreads();
        out = A.out + b * A.out_stride + tg.out_off;
        if (tg.mode > 1) {
            cws = A.ws + ((long)t * 2 * BH_NP) * A.B + b;
            cbws = cws + (long)BH_NP * A.B;
        }
        return nl;
    }
    __device__ __forceinline__ void done(int err)
    {
        if (lane == 0) A.err[b * A.ntargets + t] = err;
        if (res && lane < tg.nper)           // (lane 0 wrote res[]: same wave, program order)
            A.out[b * A.out_stride + tg.out_off + lane] = (double)res[lane];
    }
    // every lane of the team runs the driver: one of them stores
    __device__ __forceinline__ void put(SwdState &S, int k, int kmax, float v)
    {
        if (res) { if (lane == 0) res[k - 1] = v; return; }
        if (lane == 0) swd_put_direct(S, k, kmax, v);
        else if ((k & 1) && k != kmax) S.pend = v;      // (keep the state identical on all lanes)
    }
    __device__ __forceinline__ void fill_zero(SwdState &S, int k, int kmax)
    {
        if (res) {
            if (lane == 0)
                for (int i = k; i <= kmax; i++) res[i - 1] = 0.f;
            return;
        }
        if (lane == 0) swd_zero_direct(S, k, kmax);
    }
    // The team shares one copy of the model: the lanes of a wave transform it in lock step (every
    // lane reads a value before any lane writes it), a second wave must not transform it again.
    __device__ __forceinline__ void sphere(TeamLay &lay, int mmax, int ifunc)
    {
        if (threadIdx.x < SWD_T) swd_sphere(lay, mmax, ifunc);
        __syncthreads();
    }
};

// TEAM lanes per search, 64/TEAM searches per wave (workgroup = one wave).  TEAM = 64 is the lowest
// latency (most speculation); narrower teams waste fewer lanes in the sequential refinement rounds
// (1 trial x (L-1) layer matrices) and on shallow models, i.e. more searches per second.  The teams
// of a wave run the same phases between the barriers; a team whose search is over pulls the next
// one from the per-target work queue (so ragged batches do not leave teams idle next to a deep
// model) and idles only when the queue is empty.  (__syncthreads in a one-wave workgroup is an LDS
// fence, also under divergence.)
// Diagnostic build only (-DBH_TEAM_PROFILE, tools/team_phase_profile.py): shader-clock cycles per
// phase of the 64-lane team loop, summed over all searches; read back with bh_debug_team_profile.
#if defined(BH_TEAM_PROFILE)
__device__ unsigned long long g_team_prof[20];
#define BH_TP_DECL unsigned long long tp_[18] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tp_t0_ = clock64(), tp_t1_
#define BH_TP(i) (tp_t1_ = clock64(), tp_[i] += tp_t1_ - tp_t0_, tp_t0_ = tp_t1_)
#define BH_TP_COUNT(i, n) (tp_[i] += (n))
#define BH_TP_FLUSH(rounds)                                                                  \
    if (threadIdx.x == 0) {                                                                  \
        for (int i_ = 0; i_ < 18; i_++) atomicAdd(&g_team_prof[i_], tp_[i_]);                \
        atomicAdd(&g_team_prof[18], (unsigned long long)(rounds));                           \
        atomicAdd(&g_team_prof[19], 1ull);                                                   \
    }
#else
#define BH_TP_DECL
#define BH_TP(i)
#define BH_TP_COUNT(i, n)
#define BH_TP_FLUSH(rounds)
#endif

template <int TEAM>
__device__ __forceinline__ void swd_team_body(const SwdArgs &A)
{
    extern __shared__ double tlds[];
    constexpr int NSUB = SWD_T / TEAM;
    static_assert(NSUB > 1, "64 lanes and more per search: swd_teamw_body");
    const int sub = threadIdx.x / TEAM, lane = threadIdx.x % TEAM;
    const int t = blockIdx.y;
    if (!((A.tmask >> t) & 1u)) return;              // not a target of this launch (kernels.h)
    const SwdTargetDev tg = A.tg[t];
    const int nm = A.Lmax > TEAM ? A.Lmax : TEAM;
    // per team: mats[nm][19], trials[16], dels[16] (doubles), then 4*Lmax floats (padded to doubles)
    const int per_team = nm * SWD_NCA + 2 * SWD_TEAM_NT + (4 * A.Lmax + 1) / 2;
    double *mats = tlds + (long)sub * per_team, *trials = mats + (long)nm * SWD_NCA, *dels = trials + SWD_TEAM_NT;
    TeamLay lay{(float *)(dels + SWD_TEAM_NT), A.Lmax};
    TeamSrc src{A, tg, t, lane, TEAM, 0, (long)blockIdx.x, A.counters + t, nullptr};
    const double *per = A.periods + tg.per_off;
    SwdState S;
    swd_state_init(S);
    NevRegs nv;
    swd_nev_init(nv);
    {
        // A team whose search is over keeps walking through the phases with nt = 0 (no trial, no
        // matrix, no value consumed): cheaper than predicating every phase on a per-team flag.
        bool live = true;
        for (;;) {
            if (live) {
                swd_events(S, lay, src, tg, per, A.B);
                live = S.st != SWD_ST_DONE;
            }
            if (!__any(live)) break;
            const int nt = live ? swd_team_plan(S, TEAM, trials) : 0;
            __syncthreads();
            swd_team_assemble(lay, lane, TEAM, tg.iwave, S, nt, trials, mats);
            __syncthreads();
            if (tg.iwave == 2 && 8 * nt <= TEAM) swd_team_chain_ray5(lay, lane, S, nt, trials, mats, dels);
            else swd_team_chain(lay, lane, tg.iwave, S, nt, trials, mats, dels);
            __syncthreads();
            swd_team_consume(S, nv, nt, trials, dels);
            __syncthreads();
        }
    }
}

// The narrower forms carry per-team liveness through the divergent driver call: ~199 registers, no
// scratch, pinned to 2 waves per SIMD (with machine LICM on they wanted ~310 and spilled 48 bytes).
// (the kernels swd_team32 / 16 / 8 are defined behind swd_tpl_body, further down)

// ---------------------------------------------------------------------------------- SWD, wide teams
// 64*W lanes (W waves, one workgroup) per search: swd_team.h, "Wide teams".  Per round
//   plan      control wave (wave 0): slot layout (swd_teamw_round), lane j computes slot j's (c, omega),
//             slots and a {done, slots, mmax, llw} header -> LDS                        | barrier
//   assemble  all waves, lane (j, r): matrix of layer r at trial j -> LDS, column-major | barrier
//   chain     Rayleigh: quad q of wave w propagates trial 16 w + q (component i on lane i, the fifth
//             on all four; max-reduce and re-gather by DPP quad permutes); Love: lane j   | barrier
//   consume   control wave: lane j evaluates tree node j (swd_teamw_node), then values are matched by
//             (omega, c) through ballots and fed to swd_control / swd_events
template <int CTRL>
__device__ __forceinline__ double dpp_quad(double x)
{
    // (all four lanes of every quad are active wherever this is used)
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_d(double x, int j)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), j);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), j);
    return __hiloint2double(hi, lo);
}

// The search state of a wide team is the same in every lane.  Telling the compiler so (a value read
// from the first lane is uniform by construction) turns the branches of the control code into scalar
// branches: no exec-mask bookkeeping, no saved masks to spill.
__device__ __forceinline__ double uni(double x)
{
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)),
                            __builtin_amdgcn_readfirstlane(__double2loint(x)));
}
__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ float uni(float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); }
template <class T>
__device__ __forceinline__ T *uni(T *p)
{
    const unsigned long long v = (unsigned long long)p;
    return (T *)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
                 (unsigned)__builtin_amdgcn_readfirstlane((int)v));
}
struct LdsSlots {                  // period-equation values and trial velocities of the round's slots (LDS)
    const double *d, *tc;
    __device__ __forceinline__ double del(int j) const { return d[j]; }
    __device__ __forceinline__ double c(int j) const { return tc[j]; }
};
struct TeamwVals {
    double mc, mom, dl;                   // slot (lane & 63) of the round and its value
    int nt;
    TeamwNode nd;                         // tree node (lane & 63): arrival bracket (swd_teamw_node)
    unsigned long long left, right;       // nodes whose decision is SWD_GO_LEFT / SWD_GO_RIGHT
#if defined(BH_TEAM_PROFILE)
    unsigned long long *tp, *t0;
    __device__ __forceinline__ void probe(int i) const { const unsigned long long t = clock64(); tp[i] += t - *t0; *t0 = t; }
    __device__ __forceinline__ void count(int i, int n) const { tp[i] += n; }
#else
    __device__ __forceinline__ void probe(int) const {}
    __device__ __forceinline__ void count(int, int) const {}
#endif
    __device__ __forceinline__ int find(double om, double c) const
    {
        const unsigned long long m = __ballot(mc == c && mom == om && (int)(threadIdx.x & 63) < nt);
        return m ? __ffsll((long long)m) - 1 : -1;
    }
    __device__ __forceinline__ double del(int j) const { return readlane_d(dl, j); }
    __device__ __forceinline__ double c(int j) const { return readlane_d(mc, j); }
    __device__ __forceinline__ int go(const SwdState &, int j) const
    {
        return ((right >> j) & 1) ? SWD_GO_RIGHT : ((left >> j) & 1) ? SWD_GO_LEFT : SWD_GO_STOP;
    }
    // chain nodes first, first + 2, ... (at most max) in a row whose decision is `dir`: the ballots, shifted, with
    // the chain's nodes on the even bits
    __device__ __forceinline__ int chain_run(const SwdState &, int first, int dir, int max) const
    {
        const unsigned long long m = (dir == SWD_GO_RIGHT ? right : left) >> first;
        const unsigned long long stop = ~m & 0x5555555555555555ull;
        const int n = stop ? (__ffsll((long long)stop) - 1) >> 1 : 32;
        return n < max ? n : max;
    }
    __device__ __forceinline__ TeamwNode node(const SwdState &, int j) const
    {
        TeamwNode a;
        a.go = SWD_GO_STOP;
        a.out = __builtin_amdgcn_readlane(nd.out, j);
        a.c1 = readlane_d(nd.c1, j); a.d1 = readlane_d(nd.d1, j);
        a.c2 = readlane_d(nd.c2, j); a.d2 = readlane_d(nd.d2, j);
        a.c3n = readlane_d(nd.c3n, j);
        return a;
    }
    // slots first, first + stride, ... (at most count) in a row that are valid and whose value has sign
    // bit `neg`
    __device__ __forceinline__ int run(int first, int stride, int count, bool neg) const
    {
        const bool same = (mc == mc) && ((__double2hiint(dl) < 0) == neg) && (int)(threadIdx.x & 63) < nt;
        unsigned long long m = __ballot(same) >> first;
        int n;
        if (stride == 2) {
            const unsigned long long stop = ~m & 0x5555555555555555ull;     // scan slots are the even ones
            n = stop ? (__ffsll((long long)stop) - 1) >> 1 : 32;
        } else {
            const unsigned long long stop = ~m;
            n = stop ? __ffsll((long long)stop) - 1 : 64;
        }
        return n < count ? n : count;
    }
};

// The same interface for a team of K < 64 lanes that shares its wave with other teams (swd_tpl_body): the teams of
// a wave are at different places of the consuming loop, so nothing here may assume that the whole wave executes it
// -- values and node outcomes are exchanged through the team's LDS block, and a ballot is cut down to the team's
// lanes (all of which execute it together: they carry identical copies of the search state).
template <int K>
struct SubVals {
    double mc, mom, dl;                   // slot tl of the round and its value
    int nt, tl, base;                     // base: the team's first lane in the wave
    const double *tc, *dls, *ndc;         // team LDS: trial velocities, values, node brackets [5][K]
    const int *nout;                      // node outcomes [K]
    unsigned long long left, right;       // nodes whose decision is SWD_GO_LEFT / SWD_GO_RIGHT (bit j = node j)
    static constexpr unsigned long long MASK = (1ull << K) - 1;
    __device__ __forceinline__ void probe(int) const {}
    __device__ __forceinline__ void count(int, int) const {}
    __device__ __forceinline__ unsigned long long mine(bool p) const { return (__ballot(p) >> base) & MASK; }
    __device__ __forceinline__ int find(double om, double c) const
    {
        const unsigned long long m = mine(mc == c && mom == om && tl < nt);
        return m ? __ffsll((long long)m) - 1 : -1;
    }
    __device__ __forceinline__ double del(int j) const { return dls[j]; }
    __device__ __forceinline__ double c(int j) const { return tc[j]; }
    __device__ __forceinline__ int go(const SwdState &, int j) const
    {
        return ((right >> j) & 1) ? SWD_GO_RIGHT : ((left >> j) & 1) ? SWD_GO_LEFT : SWD_GO_STOP;
    }
    __device__ __forceinline__ int chain_run(const SwdState &, int first, int dir, int max) const
    {
        const unsigned long long m = (dir == SWD_GO_RIGHT ? right : left) >> first;
        const unsigned long long stop = ~m & 0x5555555555555555ull;
        const int n = stop ? (__ffsll((long long)stop) - 1) >> 1 : 32;
        return n < max ? n : max;
    }
    __device__ __forceinline__ TeamwNode node(const SwdState &, int j) const
    {
        TeamwNode a;
        a.go = SWD_GO_STOP;
        a.out = nout[j];
        a.c1 = ndc[j]; a.d1 = ndc[K + j]; a.c2 = ndc[2 * K + j]; a.d2 = ndc[3 * K + j]; a.c3n = ndc[4 * K + j];
        return a;
    }
    __device__ __forceinline__ int run(int first, int stride, int count, bool neg) const
    {
        const unsigned long long m = mine((mc == mc) && ((__double2hiint(dl) < 0) == neg) && tl < nt) >> first;
        int n;
        if (stride == 2) {
            const unsigned long long stop = ~m & 0x5555555555555555ull;
            n = stop ? (__ffsll((long long)stop) - 1) >> 1 : 32;
        } else {
            const unsigned long long stop = ~m;
            n = stop ? __ffsll((long long)stop) - 1 : 64;
        }
        return n < count ? n : count;
    }
};

// Rayleigh chain of trial (qc, qom) on the calling lane's quad; matrices of its layers in slots slot0 .. of mats.
// Operation for operation swd_dunkin_apply (ee = e * ca in the reference's order, normc; the
// max-abs of normc skips NaN entries exactly like v_max_f64 does).
struct QuadCols {                 // column i (this lane's component) and column 4 of one layer matrix
    double a0, a1, a2, a3, a4, b0, b1, b2, b3, b4;
};
// Column i of the 5x5 out of the 19 stored values (swd_team.h: SWD_MAT): rows 1-3 of columns 0..3 sit at
// i, 5 + i, 9 + i; rows 4 and 5 at k4, k5 (a lane's constants for the whole chain); column 4 is c15 c14 c35 c12 c11.
struct QuadIdx {
    int i, k4, k5;
};
__device__ __forceinline__ QuadIdx quad_idx(int i)
{
    QuadIdx x;
    x.i = i;
    x.k4 = i == 3 ? 6 : 14 + i;                       // c41 c42 c43 c22
    x.k5 = i == 0 ? 17 : i == 1 ? 14 : i == 2 ? 18 : 5;   // c51 c41 c53 c21
    return x;
}
__device__ __forceinline__ void quad_load(QuadCols &q, const double *p, const QuadIdx &x)
{
    q.a0 = p[x.i]; q.a1 = p[5 + x.i]; q.a2 = p[9 + x.i]; q.a3 = p[x.k4]; q.a4 = p[x.k5];
    q.b0 = p[4]; q.b1 = p[3]; q.b2 = p[13]; q.b3 = p[1]; q.b4 = p[0];
}
__device__ __forceinline__ void quad_apply(double e[5], const QuadCols &q)
{
    const double eeA = ((((0.0 + e[0] * q.a0) + e[1] * q.a1) + e[2] * q.a2) + e[3] * q.a3) + e[4] * q.a4;
    const double eeB = ((((0.0 + e[0] * q.b0) + e[1] * q.b1) + e[2] * q.b2) + e[3] * q.b3) + e[4] * q.b4;
    double t1 = fabs(eeA);
    t1 = __builtin_fmax(t1, dpp_quad<0xB1>(t1));      // lanes 0<->1, 2<->3
    t1 = __builtin_fmax(t1, dpp_quad<0x4E>(t1));      // lanes 0<->2, 1<->3
    t1 = __builtin_fmax(t1, fabs(eeB));
    if (t1 < 1.e-40) t1 = 1.0;
    const Recip by_t1 = recip_of(t1);
    const double enA = qdiv(eeA, by_t1);
    e[4] = qdiv(eeB, by_t1);
    e[0] = dpp_quad<0x00>(enA);
    e[1] = dpp_quad<0x55>(enA);
    e[2] = dpp_quad<0xAA>(enA);
    e[3] = dpp_quad<0xFF>(enA);
}
// one layer's values held in registers (whatever index is asked for)
struct OneLay {
    float dd, aa, bb, rr;
    __device__ __forceinline__ float d(int) const { return dd; }
    __device__ __forceinline__ float a(int) const { return aa; }
    __device__ __forceinline__ float b(int) const { return bb; }
    __device__ __forceinline__ float rho(int) const { return rr; }
};
template <class Lay>
__device__ __forceinline__ double swd_teamw_chain_quad(const Lay &lay, const OneLay &half, const SwdState &S,
                                                       double qc, double qom, const double *mats, int slot0)
{
    const QuadIdx x = quad_idx(threadIdx.x & 3);
    const int nlm = S.mmax - S.llw;
    const double wvno = qom / qc;
    double omega = qom;
    if (omega < 1.0e-4) omega = 1.0e-4;
    double e[5];
    swd_ray_halfspace(half, S.mmax, wvno, wvno * wvno, omega, e);
    // two layers per trip, their columns loaded one layer ahead into alternating register sets
    QuadCols qa, qb;
    int r = nlm;
    if (r > 0) quad_load(qa, mats + swd_mat_off(slot0 + r - 1), x);
    while (r >= 2) {
        quad_load(qb, mats + swd_mat_off(slot0 + r - 2), x);
        quad_apply(e, qa);
        if (r > 2) quad_load(qa, mats + swd_mat_off(slot0 + r - 3), x);
        quad_apply(e, qb);
        r -= 2;
    }
    if (r == 1) quad_apply(e, qa);
    return (S.llw != 1) ? swd_ray_water(lay, wvno, omega, e) : e[0];
}

// Task source of a wide team: one search per workgroup, its model already in LDS (loaded by all waves
// before the control wave starts the driver).  Only the control wave uses it.
struct WideSrc {
    const SwdArgs &A;
    const SwdTargetDev &tg;
    int t, lane, nl, taken;
    long b;
    float *res;              // the search's results wait here (LDS, [nper]) and are written as one row at
                             // the end: no global store inside the loop for a barrier to wait for
    __device__ __forceinline__ int next(TeamLay &, double *&out, double *&cws, double *&cbws)
    {
        if (taken) return 0;
        taken = 1;
        out = A.out + b * A.out_stride + tg.out_off;
        if (tg.mode > 1) {
            cws = A.ws + ((long)t * 2 * BH_NP) * A.B + b;
            cbws = cws + (long)BH_NP * A.B;
        }
        return nl;
    }
    __device__ __forceinline__ void done(int err)
    {
        if (lane == 0) A.err[b * A.ntargets + t] = err;
        if (lane < tg.nper)                  // (lane 0 wrote res[]: same wave, program order)
            A.out[b * A.out_stride + tg.out_off + lane] = (double)res[lane];
    }
    __device__ __forceinline__ void put(SwdState &, int k, int, float v) { if (lane == 0) res[k - 1] = v; }
    __device__ __forceinline__ void fill_zero(SwdState &, int k, int kmax)
    {
        if (lane == 0)
            for (int i = k; i <= kmax; i++) res[i - 1] = 0.f;
    }
    // the lanes of the control wave transform the shared copy in lock step (every lane reads a value
    // before any lane writes it); the other waves are waiting at the round's first barrier
    __device__ __forceinline__ void sphere(TeamLay &lay, int mmax, int ifunc) { swd_sphere(lay, mmax, ifunc); }
};

// Wave 0 is the control wave: it alone carries the search (driver, plan, matching, swd_control); the
// other waves are workers that assemble and chain what the plan in LDS tells them and sleep at the
// barrier while wave 0 thinks.  (Until round 2 every wave ran the control code redundantly: no
// broadcast, but with two waves per SIMD -- a thousand searches of two waves each -- the redundant
// control of one team competed with the arithmetic of another.)
template <int W>
__device__ __forceinline__ void swd_teamw_body(const SwdArgs &A)
{
    extern __shared__ double tlds[];
    constexpr int NL = SWD_T * W;
    // The one-wave team evaluates ONE TRIAL PER LANE -- 64 slots a round, each lane the whole period equation of its
    // trial (swd_dltar4 / swd_dltar1), no matrices in LDS -- like the narrow teams of swd_tpl_body, with the wave's
    // uniform control code.  (Until round 4 it spread the layers of 64 / nlm trials over its lanes like the wider
    // teams; -DBH_TEAM64_LAYERS.)
#if defined(BH_TEAM64_LAYERS)
    constexpr bool TPL = false;
#else
    constexpr bool TPL = W == 1;
#endif
    const int lane = threadIdx.x, wl = lane & 63, wave = uni(lane >> 6);
    const bool ctl = W == 1 || wave == 0;
    const int t = blockIdx.y;
    if (!((A.tmask >> t) & 1u)) return;              // not a target of this launch (kernels.h)
    const SwdTargetDev tg = A.tg[t];
    const int nm = TPL ? 0 : (A.Lmax > NL ? A.Lmax : NL);                 // (no matrix slots: one trial per lane)
    double *mats = tlds, *dels = mats + swd_mat_off(nm), *perl = dels + SWD_TEAMW_NT, *nevt = perl + BH_NP;
    double *tcl = nevt + 24 * W, *toml = tcl + SWD_TEAMW_NT;
    float *res = (float *)(toml + SWD_TEAMW_NT);                 // [BH_NP] staged results
    int *hdr = (int *)(nevt + 24);                               // W > 1: {command, nt, mmax, llw} of the round
    TeamLay lay{res + BH_NP, A.Lmax};
    NevMem nv{nevt, nevt + 12};
    // the team's one search: its model into LDS, by all waves
    long b = blockIdx.x;
    if (A.order) b = A.order[b];
    const int nl = A.nlay[b];
    if (nl < 1 || nl > A.Lmax) {
        if (lane == 0) swd_bad_depth(A, tg, t, b);
        return;
    }
    {
        const long g = b * A.mstride;
        for (int l = lane; l < nl; l += NL) {
            lay.set_d(l, (float)A.h[g + l]);
            lay.set_a(l, (float)A.vp[g + l]);
            lay.set_b(l, (float)A.vs[g + l]);
            lay.set_rho(l, (float)A.rho[g + l]);
        }
        for (int k = lane; k < tg.nper; k += NL) perl[k] = A.periods[tg.per_off + k];
    }
    __syncthreads();
    WideSrc src{A, tg, t, lane, nl, 0, b, res};
    SwdState S;
    swd_state_init(S);
    TeamwNext nxt{-1, -1, 0.0, 0, 0, 0.0, 0.0, 0.0};
    BH_TP_DECL;
    long rounds = 0;
    bool first = true;
    int cap = 1, ja = 0, ra = 0;
    OneLay mine{0.f, 0.f, 0.f, 0.f}, half{0.f, 0.f, 0.f, 0.f};
    TeamwRound R;
    R.nt = 1; R.nhalf = 0; R.ngrp = 0;
    double mc = 0.0, mom = 0.0;                                         // slot wl of the plan (control wave)
    for (;;) {
        bool fin = false;
        int nt = 1;
#if defined(BH_TEAMW_USTATE)
        S.c1 = uni(S.c1); S.c2 = uni(S.c2); S.c3 = uni(S.c3); S.del1 = uni(S.del1); S.del2 = uni(S.del2); S.del3 = uni(S.del3);
        S.clow = uni(S.clow); S.del1st = uni(S.del1st); S.ceval = uni(S.ceval); S.omega = uni(S.omega);
        S.cprev = uni(S.cprev); S.ck = uni(S.ck); S.cc = uni(S.cc); S.cfail = uni(S.cfail);
#endif
        if (ctl) {
            // (no event pending in all but one round per search: do not even enter the driver's loop --
            // the copies between the two loop headers were 7 % of a round)
            if (S.ev != SWD_EV_NONE) swd_driver(S, lay, src, tg, perl, A.B, true);
            fin = S.st == SWD_ST_DONE;
            BH_TP(0);
            if (!fin) {
                if (first) cap = TPL ? (int)SWD_TEAMW_NT : swd_teamw_cap(S.mmax - S.llw, W, tg.iwave);
                R = swd_teamw_round(S, tg, perl, cap, nxt);
                nt = R.nt;
                BH_TP(13);
                swd_teamw_trial(R, S, wl < nt ? wl : 0, &mc, &mom);
                if (lane < nt) { tcl[lane] = mc; toml[lane] = mom; }
            }
            if (W > 1 && lane == 0) { hdr[0] = fin ? 1 : 0; hdr[1] = nt; hdr[2] = S.mmax; hdr[3] = S.llw; }
        }
        __syncthreads();
        if (W > 1 && !ctl) {
            fin = hdr[0] != 0;
            nt = hdr[1];
            S.mmax = hdr[2];
            S.llw = hdr[3];
        }
        if (fin) break;
        BH_TP(1);
        const int nlm = S.mmax - S.llw;
        if (first) {          // slots per round, and which (trial, layer) this lane assembles
            first = false;
            cap = TPL ? (int)SWD_TEAMW_NT : swd_teamw_cap(nlm, W, tg.iwave);
// (lanes of a wave hold the layers of few trials.  Layer-major -- a wave holding one or two layers for
            // many trials, so that it runs one branch of `var` instead of both -- was measured: 1 % faster on 15
            // layers x 512 lanes, 8 % slower on 5 layers x 64 lanes; same-box A/B, profiles/r03_ab_team.txt)
            ja = (cap > 1 && nlm > 0) ? lane / nlm : 0;
            ra = lane - ja * nlm;
            // this lane's layer and the half-space, read from the shared copy once (after a possible
            // earth-flattening transform by the control wave)
            const int i0 = S.llw - 1 + (ra < nlm ? ra : 0), ih = S.mmax - 1;
            mine = OneLay{lay.d(i0), lay.a(i0), lay.b(i0), lay.rho(i0)};
            half = OneLay{lay.d(ih), lay.a(ih), lay.b(ih), lay.rho(ih)};
        }
        if (TPL) {
            double dl = 0.0;
            if (wl < nt && mc == mc) {                                  // (NaN: a scan slot out of bounds)
                const double wvno = mom / mc;
                dl = (tg.iwave == 1) ? swd_dltar1(lay, S.mmax, S.llw, wvno, mom) : swd_dltar4(lay, S.mmax, S.llw, wvno, mom);
            }
            dels[wl] = dl;
            __syncthreads();
            BH_TP(2);
        } else {
            const int jq = 16 * wave + (wl >> 2);
            if (cap > 1) {
                if (ja < nt && nlm > 0) {
                    const double ac = tcl[ja];
                    if (ac == ac)                                           // (NaN: a scan slot out of bounds)
                        swd_teamw_assemble_one(mine, tg.iwave, S, ra, ac, toml[ja], mats + swd_mat_off(lane));
                }
            } else {
                const double c0 = tcl[0], om0 = toml[0];
                for (int r = lane; r < nlm; r += NL)
                    swd_teamw_assemble_one(lay, tg.iwave, S, r, c0, om0, mats + swd_mat_off(r));
            }
            const bool qvalid = jq < nt;
            const double qc = tcl[qvalid ? jq : 0], qom = toml[qvalid ? jq : 0];
            __syncthreads();
            BH_TP(2);
            if (tg.iwave == 2) {
                // A wave chains 16 trials (one per quad) in the time it chains one: the trials fill waves 0, 1, ...
                // and a wave without any skips the chain -- with eight waves on four SIMDs an idle wave walking
                // through the chain (on trial 0, as an invalid quad does) took issue slots from the wave that
                // shares its SIMD: 8 700 instead of 5 700 cycles per round on 15 layers (team512).
                if (W == 1 || 16 * wave < nt) {
                    const double del = swd_teamw_chain_quad(lay, half, S, qc, qom, mats, (qvalid ? jq : 0) * nlm);
                    if (qvalid && (wl & 3) == 0) dels[jq] = del;
                }
            } else if (lane < nt) {
                dels[lane] = swd_teamw_chain_one(lay, 1, S, mc, mom, mats, lane * nlm);
            }
            __syncthreads();
        }
        BH_TP(3);
        if (ctl) {
            // refinement round: lane j evaluates node j of the bisection tree (what the search does when
            // it arrives there), the consuming loop then follows the decisions
            TeamwNode nd;
            nd.go = SWD_GO_STOP; nd.out = SWD_OUT_CONTROL; nd.c1 = nd.d1 = nd.c2 = nd.d2 = nd.c3n = 0.0;
            unsigned long long goL = 0, goR = 0;
            if (R.nhalf > 0) {
                const bool innode = wl <= R.nhalf;
                nd = swd_teamw_node(S, R, LdsSlots{dels, tcl}, innode ? wl : 0);
                goL = __ballot(innode && nd.go == SWD_GO_LEFT);
                goR = __ballot(innode && nd.go == SWD_GO_RIGHT);
            }
            BH_TP(14);
#if defined(BH_TEAM_PROFILE)
            TeamwVals v{mc, mom, wl < nt ? dels[wl] : 0.0, nt, nd, goL, goR, tp_, &tp_t0_};
#else
            TeamwVals v{mc, mom, wl < nt ? dels[wl] : 0.0, nt, nd, goL, goR};
#endif
            const int used = swd_teamw_consume(S, nv, lay, src, tg, perl, A.B, R, v);
            BH_TP(4);
            BH_TP_COUNT(8, used);
            BH_TP_COUNT(9, nt);
            (void)used;
            rounds++;
        }
    }
    BH_TP_FLUSH(rounds);
    (void)rounds;
}

// ------------------------------------------------------------------------- SWD, narrow teams (8 / 16 / 32 lanes)
// K lanes per search, 64 / K searches per wave, ONE TRIAL PER LANE: the plan, the matching and the tree walk are the
// wide teams' (swd_teamw_round / swd_teamw_consume: scan, cell midpoints, bisection tree, the search that follows),
// with K slots per round, but a lane evaluates its trial's period equation from the half-space to the surface by
// itself -- swd_dltar4 / swd_dltar1 as in swd_kernel, no matrices in LDS, no second phase.  Between the lane kernel
// (one lane, ~33 evaluations per period one behind the other) and a 64-lane team (one evaluation per round spread
// over a trial's layers, ~4 rounds per period, a wave per search): K lanes take a search through in ~4-7 rounds per
// period of one full evaluation each.  For the batches in between -- a few thousand to a few ten thousand searches,
// which fill the chip as 64 / K searches per wave but not as one search per lane.
// (Until round 4 the narrow forms spread the LAYERS of one or two trials over the team's lanes, like the wide teams,
// and speculated on the scan only: 4.5 searches/us on five layers where the lane kernel does 16.8.  swd_team_body
// below, -DBH_NARROW_LAYERS.)
// Every lane of a team runs driver, plan and consuming loop on its own copy of the team's state; the teams of a wave
// diverge there, and meet again for the evaluation.
BH_HD int swd_tpl_team_doubles(int Lmax, int K)
{
    // tc, dls [K]; node brackets [5][K]; Neville table 24; node outcomes K ints; model 4 Lmax floats
    return 7 * K + 24 + (K + 1) / 2 + (4 * Lmax + 1) / 2 + 1;
}

template <int K>
__device__ __forceinline__ void swd_tpl_body(const SwdArgs &A)
{
    extern __shared__ double tlds[];
    static_assert(K == 8 || K == 16 || K == 32, "8, 16 or 32 lanes per search");
    const int sub = threadIdx.x / K, tl = threadIdx.x % K, base = sub * K;
    double *perl = tlds;                             // the target's periods, for all teams of the wave
    double *mem = tlds + BH_NP + (long)sub * swd_tpl_team_doubles(A.Lmax, K);
    double *tc = mem, *dls = tc + K, *ndc = dls + K, *nevt = ndc + 5 * K;
    int *nout = (int *)(nevt + 24);
    TeamLay lay{(float *)(nevt + 24 + (K + 1) / 2), A.Lmax};
    // Every wave of a launch drains the targets' queues one after the other, heaviest target first (A.tord; whatever
    // blockIdx.y says): the targets of a call differ in weight by a factor of three (BASELINE cfg3: Rayleigh group
    // velocities 3.2, Love phase velocities 1.1 table searches per model, capi.hip: target_weight), and with a fixed
    // share of the waves each the call took as long as its heaviest target on a quarter of the chip -- and how long
    // that was depended on which waves the dispatcher had happened to pair on a SIMD (15.1 or 17.2 ms for the same
    // launch, profiles/r04_order_effect.txt).  Longest searches first is the list-scheduling order.
    for (int it = 0; it < A.nsel; it++) {
        const int t = A.tord[it];
        const SwdTargetDev tg = A.tg[t];
        __syncthreads();
        for (int k = threadIdx.x; k < tg.nper; k += SWD_T) perl[k] = A.periods[tg.per_off + k];
        __syncthreads();
        TeamSrc src{A, tg, t, tl, K, 0, (long)blockIdx.x, A.counters + t, nullptr};
        SwdState S;
        swd_state_init(S);
        NevMem nv{nevt, nevt + 12};
        TeamwNext nxt{-1, -1, 0.0, 0, 0, 0.0, 0.0, 0.0};
        TeamwRound R;
        R.nt = 1; R.nhalf = 0; R.ngrp = 0;
        bool live = true;
        for (;;) {
            if (live) {
                if (S.ev != SWD_EV_NONE) swd_driver(S, lay, src, tg, perl, A.B, true);
                live = S.st != SWD_ST_DONE;
            }
            if (!__any(live)) break;
            double mc = __longlong_as_double(0x7ff8000000000000ll), mom = 0.0, dl = 0.0;
            int nt = 0;
            if (live) {
                R = swd_teamw_round(S, tg, perl, K, nxt);
                nt = R.nt;
                if (tl < nt) swd_teamw_trial(R, S, tl, &mc, &mom);
            }
            if (live && tl < nt && mc == mc) {                               // (NaN: a scan slot out of bounds)
                const double wvno = mom / mc;
                dl = (tg.iwave == 1) ? swd_dltar1(lay, S.mmax, S.llw, wvno, mom) : swd_dltar4(lay, S.mmax, S.llw, wvno, mom);
            }
            tc[tl] = mc;
            dls[tl] = dl;
            __syncthreads();
            unsigned long long goL = 0, goR = 0;
            {
                // refinement round: lane j works out node j of the bisection tree (what the search does when it
                // arrives there), the consuming loop then follows the decisions
                const bool innode = live && R.nhalf > 0 && tl <= R.nhalf;
                TeamwNode nd;
                nd.go = SWD_GO_STOP; nd.out = SWD_OUT_CONTROL; nd.c1 = nd.d1 = nd.c2 = nd.d2 = nd.c3n = 0.0;
                if (innode) {
                    nd = swd_teamw_node(S, R, LdsSlots{dls, tc}, tl);
                    nout[tl] = nd.out;
                    ndc[tl] = nd.c1; ndc[K + tl] = nd.d1; ndc[2 * K + tl] = nd.c2; ndc[3 * K + tl] = nd.d2; ndc[4 * K + tl] = nd.c3n;
                }
                goL = (__ballot(innode && nd.go == SWD_GO_LEFT) >> base) & SubVals<K>::MASK;
                goR = (__ballot(innode && nd.go == SWD_GO_RIGHT) >> base) & SubVals<K>::MASK;
            }
            __syncthreads();
            if (live) {
                SubVals<K> v{mc, mom, dl, nt, tl, base, tc, dls, ndc, nout, goL, goR};
                (void)swd_teamw_consume(S, nv, lay, src, tg, perl, A.B, R, v);
            }
            __syncthreads();
        }
    }
}

// Three waves per SIMD: <= 168 VGPRs (the body needs ~140; without the bound the register allocator spreads out
// to 189 and, with the 12.9 KB of LDS a 64-lane team takes, the registers would be what limits a CU to eight teams).
#define BH_TEAMW_ATTR __attribute__((amdgpu_waves_per_eu(3)))
#if defined(BH_TEAM64_LAYERS)
#define BH_TEAM64_ATTR BH_TEAMW_ATTR
#else
#define BH_TEAM64_ATTR __attribute__((amdgpu_waves_per_eu(3)))
#endif
__global__ __launch_bounds__(SWD_T) BH_TEAM64_ATTR void swd_team_kernel(SwdArgs A) { swd_teamw_body<1>(A); }
__global__ __launch_bounds__(2 * SWD_T) BH_TEAMW_ATTR void swd_team128_kernel(SwdArgs A) { swd_teamw_body<2>(A); }
__global__ __launch_bounds__(4 * SWD_T) BH_TEAMW_ATTR void swd_team256_kernel(SwdArgs A) { swd_teamw_body<4>(A); }
__global__ __launch_bounds__(8 * SWD_T) BH_TEAMW_ATTR void swd_team512_kernel(SwdArgs A) { swd_teamw_body<8>(A); }

#if defined(BH_NARROW_LAYERS)
#define BH_NARROW_BODY swd_team_body
#define BH_NARROW_ATTR __attribute__((amdgpu_waves_per_eu(2, 2)))
#else
#define BH_NARROW_BODY swd_tpl_body
#ifndef BH_NARROW_WAVES
#define BH_NARROW_WAVES 2
#endif
#define BH_NARROW_ATTR __attribute__((amdgpu_waves_per_eu(BH_NARROW_WAVES)))
#endif
__global__ __launch_bounds__(SWD_T) BH_NARROW_ATTR void swd_team32_kernel(SwdArgs A) { BH_NARROW_BODY<32>(A); }
__global__ __launch_bounds__(SWD_T) BH_NARROW_ATTR void swd_team16_kernel(SwdArgs A) { BH_NARROW_BODY<16>(A); }
__global__ __launch_bounds__(SWD_T) BH_NARROW_ATTR void swd_team8_kernel(SwdArgs A) { BH_NARROW_BODY<8>(A); }

// -------------------------------------------------------------------------------------------- RF
// bit reversal (+ 1/sqrt(n)) and radix-2 butterflies of Mb buffers in LDS; all threads of the group
__device__ __forceinline__ void rf_block_fft(double *S, int per_model, int Mb, int n, const RfLaunch &P,
                                             const double *tw, int tid)
{
#if !(defined(BH_RF_EXP) && BH_RF_EXP == 1)
    for (int idx = tid; idx < Mb * n; idx += RF_T) {
        int m = idx / n, i = idx - m * n;
        rf_fft_bitrev_scale(S + (long)m * per_model, n, P.log2n, P.sc, i);
    }
    __syncthreads();
#endif
#if defined(BH_RF_EXP) && BH_RF_EXP == 2
    return;
#endif
    for (int l = 1; l < n; l <<= 1) {
        for (int idx = tid; idx < Mb * (n / 2); idx += RF_T) {
            int m = idx / (n / 2), bf = idx - m * (n / 2);
            rf_fft_butterfly(S + (long)m * per_model, tw, l, bf);
        }
        __syncthreads();
    }
}

// 4 waves per SIMD (128 VGPRs, nothing spilled): alone the recursion is latency-bound at 2 and equally fast
// at 3 or 4; at 128 VGPRs a workgroup also fits beside swd_kernel's two waves per SIMD (capi.hip: pick_rf_M).
// ZR: also keep the filtered vertical/radial spectra and return their traces (synrf_cwrap's fz, fr).
#ifndef BH_RF_WAVES
#define BH_RF_WAVES 4
#endif
template <bool ZR>
__global__ __launch_bounds__(RF_T) __attribute__((amdgpu_waves_per_eu(BH_RF_WAVES, BH_RF_WAVES))) void rf_kernel(RfArgs A)
{
    extern __shared__ double S[];
    const RfLaunch &P = A.P;
    const int tid = threadIdx.x;
    const int b0 = blockIdx.x * P.M;
    const int Mb = min(P.M, A.B - b0);
    const int L = P.Lmax;
    const RfLayout lo = rf_layout(L, P.nsamp);
    const int n = P.nsamp;
    const int pm = lo.per_model + (ZR ? 4 * P.nfreq : 0);   // ZR: [spec_r | spec_z] behind the block

    // P1: flatten layers
    for (int idx = tid; idx < Mb * L; idx += RF_T) {
        int m = idx / L, i = idx - m * L;
        long b = b0 + m;
        int nl = A.nlay[b];
        nl = nl < 1 ? 1 : (nl > L ? L : nl);
        if (i < nl)
            rf_phase1_layer(S + (long)m * pm, lo, nl, i, A.h + b * A.mstride,
                            A.vp + b * A.mstride, A.vs + b * A.mstride, A.rho + b * A.mstride,
                            A.qp ? A.qp + b * L : nullptr,
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
            rf_phase2_interface(S + (long)m * pm, lo, P, nl, i, A.vp[b * A.mstride],
                                A.vs[b * A.mstride]);
    }
    __syncthreads();
    // P3: (model, frequency) tasks, model-major
#if defined(BH_RF_EXP) && BH_RF_EXP == 3
    const int ntask = 0;
#else
    const int ntask = Mb * P.nact;
#endif
    for (int task = tid; task < ntask; task += RF_T) {
        int m = task / P.nact, j = task - m * P.nact;
        int nl = A.nlay[b0 + m];
        nl = nl < 1 ? 1 : (nl > L ? L : nl);
        double *Sm = S + (long)m * pm;
        const RfFreq F = rf_freq_load(A.ftab, j);
        if (ZR) {
            cd zr_r, zr_z;
            cd crf = rf_phase3_task(Sm, lo, P, nl, j, F, &zr_r, &zr_z);
            rf_xst(Sm, j, crf);
            st_cd(Sm + lo.per_model + 2 * j, zr_r);
            st_cd(Sm + lo.per_model + 2 * P.nfreq + 2 * j, zr_z);
        } else {
            cd crf = rf_phase3_task(Sm, lo, P, nl, j, F);
            rf_xst(Sm, j, crf);
        }
    }
    // frequencies whose filter weight is below 1e-24 (rf_host.h): zero
    const int nzero = P.nfreq - P.nact;
    for (int idx = tid; idx < Mb * nzero; idx += RF_T) {
        int m = idx / nzero, j = P.nact + (idx - m * nzero);
        double *Sm = S + (long)m * pm;
        rf_xst(Sm, j, mk(0., 0.));
        if (ZR) {
            st_cd(Sm + lo.per_model + 2 * j, mk(0., 0.));
            st_cd(Sm + lo.per_model + 2 * P.nfreq + 2 * j, mk(0., 0.));
        }
    }
    __syncthreads();
    // P4: inverse FFT in LDS (iftr, greens.cpp:136-158)
    const int nh = n / 2 - 1;
    for (int idx = tid; idx < Mb * nh; idx += RF_T) {
        int m = idx / nh, i = n / 2 + 1 + (idx - m * nh);
        rf_fft_hermitian(S + (long)m * pm, n, i);
    }
    __syncthreads();
    rf_block_fft(S, pm, Mb, n, P, A.tw, tid);
    for (int idx = tid; idx < Mb * P.nout; idx += RF_T) {
        int m = idx / P.nout, i = idx - m * P.nout;
        const int nl = A.nlay[b0 + m];
        const bool bad_depth = nl < 1 || nl > L;       // (the phases ran on a clamped copy: discard)
        A.out[(long)(b0 + m) * P.out_stride + P.out_off + i] =
            bad_depth ? __builtin_nan("") : P.qn * S[(long)m * pm + 2 * rf_swz(i)];
    }
    if (ZR) {   // iftr2 (greens.cpp:161-194): one FFT of cx = radial + i*vertical
        __syncthreads();
        for (int idx = tid; idx < Mb * n; idx += RF_T) {
            int m = idx / n, i = idx - m * n;
            double *Sm = S + (long)m * pm;
            rf_xst(Sm, i, rf_fft_pair_entry(Sm + lo.per_model, Sm + lo.per_model + 2 * P.nfreq, n, i));
        }
        __syncthreads();
        rf_block_fft(S, pm, Mb, n, P, A.tw, tid);
        for (int idx = tid; idx < Mb * n; idx += RF_T) {
            int m = idx / n, i = idx - m * n;
            const double *Sm = S + (long)m * pm;
            const int nl = A.nlay[b0 + m];
            const bool bad_depth = nl < 1 || nl > L;   // as for the RF row: never a trace of a truncated model
            A.out_fr[(long)(b0 + m) * n + i] = bad_depth ? __builtin_nan("") : P.qn * Sm[2 * rf_swz(i)];
            A.out_fz[(long)(b0 + m) * n + i] = bad_depth ? __builtin_nan("") : P.qn * Sm[2 * rf_swz(i) + 1];
        }
    }
}

#if defined(BH_TEAM_PROFILE)
extern "C" int bh_debug_team_profile(unsigned long long *out, int reset)
{
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_team_prof), sizeof(g_team_prof)) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[20] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_team_prof), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}
#endif

// ---------------------------------------------------------------------------------------- launch
// LDS of one workgroup of the team kernels with `team` lanes per search
size_t swd_team_lds_bytes(int Lmax, int team)
{
    if (team >= SWD_T) {           // wide teams: max(Lmax, lanes) matrix slots, dels, periods, layer stack
#if defined(BH_TEAM64_LAYERS)
        const int nm = Lmax > team ? Lmax : team;
#else
        const int nm = team == SWD_T ? 0 : (Lmax > team ? Lmax : team);      // (the one-wave team keeps no matrices)
#endif
        return ((size_t)swd_mat_off(nm) + 3 * SWD_TEAMW_NT + BH_NP + 24 * (team / SWD_T) + BH_NP / 2 + (4 * Lmax + 1) / 2) * sizeof(double);
    }
    const int nsub = SWD_T / team;
#if defined(BH_NARROW_LAYERS)
    const int nm = Lmax > team ? Lmax : team;
    return (size_t)nsub * (nm * SWD_NCA + 2 * SWD_TEAM_NT + (4 * Lmax + 1) / 2) * sizeof(double);
#else
    return ((size_t)BH_NP + (size_t)nsub * swd_tpl_team_doubles(Lmax, team)) * sizeof(double);
#endif
}

static int team_index(int team)
{
    return team == 8 ? 0 : team == 16 ? 1 : team == 32 ? 2 : team == 128 ? 4 : team == 256 ? 5 : team == 512 ? 6 : 3;
}

hipError_t launch_swd_team(const SwdArgs &A, int team, int resident_waves, hipStream_t stream)
{
    if (team != 8 && team != 16 && team != 32 && team != 128 && team != 256 && team != 512) team = 64;
    size_t lds = swd_team_lds_bytes(A.Lmax, team);
    static size_t lds_set[7][16] = {{0}, {0}, {0}, {0}, {0}, {0}, {0}};
    void (*kern)(SwdArgs) = team == 8 ? swd_team8_kernel : team == 16 ? swd_team16_kernel
                          : team == 32 ? swd_team32_kernel : team == 128 ? swd_team128_kernel
                          : team == 256 ? swd_team256_kernel : team == 512 ? swd_team512_kernel : swd_team_kernel;
    hipError_t e = ensure_dyn_lds((const void *)kern, lds, lds_set[team_index(team)]);
    if (e != hipSuccess) return e;
    if (team >= SWD_T) {           // one workgroup per search; the hardware back-fills them
        hipLaunchKernelGGL(kern, dim3(A.B, A.ntargets), dim3(team), lds, stream, A);
        return hipGetLastError();
    }
    const int nsub = SWD_T / team;
    int gx = (A.B + nsub - 1) / nsub;
    // persistent waves: no more than stay resident; their teams drain the queue
    int per_target = resident_waves / A.nsel;
    if (per_target < 1) per_target = 1;
    if (gx > per_target) gx = per_target;
    hipLaunchKernelGGL(kern, dim3(gx, A.ntargets), dim3(SWD_T), lds, stream, A);
    return hipGetLastError();
}

hipError_t launch_swd(const SwdArgs &A, int resident_waves, hipStream_t stream)
{
    // model image [4][Lmax][64] fp32, + the result image [max nper][64] fp32 when eight waves per CU
    // (the register-limited residency) still fit 160 KiB with it
    size_t lds = (size_t)4 * A.Lmax * SWD_T * sizeof(float);
    int maxper = 0;
    for (int t = 0; t < A.ntargets; t++)
        if ((A.tmask >> t) & 1u) maxper = A.tg[t].nper > maxper ? A.tg[t].nper : maxper;
    const size_t perbytes = (size_t)BH_NP * sizeof(double);                      // the target's periods
    const size_t staged = lds + (size_t)maxper * SWD_T * sizeof(float);
    SwdArgs B = A;
    static const bool no_stage = std::getenv("BH_SWD_NO_STAGE") != nullptr;      // A/B switch (diagnostic)
    B.stage = (!no_stage && maxper > 0 && 8 * (staged + perbytes) <= 160 * 1024) ? maxper : 0;   // periods staged per lane
    if (B.stage) lds = staged;
    lds += perbytes;
    static size_t lds_set[16] = {0};
    hipError_t e0 = ensure_dyn_lds((const void *)swd_kernel, lds, lds_set);
    if (e0 != hipSuccess) return e0;
    // persistent lanes: no more waves than the chip keeps resident (the queue feeds them), no more
    // than there are models
    // The queue pays once a lane runs >= 3 searches: it removes the 15 % lost to lanes idling until
    // the slowest lane of their wave is done, but ends with a tail of one search length during which
    // waves drain lane by lane ((n+0.5)T against nT/0.85).  Below that, one search per lane and as
    // many waves as models: the hardware back-fills waves as they retire.
    int per_target = resident_waves / A.nsel;
    if (per_target < 1) per_target = 1;
    int gx = (A.B + SWD_T - 1) / SWD_T;
    if (gx > 3 * per_target) gx = per_target;
    hipLaunchKernelGGL(swd_kernel, dim3(gx, A.ntargets), dim3(SWD_T), lds, stream, B);
    return hipGetLastError();
}

size_t rf_lds_bytes(int Lmax, int nsamp, int M, bool zr)
{
    RfLayout lo = rf_layout(Lmax, nsamp);
    return (size_t)M * (lo.per_model + (zr ? 4 * (nsamp / 2 + 1) : 0)) * sizeof(double);
}

hipError_t launch_rf(const RfArgs &A, hipStream_t stream)
{
    const bool zr = A.out_fz != nullptr && A.out_fr != nullptr;
    size_t lds = rf_lds_bytes(A.P.Lmax, A.P.nsamp, A.P.M, zr);
    static size_t lds_set[2][16] = {{0}, {0}};
    hipError_t e = ensure_dyn_lds(zr ? (const void *)rf_kernel<true> : (const void *)rf_kernel<false>, lds,
                                  lds_set[zr]);
    if (e != hipSuccess) return e;
    dim3 grid((A.B + A.P.M - 1) / A.P.M);
    if (zr) hipLaunchKernelGGL(rf_kernel<true>, grid, dim3(RF_T), lds, stream, A);
    else hipLaunchKernelGGL(rf_kernel<false>, grid, dim3(RF_T), lds, stream, A);
    return hipGetLastError();
}

}  // namespace bh
