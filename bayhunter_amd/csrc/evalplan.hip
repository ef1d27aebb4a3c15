// evalplan.hip -- one call per batch of proposals: pinned host block -> HBM -> processing order ->
// dispersion / receiver-function / likelihood kernels -> 8*(ntargets+2) bytes per model back.
//
// The sampler's per-iteration hand-over (reference: SingleChain.iterate -> JointTarget.evaluate, one model
// per call, src/SingleChain.py:511-589, src/Targets.py:314-347).  The chain pool used to drive this from
// Python: a dozen torch calls per batch (copies, views, argsort, allocation, events), 0.19 s of host time
// for 150 iterations of 4 096 chains -- as long as the device needed for the arithmetic.  A plan owns
// everything a batch needs (device buffers, pinned staging, two streams, events, sort scratch), so a
// submission is: three async copies (or one), at most seven kernel launches, one async copy back, one
// event.  No allocation, no Python object, no torch in the loop.
#include <hip/hip_runtime.h>
#include <cstring>                     // (before rocprim: its texture iterator calls the host memset)
#include <rocprim/rocprim.hpp>
#include <algorithm>
#include <cstdint>
#include <new>
#include <string>
#include <vector>
#include "../../include/bayhunter_amd.h"

namespace bh { int fail_arg_(const char *what); int fail_hip_(int e, const char *what); }
extern "C" const char *bh_last_error(void);

namespace {

#define EP_HIP(call)                                                         \
    do {                                                                     \
        hipError_t e_ = (call);                                              \
        if (e_ != hipSuccess) return bh::fail_hip_((int)e_, #call);          \
    } while (0)

constexpr int kOrderMin = 1024;   // up to this every team is resident at once: order is irrelevant (layout.py: ORDER_MIN)
struct Interp {                   // numpy.interp(obsx, periods, solved values) for a target with > 60 periods
    int src_off, n_src, dst_off, n_dst;
    long long *j;                 // device tables [n_dst]
    double *xm, *dx;
    unsigned char *last;
};

// y = ((f[j+1] - f[j]) / dx) * xm + f[j]; an observed period that sits on the last solved one takes f[-1]
// (numpy/core/src/multiarray/compiled_base.c: arr_interp; same expression as bayhunter_amd/engine.py)
__global__ void interp_kernel(int B, double *out, int stride, Interp T)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)B * T.n_dst) return;
    const int b = (int)(idx / T.n_dst), i = (int)(idx - (long)b * T.n_dst);
    double *row = out + (long)b * stride;
    const double *f = row + T.src_off;
    const long long j = T.j[i];
    const double f0 = f[j], f1 = f[j + 1];
    const double y = ((f1 - f0) / T.dx[i]) * T.xm[i] + f0;
    row[T.dst_off + i] = T.last[i] ? f[T.n_src - 1] : y;
}

// A plan without dispersion targets has no kernel that raises BH_MODEL_BAD_DEPTH: the flag of a model whose
// layer count is outside 1..L (its receiver-function row is NaN) is set here, as ForwardEngine.run does.
__global__ void depth_flags_kernel(int n, const int *nlay, int L, int nflags, int *err)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int nl = nlay[i];
    err[(long)i * nflags] = (nl < 1 || nl > L) ? BH_MODEL_BAD_DEPTH : 0;
    for (int k = 1; k < nflags; k++) err[(long)i * nflags + k] = 0;
}

__global__ void iota_kernel(int n, int *v)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = i;
}

}  // namespace

struct bh_eval_plan {
    int dev = 0, rows = 0, Lmax = 0, row = 0, nswd = 0, nrf = 0, T = 0, nflags = 0, use_mfma = 1;
    double tmax = 0.0;
    std::vector<bh_swd_target> swd;
    std::vector<bh_rf_params> rf;
    std::vector<bh_like_target> like;
    std::vector<Interp> interp;
    // pinned host staging: [packed rows*4*Lmax | noise rows*2T] doubles, then [nlay rows | chain rows] ints
    char *hblock = nullptr;
    size_t off_noise = 0, off_nlay = 0, off_chain = 0, hbytes = 0;
    double *hres = nullptr;       // [rows] logL, then [rows][T+1] misfits of the last submission
    // device
    char *dblock = nullptr;       // same layout as hblock up to the end of nlay
    double *periods = nullptr, *yobs = nullptr, *aux = nullptr, *out = nullptr, *dres = nullptr;
    int *err = nullptr, *keys = nullptr, *keys_out = nullptr, *iota = nullptr, *order = nullptr;
    void *sort_tmp = nullptr, *like_ws = nullptr, *swd_ws = nullptr;
    size_t sort_bytes = 0, like_bytes = 0, swd_bytes = 0;
    hipStream_t st = nullptr, side = nullptr;
    hipEvent_t fork = nullptr, join = nullptr, done = nullptr;
    int concurrency = 1;          // plans taking turns on the device (bh_eval_set_concurrency)
    bool gauss_on_side = false;   // every dense-Gaussian target is a receiver function's: its product runs behind
                                  // rf_kernel on the side stream, beside the dispersion searches (bh_eval_submit)
    int last_count = 0;           // models of the submission `done` belongs to (set once `done` is recorded)
    bool failed = false;          // the last submission returned an error: nothing to wait for, no results
    std::string failure;
};

static void plan_free(bh_eval_plan *p)
{
    if (!p) return;
    (void)hipSetDevice(p->dev);
    if (p->st) (void)hipStreamSynchronize(p->st);
    if (p->side) (void)hipStreamSynchronize(p->side);
    for (auto &t : p->interp) { (void)hipFree(t.j); (void)hipFree(t.xm); (void)hipFree(t.dx); (void)hipFree(t.last); }
    void *dptr[] = {p->dblock, p->periods, p->yobs, p->aux, p->out, p->dres, p->err, p->keys, p->keys_out,
                    p->iota, p->order, p->sort_tmp, p->like_ws, p->swd_ws};
    for (void *d : dptr)
        if (d) (void)hipFree(d);
    if (p->hblock) (void)hipHostFree(p->hblock);
    if (p->hres) (void)hipHostFree(p->hres);
    if (p->fork) (void)hipEventDestroy(p->fork);
    if (p->join) (void)hipEventDestroy(p->join);
    if (p->done) (void)hipEventDestroy(p->done);
    // the library's own events on these streams (work-queue slot guards, capi.hip) go before the streams do
    if (p->st) (void)bh_stream_retire(p->st);
    if (p->side) (void)bh_stream_retire(p->side);
    if (p->st) (void)hipStreamDestroy(p->st);
    if (p->side) (void)hipStreamDestroy(p->side);
    delete p;
}

template <class T>
static int upload(T **d, const T *h, size_t n)
{
    EP_HIP(hipMalloc((void **)d, std::max<size_t>(n, 1) * sizeof(T)));
    if (n) EP_HIP(hipMemcpy(*d, h, n * sizeof(T), hipMemcpyHostToDevice));
    return BH_OK;
}

extern "C" {

int bh_eval_create(int max_models, int Lmax, int row, int nswd, const bh_swd_target *swd,
                   const double *periods, int nperiods, int nrf, const bh_rf_params *rf, int ntargets,
                   const bh_like_target *like, int nflags, const double *yobs, const double *aux,
                   size_t naux, int ninterp, const bh_eval_interp *interp, int use_mfma,
                   bh_eval_plan **plan)
{
    if (!plan) return bh::fail_arg_("plan is NULL");
    *plan = nullptr;
    if (max_models < 1 || Lmax < 1 || Lmax > BH_MAX_LAYERS || row < 1) return bh::fail_arg_("max_models/Lmax/row out of range");
    if (nswd < 0 || nswd > BH_MAX_TARGETS || nrf < 0 || nswd + nrf < 1) return bh::fail_arg_("no forward targets");
    if (ntargets < 1 || ntargets > BH_MAX_TARGETS || !like || !yobs) return bh::fail_arg_("likelihood targets missing");
    if ((nswd && (!swd || !periods)) || (nrf && !rf) || (ninterp && !interp)) return bh::fail_arg_("NULL pointer");
    if (nflags != (nswd > 0 ? nswd : 1)) return bh::fail_arg_("nflags must be the number of dispersion targets (1 without any)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        bh::fail_arg_("no usable HIP device (libbayhunter_amd has no CPU fallback)");
        return BH_ERR_NO_DEVICE;
    }
    bh_eval_plan *p = new (std::nothrow) bh_eval_plan;
    if (!p) return bh::fail_arg_("out of memory");
    int rc = BH_OK;
    auto bail = [&](int code) { plan_free(p); return code; };
    if (hipGetDevice(&p->dev) != hipSuccess) return bail(bh::fail_arg_("hipGetDevice failed"));
    p->rows = max_models; p->Lmax = Lmax; p->row = row; p->nswd = nswd; p->nrf = nrf; p->T = ntargets;
    p->nflags = nflags; p->use_mfma = use_mfma;
    p->swd.assign(swd, swd + nswd);
    p->rf.assign(rf, rf + nrf);
    p->like.assign(like, like + ntargets);
    for (int t = 0; t < nswd; t++) {
        if (swd[t].per_off < 0 || swd[t].nper < 0 || swd[t].per_off + swd[t].nper > nperiods)
            return bail(bh::fail_arg_("a target's periods lie outside the period array"));
        for (int k = 0; k < swd[t].nper; k++) p->tmax = std::max(p->tmax, periods[swd[t].per_off + k]);
    }
    const size_t R = (size_t)max_models;
    p->off_noise = R * 4 * Lmax * sizeof(double);
    p->off_nlay = p->off_noise + R * 2 * ntargets * sizeof(double);
    p->off_chain = p->off_nlay + R * sizeof(int);
    p->hbytes = p->off_chain + R * sizeof(int);
    if (hipHostMalloc((void **)&p->hblock, p->hbytes, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void **)&p->hres, R * (ntargets + 2) * sizeof(double), hipHostMallocDefault) != hipSuccess)
        return bail(bh::fail_hip_((int)hipErrorOutOfMemory, "hipHostMalloc(staging)"));
    std::memset(p->hblock, 0, p->hbytes);
    if (hipMalloc((void **)&p->dblock, p->off_chain) != hipSuccess ||
        hipMalloc((void **)&p->out, R * row * sizeof(double)) != hipSuccess ||
        hipMalloc((void **)&p->err, R * nflags * sizeof(int)) != hipSuccess ||
        hipMalloc((void **)&p->dres, R * (ntargets + 2) * sizeof(double)) != hipSuccess ||
        hipMalloc((void **)&p->keys, R * sizeof(int)) != hipSuccess ||
        hipMalloc((void **)&p->keys_out, R * sizeof(int)) != hipSuccess ||
        hipMalloc((void **)&p->iota, R * sizeof(int)) != hipSuccess ||
        hipMalloc((void **)&p->order, R * sizeof(int)) != hipSuccess)
        return bail(bh::fail_hip_((int)hipErrorOutOfMemory, "hipMalloc(plan buffers)"));
    if ((rc = upload(&p->periods, periods, (size_t)(nswd ? nperiods : 0)))) return bail(rc);
    if ((rc = upload(&p->yobs, yobs, (size_t)row))) return bail(rc);
    if ((rc = upload(&p->aux, aux, aux ? naux : 0))) return bail(rc);
    for (int i = 0; i < ninterp; i++) {
        const bh_eval_interp &s = interp[i];
        if (s.target < 0 || s.target >= nswd || !s.obsx || s.n_dst < 1 || swd[s.target].nper < 2)
            return bail(bh::fail_arg_("bad interpolation descriptor"));
        const bh_swd_target &tg = swd[s.target];
        const double *xp = periods + tg.per_off;
        const int n = tg.nper;
        std::vector<long long> j(s.n_dst);
        std::vector<double> xm(s.n_dst), dx(s.n_dst);
        std::vector<unsigned char> last(s.n_dst);
        for (int k = 0; k < s.n_dst; k++) {          // np.clip(np.searchsorted(xp, x, side='right') - 1, 0, n - 2)
            const double x = s.obsx[k];
            long long jj = (long long)(std::upper_bound(xp, xp + n, x) - xp) - 1;
            jj = std::min<long long>(std::max<long long>(jj, 0), n - 2);
            j[k] = jj; xm[k] = x - xp[jj]; dx[k] = xp[jj + 1] - xp[jj]; last[k] = x == xp[n - 1];
        }
        Interp T{tg.out_off, n, s.dst_off, s.n_dst, nullptr, nullptr, nullptr, nullptr};
        if ((rc = upload(&T.j, j.data(), j.size())) || (rc = upload(&T.xm, xm.data(), xm.size())) ||
            (rc = upload(&T.dx, dx.data(), dx.size())) || (rc = upload(&T.last, last.data(), last.size()))) {
            p->interp.push_back(T);
            return bail(rc);
        }
        p->interp.push_back(T);
    }
    p->like_bytes = use_mfma ? bh_likelihood_workspace_bytes(max_models, ntargets, like) : 0;
    p->swd_bytes = nswd ? bh_swd_workspace_bytes(max_models, nswd, swd) : 0;
    if ((p->like_bytes && hipMalloc(&p->like_ws, p->like_bytes) != hipSuccess) ||
        (p->swd_bytes && hipMalloc(&p->swd_ws, p->swd_bytes) != hipSuccess))
        return bail(bh::fail_hip_((int)hipErrorOutOfMemory, "hipMalloc(workspaces)"));
    // The receiver-function kernel only back-fills the draining tail of the dispersion kernel when the two
    // streams differ in priority: two plain streams behaved like one (68.1 vs 66.0 ms per 524 288-model step,
    // tools/prio_exp.py), apparently sharing a hardware queue.
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    if (hipStreamCreateWithPriority(&p->st, hipStreamNonBlocking, prio_hi) != hipSuccess ||
        hipStreamCreateWithPriority(&p->side, hipStreamNonBlocking, prio_lo) != hipSuccess ||
        hipEventCreateWithFlags(&p->fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&p->join, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&p->done, hipEventDisableTiming) != hipSuccess)
        return bail(bh::fail_hip_((int)hipErrorUnknown, "stream / event creation"));
    if (p->like_bytes && nswd > 0 && nrf > 0) {
        bool all_rf = true, any = false;
        for (int t = 0; t < ntargets; t++) {
            if (like[t].cov != BH_COV_GAUSS) continue;
            any = true;
            bool in_rf = false;
            for (int i = 0; i < nrf; i++)
                in_rf = in_rf || (like[t].off >= rf[i].out_off && like[t].off + like[t].n <= rf[i].out_off + rf[i].nout);
            all_rf = all_rf && in_rf;
        }
        p->gauss_on_side = any && all_rf;
    }
    hipLaunchKernelGGL(iota_kernel, dim3((max_models + 255) / 256), dim3(256), 0, p->st, max_models, p->iota);
    if (max_models > kOrderMin && nswd) {
        if (rocprim::radix_sort_pairs(nullptr, p->sort_bytes, p->keys, p->keys_out, p->iota, p->order,
                                      (size_t)max_models, 0, 32, p->st) != hipSuccess ||
            hipMalloc(&p->sort_tmp, std::max<size_t>(p->sort_bytes, 16)) != hipSuccess)
            return bail(bh::fail_hip_((int)hipErrorOutOfMemory, "radix sort scratch"));
    }
    if (hipStreamSynchronize(p->st) != hipSuccess) return bail(bh::fail_hip_((int)hipErrorUnknown, "plan set-up"));
    *plan = p;
    return BH_OK;
}

void bh_eval_destroy(bh_eval_plan *plan) { plan_free(plan); }

int bh_eval_buffers(bh_eval_plan *p, double **packed, int **nlay, double **noise, int **chain,
                    double **results)
{
    if (!p) return bh::fail_arg_("plan is NULL");
    if (packed) *packed = (double *)p->hblock;
    if (noise) *noise = (double *)(p->hblock + p->off_noise);
    if (nlay) *nlay = (int *)(p->hblock + p->off_nlay);
    if (chain) *chain = (int *)(p->hblock + p->off_chain);
    if (results) *results = p->hres;
    return BH_OK;
}

// the launches of one submission; `forked` tells the caller whether the side stream was made to wait
static int submit_batch(bh_eval_plan *p, int count, bool *forked)
{
    EP_HIP(hipSetDevice(p->dev));
    const int L = p->Lmax, T = p->T;
    const int *hnlay = (const int *)(p->hblock + p->off_nlay);
    int depth = 1;
    long layers = 0;
    for (int i = 0; i < count; i++) { depth = std::max(depth, hnlay[i]); layers += hnlay[i] > 0 ? hnlay[i] : 0; }
    // the kernels size their LDS images by the deepest model of the batch (even, so that rows can be
    // fetched two layers at a time), not by the allocated row length
    const int Leff = std::min(L, depth + (depth & 1));
    // small pools: the whole block in one copy; large ones: only the used part of each section
    if (p->rows <= 16384) {
        EP_HIP(hipMemcpyAsync(p->dblock, p->hblock, p->off_nlay + (size_t)count * sizeof(int), hipMemcpyHostToDevice, p->st));
    } else {
        EP_HIP(hipMemcpyAsync(p->dblock, p->hblock, (size_t)count * 4 * L * sizeof(double), hipMemcpyHostToDevice, p->st));
        EP_HIP(hipMemcpyAsync(p->dblock + p->off_noise, p->hblock + p->off_noise, (size_t)count * 2 * T * sizeof(double),
                              hipMemcpyHostToDevice, p->st));
        EP_HIP(hipMemcpyAsync(p->dblock + p->off_nlay, p->hblock + p->off_nlay, (size_t)count * sizeof(int),
                              hipMemcpyHostToDevice, p->st));
    }
    const double *dm = (const double *)p->dblock;
    const double *dnoise = (const double *)(p->dblock + p->off_noise);
    const int *dnlay = (const int *)(p->dblock + p->off_nlay);
    const double *h = dm, *vp = dm + L, *vs = dm + 2 * L, *rho = dm + 3 * L;
    int rc;
    const bool overlap = p->nswd > 0 && p->nrf > 0;
    if (overlap) {
        EP_HIP(hipEventRecord(p->fork, p->st));
        EP_HIP(hipStreamWaitEvent(p->side, p->fork, 0));
        *forked = true;
    }
    if (p->nswd) {
        const int *order = nullptr;
        if (count > kOrderMin) {          // deepest first, longest searches first, alike neighbours (engine.reorder)
            if ((rc = bh_swd_order_keys(count, L, 4 * L, dnlay, h, vs, p->tmax, 1, p->keys, p->st))) return rc;
            size_t bytes = p->sort_bytes;
            EP_HIP(rocprim::radix_sort_pairs(p->sort_tmp, bytes, p->keys, p->keys_out, p->iota, p->order,
                                             (size_t)count, 0, 32, p->st));
            order = p->order;
        }
        // the batch is ragged: the planner prices it by its mean depth, not by its deepest model (capi.hip: plan_forms)
        if ((rc = bh_swd_hint((double)layers / (double)count, p->concurrency))) return rc;
        if ((rc = bh_swd_batch_ordered(count, Leff, 4 * L, dnlay, h, vp, vs, rho, p->nswd, p->swd.data(), p->periods,
                                       p->out, p->row, p->err, order, p->swd_ws, p->swd_bytes, p->st)))
            return rc;
        for (const Interp &t : p->interp) {
            const long n = (long)count * t.n_dst;
            hipLaunchKernelGGL(interp_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, p->st, count, p->out, p->row, t);
            EP_HIP(hipGetLastError());
        }
    } else {
        hipLaunchKernelGGL(depth_flags_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, p->st, count, dnlay,
                           Leff, p->nflags, p->err);
        EP_HIP(hipGetLastError());
    }
    hipStream_t rst = overlap ? p->side : p->st;
    for (const bh_rf_params &r : p->rf)
        if ((rc = bh_rf_batch(count, Leff, 4 * L, dnlay, h, vp, vs, rho, nullptr, nullptr, &r, p->out, p->row, nullptr, 0, rst)))
            return rc;
    double *logL = p->dres, *mis = p->dres + count;
    // The dense Gaussian product of a receiver-function target needs that target's columns only: it follows
    // rf_kernel on the side stream, beside this batch's dispersion searches.  Behind the join it was on the critical
    // path of every batch -- and, a short kernel that asks for four SIMDs' worth of registers at once, it waited a
    // third of a millisecond for the OTHER chain group's teams to drain (rocprofv3 of a 4 096-chain pool: 0.31 ms per
    // call where it takes 0.04 ms alone, 22 % of the kernel time).
    const bool staged = overlap && p->gauss_on_side;
    if (staged && (rc = bh_likelihood_stage(BH_LIKE_STAGE_GAUSS, count, T, p->like.data(), p->out, p->row, p->err, p->nflags,
                                            p->yobs, dnoise, p->aux, logL, mis, p->like_ws, p->like_bytes, rst)))
        return rc;
    if (overlap) {
        EP_HIP(hipEventRecord(p->join, p->side));
        EP_HIP(hipStreamWaitEvent(p->st, p->join, 0));
        *forked = false;                  // joined
    }
    if ((rc = bh_likelihood_stage(staged ? BH_LIKE_STAGE_REST : (BH_LIKE_STAGE_GAUSS | BH_LIKE_STAGE_REST), count, T,
                                  p->like.data(), p->out, p->row, p->err, p->nflags, p->yobs, dnoise, p->aux, logL, mis,
                                  p->like_ws, p->like_bytes, p->st)))
        return rc;
    // results: [count] logL then [count][T+1] misfits, contiguous on both sides
    EP_HIP(hipMemcpyAsync(p->hres, p->dres, (size_t)count * (T + 2) * sizeof(double), hipMemcpyDeviceToHost, p->st));
    EP_HIP(hipEventRecord(p->done, p->st));
    return BH_OK;
}

int bh_eval_submit(bh_eval_plan *p, int count)
{
    if (!p) return bh::fail_arg_("plan is NULL");
    if (count < 0 || count > p->rows) return bh::fail_arg_("count out of range");
    p->failed = false;
    p->last_count = 0;
    if (count == 0) return BH_OK;
    bool forked = false;
    const int rc = submit_batch(p, count, &forked);
    if (rc != BH_OK) {
        // Nothing of this submission may be mistaken for a result: bh_eval_wait reports the failure instead of
        // finding a stale (or never recorded, hence "complete") event.  What was queued is drained: the side stream
        // may already be waiting on `fork`, so the main stream joins it before the plan is used or freed.
        p->failed = true;
        p->failure = bh_last_error();
        if (forked && hipEventRecord(p->join, p->side) == hipSuccess) (void)hipStreamWaitEvent(p->st, p->join, 0);
        bh::fail_arg_(p->failure.c_str());           // (the calls above may have replaced the message)
        return rc;
    }
    p->last_count = count;                            // only now: `done` has been recorded for this batch
    return BH_OK;
}

int bh_eval_set_concurrency(bh_eval_plan *p, int plans_in_flight)
{
    if (!p || plans_in_flight < 1) return bh::fail_arg_("bh_eval_set_concurrency: bad argument");
    p->concurrency = plans_in_flight;
    return BH_OK;
}

int bh_eval_wait(bh_eval_plan *p, int *count)
{
    if (!p) return bh::fail_arg_("plan is NULL");
    if (p->failed) {
        if (count) *count = 0;
        return bh::fail_arg_(("the last bh_eval_submit of this plan failed: " + p->failure).c_str());
    }
    if (p->last_count > 0) EP_HIP(hipEventSynchronize(p->done));
    if (count) *count = p->last_count;
    return BH_OK;
}

}  // extern "C"
