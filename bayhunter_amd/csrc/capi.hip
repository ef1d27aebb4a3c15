// capi.hip -- the C ABI of libbayhunter_amd.so (include/bayhunter_amd.h).  Host code only.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <tuple>
#include <vector>
#include "../../include/bayhunter_amd.h"
#include "kernels.h"
#include "swd_form_table.h"
#include "rf_host.h"

namespace {

thread_local std::string g_err;

int fail_hip(hipError_t e, const char *what)
{
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    return BH_ERR_HIP;
}
int fail_arg(const char *what)
{
    g_err = what;
    return BH_ERR_ARG;
}
}  // namespace
// shared with chains.cpp (host-only translation unit)
namespace bh {
int fail_arg_(const char *what) { g_err = what; return BH_ERR_ARG; }
int fail_hip_(int e, const char *what) { return fail_hip((hipError_t)e, what); }     // evalplan.hip
}
namespace {
#define BH_HIP(call)                                          \
    do {                                                      \
        hipError_t e_ = (call);                               \
        if (e_ != hipSuccess) return fail_hip(e_, #call);     \
    } while (0)

// HIP is initialised lazily on first use, never at load time: the reference forks one process per
// chain (mcmcOptimizer.py:248-252) and a HIP context must not be created before that fork.
int ensure_device()
{
    static thread_local int ok = 0;
    if (ok) return BH_OK;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        g_err = "no usable HIP device (libbayhunter_amd has no CPU fallback)";
        return BH_ERR_NO_DEVICE;
    }
    ok = 1;
    return BH_OK;
}

// FFT twiddle tables, one per (device, nsamp), built on the host like fork.cpp:50-51.
std::mutex g_tw_mutex;
std::map<std::pair<int, int>, double *> g_tw;

int get_twiddles(int nsamp, const double **out)
{
    int dev = 0;
    BH_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_tw_mutex);
    auto key = std::make_pair(dev, nsamp);
    auto it = g_tw.find(key);
    if (it == g_tw.end()) {
        std::vector<double> tw(2 * (size_t)nsamp);
        bh::rf_fill_twiddles(tw.data(), nsamp);
        double *d = nullptr;
        BH_HIP(hipMalloc((void **)&d, tw.size() * sizeof(double)));
        BH_HIP(hipMemcpy(d, tw.data(), tw.size() * sizeof(double), hipMemcpyHostToDevice));
        it = g_tw.emplace(key, d).first;
    }
    *out = it->second;
    return BH_OK;
}

// Per-frequency constants of the receiver-function kernel (rf_host.h, rf_fill_freq_table): one table per
// (device, nsamp, fsamp, gauss, tshift), built on the host with the reference's expressions.  A run uses a
// handful of parameter sets, but a caller of the single-model drop-ins may vary the Gauss width or the time
// shift per call (a sweep of 24 585 configurations did), so the cache is bounded: beyond kFtabMax tables the
// least recently used one that no call holds goes.  A table is pinned from the look-up to the launch; hipFree
// waits for the device, so a launch that was started with the table has finished when it is released.
constexpr size_t kFtabMax = 64;
struct FreqTable { double *d = nullptr; unsigned long stamp = 0; int pins = 0; };
using FreqKey = std::tuple<int, int, double, double, double>;
std::map<FreqKey, FreqTable> g_ftab;
unsigned long g_ftab_clock = 0;

struct FreqTablePin {               // releases the pin when the call that looked the table up is done with it
    FreqTable *t = nullptr;
    ~FreqTablePin()
    {
        if (!t) return;
        std::lock_guard<std::mutex> lock(g_tw_mutex);
        t->pins--;
    }
};

int get_freq_table(const bh::RfLaunch &P, double fsamp, const double **out, FreqTablePin *pin)
{
    int dev = 0;
    BH_HIP(hipGetDevice(&dev));
    const FreqKey key = std::make_tuple(dev, P.nsamp, fsamp, P.gauss, P.tshift);
    std::vector<double> tab;
    {
        std::lock_guard<std::mutex> lock(g_tw_mutex);
        auto it = g_ftab.find(key);
        if (it != g_ftab.end()) {
            it->second.stamp = ++g_ftab_clock;
            it->second.pins++;
            pin->t = &it->second;                 // (std::map: the address of a value is stable)
            *out = it->second.d;
            return BH_OK;
        }
    }
    // not cached: table and upload outside the lock (two threads may both build one; the second is dropped)
    tab.resize((size_t)bh::RF_FTAB * P.nfreq);
    bh::rf_fill_freq_table(P, tab.data());
    double *d = nullptr;
    BH_HIP(hipMalloc((void **)&d, tab.size() * sizeof(double)));
    hipError_t ce = hipMemcpy(d, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice);
    if (ce != hipSuccess) { (void)hipFree(d); return fail_hip(ce, "hipMemcpy(frequency table)"); }
    std::vector<double *> evicted;
    {
        std::lock_guard<std::mutex> lock(g_tw_mutex);
        auto ins = g_ftab.emplace(key, FreqTable{});
        if (ins.second) ins.first->second.d = d; else evicted.push_back(d);
        ins.first->second.stamp = ++g_ftab_clock;
        ins.first->second.pins++;
        pin->t = &ins.first->second;
        *out = ins.first->second.d;
        while (g_ftab.size() > kFtabMax) {
            auto victim = g_ftab.end();
            for (auto jt = g_ftab.begin(); jt != g_ftab.end(); ++jt)
                if (jt->second.pins == 0 && (victim == g_ftab.end() || jt->second.stamp < victim->second.stamp)) victim = jt;
            if (victim == g_ftab.end()) break;    // every table is held by a call in progress
            evicted.push_back(victim->second.d);
            g_ftab.erase(victim);
        }
    }
    for (double *e : evicted) (void)hipFree(e);   // (synchronises with the device)
    return BH_OK;
}

// Work-queue heads for the dispersion kernels: a ring of slots per device so that launches in flight
// on different streams never share a counter; each launch zeroes its slot on its own stream.  A slot
// is handed out again only when the launch that used it last has finished: every slot carries an
// event recorded behind its launch (release_queue_slot), and a slot whose event is still pending is
// waited for -- with more launches in flight than slots the caller blocks instead of two kernels
// sharing (and corrupting) one counter.  BH_SWD_QUEUE_SLOTS shrinks the ring (test hook).
constexpr int kQueueSlots = 256;
constexpr int kAuxStreams = 2;       // + the caller's stream: at most three kernel forms per call
struct DevState {
    unsigned int *queue = nullptr;
    unsigned long next = 0;
    int cus = 0, nslots = kQueueSlots;
    hipEvent_t done[kQueueSlots];
    bool used[kQueueSlots];
    bool claimed[kQueueSlots];        // taken by a host thread that has not recorded its event yet
    unsigned long waits = 0;          // how often a launch had to wait for its slot (diagnostic)
    // a call whose targets go to different kernel forms launches them on the caller's stream and on these
    // (shared by all calls on the device); the fork/join events belong to the slot, like `done`
    hipStream_t aux[kAuxStreams] = {nullptr, nullptr};
    hipEvent_t fork[kQueueSlots], join[kQueueSlots][kAuxStreams];
    bool forked[kQueueSlots];
    // The stream `done` / `fork` were last recorded on.  A HIP event keeps a pointer to that stream and
    // hipEventQuery / hipEventSynchronize / hipStreamWaitEvent look at its capture state: an event must not
    // outlive the stream it was recorded on.  The caller's stream is the caller's to destroy, so the owner of a
    // stream retires it first (bh_stream_retire: evaluation plans do, evalplan.hip plan_free) and the slot's
    // events are destroyed with it.  (Round 3 did not: a plan's streams died under recorded slot events, and a
    // later launch that claimed such a slot queried freed memory -- "operation not permitted when stream is
    // capturing" in the driver's bench run)
    hipStream_t done_stream[kQueueSlots], fork_stream[kQueueSlots];
};
std::mutex g_dev_mutex;
std::map<int, DevState> g_dev;

int get_queue_slot(unsigned int **slot, int *slot_index, int *resident_waves)
{
    int dev = 0;
    BH_HIP(hipGetDevice(&dev));
    std::unique_lock<std::mutex> lock(g_dev_mutex);
    DevState &d = g_dev[dev];
    if (!d.queue) {
        BH_HIP(hipMalloc((void **)&d.queue, (size_t)kQueueSlots * bh::BH_NT * sizeof(unsigned int)));
        hipDeviceProp_t prop;
        BH_HIP(hipGetDeviceProperties(&prop, dev));
        d.cus = prop.multiProcessorCount;
        for (int i = 0; i < kQueueSlots; i++) {
            d.used[i] = d.claimed[i] = d.forked[i] = false;
            d.done_stream[i] = d.fork_stream[i] = nullptr;
        }
        if (const char *e = std::getenv("BH_SWD_QUEUE_SLOTS")) {
            int v = std::atoi(e);
            if (v >= 1 && v <= kQueueSlots) d.nslots = v;
        }
    }
    int i = (int)(d.next++ % (unsigned long)d.nslots);
    for (int tries = 0; d.claimed[i]; tries++) {       // another thread is between get and release
        if (tries >= d.nslots) {
            g_err = "all work-queue slots are claimed by concurrent launches";
            return BH_ERR_WORKSPACE;
        }
        i = (int)(d.next++ % (unsigned long)d.nslots);
    }
    d.claimed[i] = true;
    if (d.used[i]) {
        hipError_t q = hipEventQuery(d.done[i]);
        if (q == hipErrorNotReady) {
            d.waits++;
            hipEvent_t ev = d.done[i];
            // the ring is full of launches in flight: wait for the oldest one.  Slot i stays ours
            // (next has moved on), so the mutex need not be held while waiting.
            lock.unlock();
            hipError_t we = hipEventSynchronize(ev);
            lock.lock();
            if (we != hipSuccess) { d.claimed[i] = false; return fail_hip(we, "hipEventSynchronize(queue slot)"); }
        } else if (q != hipSuccess) {
            d.claimed[i] = false;
            return fail_hip(q, "hipEventQuery(queue slot)");
        }
    } else {
        hipError_t ce = hipEventCreateWithFlags(&d.done[i], hipEventDisableTiming);
        if (ce != hipSuccess) { d.claimed[i] = false; return fail_hip(ce, "hipEventCreate(queue slot)"); }
        d.used[i] = true;
    }
    *slot = d.queue + (size_t)i * bh::BH_NT;
    *slot_index = i;
    *resident_waves = d.cus * 8;   // 2 waves/SIMD x 4 SIMDs (VGPR-limited, kernels.hip)
    static const char *e = std::getenv("BH_SWD_RESIDENT_WAVES");   // test hook: forces the queue path
    if (e && std::atoi(e) > 0) *resident_waves = std::atoi(e);
    return BH_OK;
}

// marks the end of the launch that used slot `i` on `stream`
int release_queue_slot(int i, hipStream_t stream)
{
    int dev = 0;
    BH_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_dev_mutex);
    DevState &d = g_dev[dev];
    hipError_t e = hipEventRecord(d.done[i], stream);
    d.done_stream[i] = stream;
    d.claimed[i] = false;
    if (e != hipSuccess) return fail_hip(e, "hipEventRecord(queue slot)");
    return BH_OK;
}

// streams and events for a call that launches `naux` kernels beside the one on the caller's stream (slot `i` is
// claimed by the calling thread)
struct SlotStreams {
    hipStream_t aux[kAuxStreams];
    hipEvent_t fork, join[kAuxStreams];
};
int get_slot_streams(int i, int naux, SlotStreams *out)
{
    int dev = 0;
    BH_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_dev_mutex);
    DevState &d = g_dev[dev];
    if (naux > kAuxStreams) return fail_arg("too many concurrent kernel forms");
    for (int k = 0; k < naux; k++)
        if (!d.aux[k]) BH_HIP(hipStreamCreateWithFlags(&d.aux[k], hipStreamNonBlocking));
    if (!d.forked[i]) {
        BH_HIP(hipEventCreateWithFlags(&d.fork[i], hipEventDisableTiming));
        for (int k = 0; k < kAuxStreams; k++) BH_HIP(hipEventCreateWithFlags(&d.join[i][k], hipEventDisableTiming));
        d.forked[i] = true;
    }
    out->fork = d.fork[i];
    for (int k = 0; k < kAuxStreams; k++) { out->aux[k] = d.aux[k]; out->join[k] = d.join[i][k]; }
    return BH_OK;
}

// the fork event of slot `i` (claimed by the calling thread) was recorded on `stream`
void note_fork_stream(int i, hipStream_t stream)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    std::lock_guard<std::mutex> lock(g_dev_mutex);
    g_dev[dev].fork_stream[i] = stream;
}

// Everything the library has recorded on `stream` is finished and forgotten: the caller may destroy it.
int retire_stream(hipStream_t stream)
{
    if (!stream) return BH_OK;                       // the null stream is never destroyed
    BH_HIP(hipStreamSynchronize(stream));
    std::unique_lock<std::mutex> lock(g_dev_mutex);
    hipError_t first = hipSuccess;
    auto note = [&](hipError_t e) { if (e != hipSuccess && first == hipSuccess) first = e; };
    for (auto &kv : g_dev) {
        DevState &d = kv.second;
        if (!d.queue) continue;
        for (int i = 0; i < kQueueSlots; i++) {
            // A slot another thread holds for a launch on ITS stream may still carry this stream's event (that
            // thread may even be waiting for it -- it completed with the synchronisation above): wait until the
            // slot is released, by then its event has been recorded anew on the other stream.
            while (d.claimed[i] && ((d.used[i] && d.done_stream[i] == stream) || (d.forked[i] && d.fork_stream[i] == stream))) {
                lock.unlock();
                std::this_thread::yield();
                lock.lock();
            }
            if (d.claimed[i]) continue;
            if (d.used[i] && d.done_stream[i] == stream) {
                note(hipEventDestroy(d.done[i]));
                d.used[i] = false;
                d.done_stream[i] = nullptr;
            }
            if (d.forked[i] && d.fork_stream[i] == stream) {
                // the join events were recorded on the library's own streams, which live as long as the
                // process; they go with the fork event only to keep the slot's three events one unit
                note(hipEventDestroy(d.fork[i]));
                for (int k = 0; k < kAuxStreams; k++) note(hipEventDestroy(d.join[i][k]));
                d.forked[i] = false;
                d.fork_stream[i] = nullptr;
            }
        }
    }
    if (first != hipSuccess) return fail_hip(first, "hipEventDestroy(retired stream)");
    return BH_OK;
}

// kernel choice of bh_swd_batch: process-wide default (bh_swd_set_kernel), read once per call
std::atomic<int> g_swd_mode{BH_SWD_AUTO};
thread_local int g_last_form = -1;
thread_local int g_last_forms[bh::BH_NT] = {0};   // per target (bh_swd_last_forms)
thread_local int g_forced_forms[bh::BH_NT] = {0};
// what the caller knows about its next batch and the planner cannot see in device memory (bh_swd_hint): the mean
// layer count of the models (0: unknown) and how many such calls are in flight together (>= 1)
thread_local double g_hint_mean = 0.0;
thread_local int g_hint_load = 1;
thread_local int g_nforced = 0;                    // > 0: bh_swd_set_forms is in force for calls with that many targets    // what the last bh_swd_batch of this thread launched (bh_swd_last_form)

long team_threshold()
{
    static const char *e = std::getenv("BH_SWD_TEAM_MAX");
    return e ? std::atol(e) : 0;       // 0: eight searches per resident wave
}

// Work of one search of a target relative to the table's unit (Rayleigh phase velocity at 21 periods): group
// velocities solve each period twice (x1.66 period-equation evaluations, bench.py N_DLTAR), a Love layer step
// costs 0.56 of a Rayleigh one (tools/target_forms.py), mode m runs m root searches per period.
double target_weight(const bh_swd_target &s)
{
    return (s.nper / 21.0) * (s.igr ? 1.66 : 1.0) * (s.iwave == 1 ? 0.56 : 1.0) * (double)s.mode;
}

// Which kernel form runs each target of a call (width[t]: 0 = lane kernel, else lanes per search).
// swd_form_table.h holds the measured milliseconds per call of every form by depth regime and number of
// searches (tools/team_sweep.py on MI355X, 256 CUs, one Rayleigh phase target at 21 periods;
// profiles/r03_team_widths.txt); the planner interpolates in log(searches), scaled to the CU count, and takes
// the fastest.  Wide teams (64 W lanes, speculation across root searches) win up to a few thousand searches,
// the 64-lane teams up to ~20 000, the lane kernel beyond.  bh_swd_set_kernel / bh_swd_set_forms override;
// BH_SWD_TEAM_MAX (searches) caps the use of team kernels.
//
// Several targets: first the one form that is best for all of them (as above, with B x ntargets searches).  One
// refinement, where it was measured to pay: the call is latency-bound on the lane kernel -- a few thousand models,
// every search has a SIMD to itself and the call takes as long as its heaviest target (four targets x 8192
// ten-layer models, 40 periods: 22.7 ms, the Rayleigh group velocities) -- and the batch is small enough for a team
// kernel to take that target beside the others (B <= 32 searches per CU, at most 20 layers).  The heaviest target
// then runs on a team kernel on a second stream if the estimate max(max_t lat w_t, sum_t B w_t / rate) drops by
// 10 % or more (same example: 17.0 ms).  Nothing else is moved: with more models the teams share SIMDs and lose
// more than the lane kernel gains, and moves between team forms measured worse throughout (tools/auto_forms.py,
// tools/forced_forms.py, profiles/r03_mixed_forms.txt).  BH_SWD_NO_MIXED=1 keeps one form per call (A/B).
// Depth.  The table is measured on batches of one depth.  A sampler's batch is ragged -- proposals of a tutorial
// pool have 2 to 14 layers, 4.8 on average -- and taking the regime of its DEEPEST model prices every search as
// deep as that one: the planner put pools of 8 192 and 16 384 chains on 128-lane teams where 64-lane teams are 35 %
// faster (profiles/r04_pool_forms.txt).  With the mean layer count known (bh_swd_hint; the evaluation plan and the
// engine know it from the host's copy of nlay) a form costs max(its latency on the deepest model, its time for all
// searches at the mean depth): the deep minority is started first (processing order) and bounds the call from
// below, the rest is throughput.  `load` calls in flight together (the chain groups of a pool alternate on the
// device) share the chip: the throughput term is read at load x searches.
void plan_forms(int B, int Lmax, int ntargets, const bh_swd_target *targets, long cus, int swd_mode, int *width,
                double mean_layers = 0.0, int load = 1)
{
    if (g_nforced == ntargets) {                       // bh_swd_set_forms (tests, experiments)
        for (int t = 0; t < ntargets; t++) width[t] = g_forced_forms[t];
        return;
    }
    if (swd_mode != BH_SWD_AUTO) {
        const int w = swd_mode == BH_SWD_LANE ? 0 : swd_mode == BH_SWD_TEAM8 ? 8 : swd_mode == BH_SWD_TEAM16 ? 16
                    : swd_mode == BH_SWD_TEAM32 ? 32 : swd_mode == BH_SWD_TEAM128 ? 128
                    : swd_mode == BH_SWD_TEAM256 ? 256 : swd_mode == BH_SWD_TEAM512 ? 512 : 64;
        for (int t = 0; t < ntargets; t++) width[t] = w;
        return;
    }
    auto regime_of = [](double L) { return L <= 3 ? 0 : L <= 6 ? 1 : L <= 12 ? 2 : L <= 20 ? 3 : 4; };
    const int regime = regime_of((double)Lmax);
    const int regime_thr = (mean_layers > 0.0 && mean_layers < (double)Lmax) ? regime_of(mean_layers) : regime;
    if (load < 1) load = 1;
    const long searches = (long)B * ntargets;
    // (calls of fewer than ~3 000 searches do not feel the other group's: with the load priced in, pools of 512 and
    // 1 024 chains -- 2 000-2 300 proposals a call -- were put on 32-lane teams where the one-wave team is 10 % faster,
    // a 1 024-chain inversion 6.1 instead of 5.4 s; from 2 048 chains on -- 3 500 proposals a call -- pricing it is
    // worth 2-20 %: profiles/r04_plan_load.txt)
    if (searches < 3000) load = 1;
    const double scale = 256.0 / (double)cus;                  // the table's chip has 256 CUs
    auto allowed = [&](int k) {
        if (bh::kFormWidth[k] == 0) return true;
        if (team_threshold() > 0 && searches > team_threshold()) return false;
        return bh::swd_team_lds_bytes(Lmax, bh::kFormWidth[k]) <= 160 * 1024;
    };
    // measured range of form k, its time for s searches (1e300: beyond what a wide team was measured for)
    auto measured = [&](int k) {
        int n = bh::kFormSizes;
        while (n > 1 && bh::kFormMs[regime][k][n - 1] < 0) n--;
        return n;
    };
    auto form_ms_in = [&](int rg, int k, double s) {
        const float *t = bh::kFormMs[rg][k], *S = bh::kFormSearches;
        int n = bh::kFormSizes;
        while (n > 1 && t[n - 1] < 0) n--;
        if (s <= S[0]) return (double)t[0];
        if (s > S[n - 1]) return n < bh::kFormSizes ? 1e300 : (double)t[n - 1] * s / S[n - 1];
        if (s == S[n - 1]) return (double)t[n - 1];
        int i = 0;
        while (S[i + 1] <= s) i++;
        const double f = (std::log(s) - std::log(S[i])) / (std::log(S[i + 1]) - std::log(S[i]));
        return t[i] + f * (t[i + 1] - t[i]);
    };
    auto form_ms = [&](int k, double s) {
        const float *t = bh::kFormMs[regime][k], *S = bh::kFormSearches;
        const int n = measured(k);
        if (s <= S[0]) return (double)t[0];
        if (s > S[n - 1]) return n < bh::kFormSizes ? 1e300 : (double)t[n - 1] * s / S[n - 1];
        if (s == S[n - 1]) return (double)t[n - 1];
        int i = 0;
        while (S[i + 1] <= s) i++;
        const double f = (std::log(s) - std::log(S[i])) / (std::log(S[i + 1]) - std::log(S[i]));
        return t[i] + f * (t[i + 1] - t[i]);
    };
    // Targets that are not the table's (one Rayleigh phase velocity at 21 periods): the call's work is that of
    // B x sum of the target weights such searches, and it cannot be shorter than the form's latency for the heaviest
    // target (x 0.7: a search of weight w is w times the evaluations, not w times the rounds).  Four targets x 40
    // periods on ten layers, 2 048 / 4 096 / 8 192 / 16 384 models: this picks 16- / 16- / 8-lane teams / the lane
    // kernel -- measured best of the uniform plans: 16 / 8 (16: +17 %) / 8 / lane (profiles/r04_mixed_tpl.txt).
    double wsum = 0.0, wmax = 0.0;
    for (int t = 0; t < ntargets; t++) {
        const double w = target_weight(targets[t]);
        wsum += w;
        wmax = std::fmax(wmax, w);
    }
    const double eff = (double)B * wsum * scale;
    const double heavy = std::fmax(1.0, 0.7 * wmax);
    int uniform = 0;
    double best = 1e300;
    for (int k = 0; k < 8; k++) {
        if (!allowed(k)) continue;
        double cost = std::fmax(heavy * (double)bh::kFormMs[regime][k][0], form_ms(k, eff));
        if (regime_thr != regime || load > 1)         // ragged batch / shared chip: latency of the deepest | throughput
            cost = std::fmax(heavy * (double)bh::kFormMs[regime][k][0], form_ms_in(regime_thr, k, eff * load));
        if (cost < best) { best = cost; uniform = k; }
    }
    int form[bh::BH_NT];
    for (int t = 0; t < ntargets; t++) form[t] = uniform;
    static const bool no_mixed = std::getenv("BH_SWD_NO_MIXED") != nullptr;
    // the form the heaviest target moves to: 128-lane teams from seven layers up, 8-lane teams below (of 8-, 64-
    // and 128-lane teams beside the lane kernel, seven shapes: profiles/r03_mixed_forms.txt, part 4)
    const int moved = (regime >= 2 && allowed(5)) ? 5 : 1;
    if (ntargets > 1 && !no_mixed && uniform == 0 && regime <= 3 && (long)B <= 32 * cus && allowed(moved)) {
        double w[bh::BH_NT];
        for (int t = 0; t < ntargets; t++) w[t] = target_weight(targets[t]);
        // latency (chip mostly idle) and saturation rate (searches per ms) of a form, from the table's ends
        auto lat = [&](int k) { return (double)bh::kFormMs[regime][k][0]; };
        auto thr = [&](int k) {
            const int n = measured(k);
            return (double)bh::kFormSearches[n - 1] / bh::kFormMs[regime][k][n - 1] / scale;
        };
        auto cost_of = [&](const int *fm) {
            double l = 0.0, sum = 0.0;
            for (int t = 0; t < ntargets; t++) {
                l = std::fmax(l, lat(fm[t]) * w[t]);
                sum += (double)B * w[t] / thr(fm[t]);
            }
            return std::fmax(l, sum);
        };
        double cur = cost_of(form);
        for (int pass = 0; pass + 1 < ntargets; pass++) {
            int worst = -1;                                  // the lane-kernel target the call is waiting for
            for (int t = 0; t < ntargets; t++)
                if (form[t] == 0 && (worst < 0 || w[t] > w[worst])) worst = t;
            if (worst < 0) break;
            form[worst] = moved;
            const double c = cost_of(form);
            if (c > 0.9 * cur) { form[worst] = 0; break; }
            cur = c;
        }
    }
    for (int t = 0; t < ntargets; t++) width[t] = bh::kFormWidth[form[t]];
}

int pick_rf_M(int B, int Lmax, int nsamp)
{
    // Models per workgroup.  rf_kernel is built for four waves per SIMD (128 VGPRs), so that one of its
    // workgroups fits next to the two 192-VGPR waves per SIMD of swd_kernel (2 x 192 + 128 = 512) and the two
    // kernels of a joint evaluation share the CUs from the start instead of running one after the other; the LDS
    // has to allow it too: eight swd_kernel waves hold 8 x 15.7 KiB of a CU's 160 KiB (ten layers, 21 periods),
    // which leaves ~31 KiB -- three 512-sample models.  Alone, four such workgroups per CU (98 KiB) keep all
    // four waves per SIMD resident.  (Six models per workgroup, the round-2 choice, is 1 % faster alone -- no
    // partial last round of tasks -- and can never be co-resident: 65.1 -> 61.2 ms per joint step, same box,
    // profiles/r03_ab_rf_coresident.txt.)
    size_t per = bh::rf_lds_bytes(Lmax, nsamp, 1);
    int M = (int)((26 * 1024) / per);
    static const char *force = std::getenv("BH_RF_M");       // A/B switch (diagnostic)
    if (force && std::atoi(force) > 0) M = std::atoi(force);
    if (M > 8) M = 8;
    if (M < 1) M = 1;
    if (M > B) M = B;
    return M;
}

}  // namespace

extern "C" {

#ifndef BH_SRC_HASH
#define BH_SRC_HASH "unknown"
#endif
const char *bh_version(void) { return "bayhunter_amd 0.2 (gfx950) src " BH_SRC_HASH; }
const char *bh_last_error(void) { return g_err.c_str(); }

int bh_device_count(int *count)
{
    if (!count) return fail_arg("count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    *count = (e == hipSuccess) ? n : 0;
    return BH_OK;
}

int bh_set_device(int device)
{
    int rc = ensure_device();
    if (rc) return rc;
    BH_HIP(hipSetDevice(device));
    return BH_OK;
}

int bh_swd_set_kernel(int mode)
{
    if (mode < BH_SWD_AUTO || mode > BH_SWD_TEAM512) return fail_arg("unknown kernel mode");
    g_swd_mode.store(mode, std::memory_order_relaxed);
    return BH_OK;
}

size_t bh_swd_workspace_bytes(int B, int ntargets, const bh_swd_target *targets)
{
    if (!targets || B <= 0) return 0;
    for (int t = 0; t < ntargets; t++)
        if (targets[t].mode > 1) return (size_t)ntargets * 2 * bh::BH_NP * (size_t)B * sizeof(double);
    return 0;
}

int bh_swd_batch(int B, int Lmax, int model_stride, const int *nlay, const double *h,
                 const double *vp, const double *vs, const double *rho, int ntargets,
                 const bh_swd_target *targets, const double *periods, double *out, int out_stride,
                 int *err, void *workspace, size_t workspace_bytes, void *stream)
{
    return bh_swd_batch_ordered(B, Lmax, model_stride, nlay, h, vp, vs, rho, ntargets, targets, periods, out,
                                out_stride, err, nullptr, workspace, workspace_bytes, stream);
}

int bh_swd_batch_ordered(int B, int Lmax, int model_stride, const int *nlay, const double *h,
                         const double *vp, const double *vs, const double *rho, int ntargets,
                         const bh_swd_target *targets, const double *periods, double *out,
                         int out_stride, int *err, const int *order, void *workspace,
                         size_t workspace_bytes, void *stream)
{
    const double hint_mean = g_hint_mean;              // a hint is about ONE call, whatever becomes of it
    const int hint_load = g_hint_load;
    g_hint_mean = 0.0; g_hint_load = 1;
    if (B < 0 || Lmax < 1 || Lmax > BH_MAX_LAYERS) return fail_arg("B/Lmax out of range");
    if (model_stride < Lmax) return fail_arg("model_stride < Lmax");
    if (ntargets < 1 || ntargets > BH_MAX_TARGETS) return fail_arg("ntargets out of range");
    if (B == 0) return BH_OK;   // nothing to do; an empty device buffer may legitimately be NULL
    if (!nlay || !h || !vp || !vs || !rho || !targets || !periods || !out || !err)
        return fail_arg("NULL pointer");
    int rc = ensure_device();
    if (rc) return rc;
    if (B == 0) return BH_OK;
    bh::SwdArgs A;
    std::memset(&A, 0, sizeof(A));
    for (int t = 0; t < ntargets; t++) {
        const bh_swd_target &s = targets[t];
        if (s.iwave != 1 && s.iwave != 2) return fail_arg("iwave must be 1 (Love) or 2 (Rayleigh)");
        if (s.nper < 0 || s.nper > BH_MAX_PERIODS) return fail_arg("nper out of range (max 60)");
        if (s.mode < 1) return fail_arg("mode must be >= 1");
        if (s.out_off < 0 || s.out_off + s.nper > out_stride) return fail_arg("target does not fit the output row");
        A.tg[t] = bh::SwdTargetDev{s.iwave, s.igr, s.mode, s.iflsph, s.nper, s.per_off, s.out_off, 0};
    }
    size_t need = bh_swd_workspace_bytes(B, ntargets, targets);
    if (need > 0 && (!workspace || workspace_bytes < need)) {
        g_err = "workspace too small for mode > 1 (bh_swd_workspace_bytes)";
        return BH_ERR_WORKSPACE;
    }
    A.B = B; A.Lmax = Lmax; A.ntargets = ntargets; A.out_stride = out_stride; A.mstride = model_stride;
    A.nlay = nlay; A.order = order; A.h = h; A.vp = vp; A.vs = vs; A.rho = rho; A.periods = periods;
    A.out = out; A.err = err; A.ws = (double *)workspace;
    A.vec2 = (Lmax % 2 == 0 && model_stride % 2 == 0 &&
              (((uintptr_t)h | (uintptr_t)vp | (uintptr_t)vs | (uintptr_t)rho) & 15) == 0) ? 1 : 0;
    int resident = 0, slot = 0;
    rc = get_queue_slot(&A.counters, &slot, &resident);
    if (rc) return rc;
    const long cus = resident > 0 ? resident / 8 : 256;
    int width[bh::BH_NT];
    plan_forms(B, Lmax, ntargets, targets, cus, g_swd_mode.load(std::memory_order_relaxed), width, hint_mean, hint_load);
    // one launch per kernel form; the form with the heaviest target goes first, on the caller's stream
    struct Launch { int width, n; double weight; unsigned char sel[bh::BH_NT]; };
    Launch launches[bh::BH_NT];
    int nlaunch = 0;
    for (int t = 0; t < ntargets; t++) {
        int k = 0;
        while (k < nlaunch && launches[k].width != width[t]) k++;
        if (k == nlaunch) { launches[k].width = width[t]; launches[k].n = 0; launches[k].weight = 0.0; nlaunch++; }
        launches[k].sel[launches[k].n++] = (unsigned char)t;
        launches[k].weight = std::fmax(launches[k].weight, target_weight(targets[t]));
        g_last_forms[t] = width[t];
    }
    for (int a = 1; a < nlaunch; a++)
        if (launches[a].weight > launches[0].weight) std::swap(launches[a], launches[0]);
    hipStream_t main_stream = (hipStream_t)stream;
    // the work-queue heads: only the lane kernel and the narrow teams pull searches from a queue (a wide team is one
    // workgroup per search) -- small pools on wide teams are spared a fill kernel per call
    bool queued = false;
    for (int a = 0; a < nlaunch; a++) queued = queued || launches[a].width < 64;
    hipError_t le = queued ? hipMemsetAsync(A.counters, 0, bh::BH_NT * sizeof(unsigned int), main_stream) : hipSuccess;
    SlotStreams ss{};
    bool forked = false;                 // the aux streams and the slot's fork/join events are in `ss`
    if (le == hipSuccess && nlaunch > 1) {
        rc = get_slot_streams(slot, nlaunch - 1, &ss);
        if (rc) { release_queue_slot(slot, main_stream); return rc; }
        forked = true;
        le = hipEventRecord(ss.fork, main_stream);            // the models (and the zeroed counters) are ready
        note_fork_stream(slot, main_stream);
    }
    for (int a = 0; a < nlaunch && le == hipSuccess; a++) {
        const Launch &l = launches[a];
        hipStream_t st = a == 0 ? main_stream : ss.aux[a - 1];
        if (a > 0) le = hipStreamWaitEvent(st, ss.fork, 0);
        if (le != hipSuccess) break;
        A.nsel = l.n;
        A.tmask = 0;
        for (int k = 0; k < l.n; k++) A.tmask |= 1u << l.sel[k];
        for (int k = 0; k < l.n; k++) {                       // heaviest first (insertion sort, at most BH_NT entries)
            int i = k;
            while (i > 0 && target_weight(targets[A.tord[i - 1]]) < target_weight(targets[l.sel[k]])) { A.tord[i] = A.tord[i - 1]; i--; }
            A.tord[i] = l.sel[k];
        }
        if (l.width > 0) {
            int w = l.width;
            while (w > 64 && bh::swd_team_lds_bytes(Lmax, w) > 160 * 1024) w /= 2;
            int team_resident = resident;
#ifndef BH_NARROW_WAVES
#define BH_NARROW_WAVES 2
#endif
            constexpr long BH_NARROW_WAVES_HOST = BH_NARROW_WAVES;     // waves per SIMD the narrow kernels are built for
            if (w < 64) {           // persistent waves of a narrow-team kernel that stay resident
                long per_cu = (long)(160 * 1024 / bh::swd_team_lds_bytes(Lmax, w));
                long waves = cus * (per_cu > 4 * BH_NARROW_WAVES_HOST ? 4 * BH_NARROW_WAVES_HOST : per_cu);
                team_resident = (int)(waves > 0 ? waves : 1);
            }
            le = bh::launch_swd_team(A, w, team_resident, st);
        } else {
            le = bh::launch_swd(A, resident, st);
        }
        if (a > 0 && le == hipSuccess) le = hipEventRecord(ss.join[a - 1], st);
    }
    // the caller's stream continues when every launch is done -- also after a failed launch, for those that
    // were started
    // (an event that was not recorded in this call is complete: waiting for it costs nothing)
    for (int a = 1; a < nlaunch && forked; a++) {
        hipError_t we = hipStreamWaitEvent(main_stream, ss.join[a - 1], 0);
        if (le == hipSuccess) le = we;
    }
    g_last_form = launches[0].width;
    rc = release_queue_slot(slot, main_stream);    // also after a failed launch: the slot is free
    if (le != hipSuccess) return fail_hip(le, "dispersion kernel launch");
    return rc;
}

int bh_swd_last_form(void) { return g_last_form; }

int bh_swd_plan_forms(int B, int Lmax, int ntargets, const bh_swd_target *targets, int cus, int *forms)
{
    const double hint_mean = g_hint_mean;
    const int hint_load = g_hint_load;
    g_hint_mean = 0.0; g_hint_load = 1;
    if (B < 1 || Lmax < 1 || Lmax > BH_MAX_LAYERS || ntargets < 1 || ntargets > BH_MAX_TARGETS || !targets || !forms)
        return fail_arg("bh_swd_plan_forms: bad argument");
    plan_forms(B, Lmax, ntargets, targets, cus > 0 ? cus : 256, g_swd_mode.load(std::memory_order_relaxed), forms,
               hint_mean, hint_load);
    return BH_OK;
}

int bh_swd_hint(double mean_layers, int concurrent_calls)
{
    if (!(mean_layers >= 0.0) || concurrent_calls < 1) return fail_arg("bh_swd_hint: mean_layers >= 0, concurrent_calls >= 1");
    g_hint_mean = mean_layers;
    g_hint_load = concurrent_calls;
    return BH_OK;
}

int bh_swd_set_forms(const int *forms, int ntargets)
{
    if (ntargets == 0 || !forms) { g_nforced = 0; return BH_OK; }
    if (ntargets < 0 || ntargets > BH_MAX_TARGETS) return fail_arg("bh_swd_set_forms: ntargets out of range");
    int distinct = 0;
    for (int t = 0; t < ntargets; t++) {
        const int w = forms[t];
        if (w != 0 && w != 8 && w != 16 && w != 32 && w != 64 && w != 128 && w != 256 && w != 512)
            return fail_arg("bh_swd_set_forms: a form is 0 (lane kernel) or 8, 16, ..., 512 lanes per search");
        bool seen = false;
        for (int u = 0; u < t; u++) seen = seen || forms[u] == w;
        distinct += seen ? 0 : 1;
    }
    if (distinct > kAuxStreams + 1) return fail_arg("bh_swd_set_forms: at most three different forms per call");
    for (int t = 0; t < ntargets; t++) g_forced_forms[t] = forms[t];
    g_nforced = ntargets;
    return BH_OK;
}

int bh_swd_last_forms(int *forms, int ntargets)
{
    if (!forms || ntargets < 0 || ntargets > BH_MAX_TARGETS) return fail_arg("bh_swd_last_forms: bad argument");
    for (int t = 0; t < ntargets; t++) forms[t] = g_last_forms[t];
    return BH_OK;
}

int bh_swd_order_keys(int B, int Lmax, int model_stride, const int *nlay, const double *h, const double *vs,
                      double longest_period, int by_length, int *keys, void *stream)
{
    if (B < 0 || Lmax < 1 || Lmax > BH_MAX_LAYERS) return fail_arg("B/Lmax out of range");
    if (model_stride < Lmax) return fail_arg("model_stride < Lmax");
    if (B == 0) return BH_OK;
    if (!nlay || !h || !vs || !keys) return fail_arg("NULL pointer");
    if (!(longest_period > 0)) return fail_arg("longest_period must be positive");
    int rc = ensure_device();
    if (rc) return rc;
    BH_HIP(bh::launch_order_keys(B, Lmax, model_stride, nlay, h, vs, 0.35 * 3.5 * longest_period, by_length, keys,
                                 (hipStream_t)stream));
    return BH_OK;
}

size_t bh_rf_workspace_bytes(int, int, const bh_rf_params *) { return 0; }

int bh_rf_active_frequencies(const bh_rf_params *par)
{
    if (!par || par->nsamp < 8 || par->nsamp > 4096 || !(par->gauss > 0) || !(par->fsamp > 0)) return 0;
    bh::RfLaunch P;
    std::memset(&P, 0, sizeof(P));
    bh::rf_fill_launch(P, par->p, par->gauss, par->nsamp, par->fsamp, par->tshift, par->nsv, par->waveno, par->nout);
    return P.nact;
}

static int rf_launch_common(int B, int Lmax, int model_stride, const int *nlay, const double *h, const double *vp,
                            const double *vs, const double *rho, const double *qp, const double *qs,
                            const bh_rf_params *par, double sigma, int depth_input, double *out,
                            int out_stride, void *stream, double *out_fz = nullptr,
                            double *out_fr = nullptr)
{
    if (!par) return fail_arg("par is NULL");
    if (B < 0 || Lmax < 1 || Lmax > BH_MAX_LAYERS) return fail_arg("B/Lmax out of range");
    if (model_stride < Lmax) return fail_arg("model_stride < Lmax");
    if (B == 0) return BH_OK;
    if (!nlay || !h || !vp || !vs || !rho || !out) return fail_arg("NULL pointer");
    int n = par->nsamp;
    if (n < 8 || n > 4096 || (n & (n - 1))) return fail_arg("nsamp must be a power of two in 8..4096");
    if (par->waveno != 0 && par->waveno != 1) return fail_arg("waveno must be 0 (P) or 1 (SV)");
    if (par->nout < 1 || par->nout > n) return fail_arg("nout out of range");
    if (par->out_off < 0 || par->out_off + par->nout > out_stride) return fail_arg("RF does not fit the output row");
    if (!(par->gauss > 0) || !(par->fsamp > 0)) return fail_arg("gauss and fsamp must be positive");
    int rc = ensure_device();
    if (rc) return rc;
    if (B == 0) return BH_OK;
    bh::RfArgs A;
    std::memset(&A, 0, sizeof(A));
    bh::rf_fill_launch(A.P, par->p, par->gauss, n, par->fsamp, par->tshift, par->nsv, par->waveno, par->nout);
    A.P.sigma = sigma;
    A.P.out_off = par->out_off;
    A.P.out_stride = out_stride;
    A.P.Lmax = Lmax;
    A.P.depth_input = depth_input;
    A.P.M = (out_fz && out_fr) ? 1 : pick_rf_M(B, Lmax, n);
    if (bh::rf_lds_bytes(Lmax, n, 1, out_fz && out_fr) > 160 * 1024) return fail_arg("model does not fit LDS");
    A.out_fz = out_fz; A.out_fr = out_fr;
    rc = get_twiddles(n, &A.tw);
    if (rc) return rc;
    FreqTablePin pin;                              // held until the launch below has been queued
    rc = get_freq_table(A.P, par->fsamp, &A.ftab, &pin);
    if (rc) return rc;
    A.B = B; A.mstride = model_stride; A.nlay = nlay; A.h = h; A.vp = vp; A.vs = vs; A.rho = rho; A.qp = qp; A.qs = qs;
    A.out = out;
    BH_HIP(bh::launch_rf(A, (hipStream_t)stream));
    return BH_OK;
}

int bh_rf_batch(int B, int Lmax, int model_stride, const int *nlay, const double *h,
                const double *vp, const double *vs, const double *rho, const double *qp,
                const double *qs, const bh_rf_params *par, double *out, int out_stride, void *, size_t,
                void *stream)
{
    return rf_launch_common(B, Lmax, model_stride, nlay, h, vp, vs, rho, qp, qs, par, std::nan(""), 0,
                            out, out_stride, stream);
}

int bh_voronoi_to_layers(int B, int Lmax, const int *nlay, const double *vs_nuclei,
                         const double *z_nuclei, const double *vpvs, const bh_model_priors *pri,
                         double *model, int *valid, void *stream)
{
    if (B < 0 || Lmax < 1 || Lmax > BH_MAX_LAYERS) return fail_arg("B/Lmax out of range");
    if (B == 0) return BH_OK;
    if (!nlay || !vs_nuclei || !z_nuclei || !vpvs || !pri || !model || !valid) return fail_arg("NULL pointer");
    int rc = ensure_device();
    if (rc) return rc;
    if (B == 0) return BH_OK;
    bh::VoronoiArgs A;
    A.B = B; A.Lmax = Lmax; A.nlay = nlay; A.vs = vs_nuclei; A.z = z_nuclei; A.vpvs = vpvs;
    A.model = model; A.valid = valid;
    A.pri = bh::ModelPriorsDev{pri->layers_min, pri->layers_max, pri->vs_min, pri->vs_max, pri->z_min,
                               pri->z_max, pri->thickmin, pri->lowvelperc, pri->highvelperc,
                               pri->mantle_vs, pri->mantle_vpvs};
    BH_HIP(bh::launch_voronoi(A, (hipStream_t)stream));
    return BH_OK;
}

size_t bh_likelihood_workspace_bytes(int B, int ntargets, const bh_like_target *targets)
{
    if (!targets || B <= 0) return 0;
    int groups = 0;                         // [ntargets][B][groups][2]: kernels.h, LikeArgs::gq
    for (int t = 0; t < ntargets; t++)
        if (targets[t].cov == BH_COV_GAUSS && targets[t].n > 0) groups = std::max(groups, bh::gq_groups_of(targets[t].n));
    return (size_t)ntargets * (size_t)B * 2 * (size_t)groups * sizeof(double);
}

int bh_likelihood_batch(int B, int ntargets, const bh_like_target *targets, const double *out,
                        int out_stride, const int *err, int nflags, const double *yobs,
                        const double *noise, const double *aux, double *logL, double *misfits,
                        void *workspace, size_t workspace_bytes, void *stream)
{
    return bh_likelihood_stage(BH_LIKE_STAGE_GAUSS | BH_LIKE_STAGE_REST, B, ntargets, targets, out, out_stride, err, nflags,
                               yobs, noise, aux, logL, misfits, workspace, workspace_bytes, stream);
}

int bh_likelihood_stage(int stages, int B, int ntargets, const bh_like_target *targets, const double *out,
                        int out_stride, const int *err, int nflags, const double *yobs,
                        const double *noise, const double *aux, double *logL, double *misfits,
                        void *workspace, size_t workspace_bytes, void *stream)
{
    if (stages < 1 || stages > 3) return fail_arg("bh_likelihood_stage: stages is BH_LIKE_STAGE_GAUSS | BH_LIKE_STAGE_REST");
    if (B < 0 || ntargets < 1 || ntargets > BH_MAX_TARGETS) return fail_arg("B/ntargets out of range");
    if (B == 0) return BH_OK;
    if (!targets || !out || !yobs || !noise || !logL || !misfits) return fail_arg("NULL pointer");
    if (nflags < 0 || (nflags > 0 && !err)) return fail_arg("err is NULL but nflags > 0");
    int rc = ensure_device();
    if (rc) return rc;
    if (B == 0) return BH_OK;
    bh::LikeArgs A;
    std::memset(&A, 0, sizeof(A));
    int nmax = 1;
    for (int t = 0; t < ntargets; t++) {
        const bh_like_target &s = targets[t];
        if (s.n < 1 || s.n > bh::LIKE_NMAX) return fail_arg("target size out of range (1..1024)");
        if (s.off < 0 || s.off + s.n > out_stride) return fail_arg("target does not fit the output row");
        if (s.cov < 0 || s.cov > 3) return fail_arg("unknown covariance model");
        if ((s.cov == BH_COV_NOCORR_SCALED || s.cov == BH_COV_GAUSS) && !aux) return fail_arg("aux is NULL");
        A.tg[t] = bh::LikeTargetDev{s.n, s.off, s.cov, s.aux_off, s.logdet_extra};
        if (s.n > nmax) nmax = s.n;
    }
    A.B = B; A.ntargets = ntargets; A.out_stride = out_stride; A.nflags = nflags;
    A.out = out; A.err = err; A.yobs = yobs; A.noise = noise; A.aux = aux;
    A.logL = logL; A.misfits = misfits;
    size_t need = bh_likelihood_workspace_bytes(B, ntargets, targets);
    A.gq = (need > 0 && workspace && workspace_bytes >= need) ? (double *)workspace : nullptr;
    A.gq_groups = (int)(need / ((size_t)ntargets * (size_t)B * 2 * sizeof(double)));
    if (stages != 3 && !A.gq) return fail_arg("bh_likelihood_stage: the stages can only be split with a workspace");
    BH_HIP(bh::launch_like(A, nmax, (hipStream_t)stream, stages));
    return BH_OK;
}

// ---- single-model drop-ins ------------------------------------------------------------------
namespace {
struct Scratch {  // per-thread device scratch for the synchronous single-model calls
    void *p = nullptr;
    size_t n = 0;
    int dev = -1;
    int ensure(size_t bytes)
    {
        int cur = 0;
        BH_HIP(hipGetDevice(&cur));
        if (cur != dev) { p = nullptr; n = 0; dev = cur; }   // buffer of another device: leave it be
        if (bytes <= n) return BH_OK;
        if (p) (void)hipFree(p);
        p = nullptr; n = 0;
        BH_HIP(hipMalloc(&p, bytes));
        n = bytes;
        return BH_OK;
    }
};
thread_local Scratch g_scratch;
}  // namespace

int bh_surfdisp96(const float *thkm, const float *vpm, const float *vsm, const float *rhom,
                  int nlayer, int iflsph, int iwave, int mode, int igr, int kmax, const double *t,
                  double *cg, int *err)
{
    if (!thkm || !vpm || !vsm || !rhom || !t || !cg || !err) return fail_arg("NULL pointer");
    if (nlayer < 1 || nlayer > BH_MAX_LAYERS) return fail_arg("nlayer out of range (max 100)");
    if (kmax < 0 || kmax > BH_MAX_PERIODS) return fail_arg("kmax out of range (max 60)");
    int rc = ensure_device();
    if (rc) return rc;
    const int L = nlayer;
    // host staging: [h|vp|vs|rho] fp64 (exactly representable fp32 values), periods, nlay
    std::vector<double> hb(4 * (size_t)L + BH_MAX_PERIODS);
    for (int i = 0; i < L; i++) {
        hb[i] = thkm[i]; hb[L + i] = vpm[i]; hb[2 * L + i] = vsm[i]; hb[3 * L + i] = rhom[i];
    }
    for (int k = 0; k < kmax; k++) hb[4 * L + k] = t[k];
    size_t off_out = hb.size() * sizeof(double);
    size_t off_ws = off_out + BH_MAX_PERIODS * sizeof(double);
    size_t off_int = off_ws + 2 * BH_MAX_PERIODS * sizeof(double);
    size_t total = off_int + 2 * sizeof(int);
    rc = g_scratch.ensure(total);
    if (rc) return rc;
    char *d = (char *)g_scratch.p;
    BH_HIP(hipMemcpy(d, hb.data(), hb.size() * sizeof(double), hipMemcpyHostToDevice));
    int ints[2] = {nlayer, 0};
    BH_HIP(hipMemcpy(d + off_int, ints, sizeof(ints), hipMemcpyHostToDevice));
    bh_swd_target tg = {iwave, igr, mode, iflsph, kmax, 0, 0, 0};
    const double *dm = (const double *)d;
    rc = bh_swd_batch(1, L, L, (const int *)(d + off_int), dm, dm + L, dm + 2 * L, dm + 3 * L, 1, &tg,
                      dm + 4 * L, (double *)(d + off_out), BH_MAX_PERIODS, (int *)(d + off_int) + 1,
                      d + off_ws, 2 * BH_MAX_PERIODS * sizeof(double), nullptr);
    if (rc) return rc;
    BH_HIP(hipDeviceSynchronize());
    if (kmax > 0) BH_HIP(hipMemcpy(cg, d + off_out, kmax * sizeof(double), hipMemcpyDeviceToHost));
    BH_HIP(hipMemcpy(err, (int *)(d + off_int) + 1, sizeof(int), hipMemcpyDeviceToHost));
    return BH_OK;
}

int bh_synrf(int nsamp, double fsamp, double tshift, double p, double a, double nsv, double sigma,
             int waveno, int nlay, const double *z, const double *vp, const double *vs,
             const double *rh, const double *qp, const double *qs, double *fz, double *fr, double *rf)
{
    if (!z || !vp || !vs || !rh || !qp || !qs || !rf) return fail_arg("NULL pointer");
    if ((fz == nullptr) != (fr == nullptr)) return fail_arg("pass both fz and fr, or neither");
    if (nlay < 1 || nlay > BH_MAX_LAYERS) return fail_arg("nlay out of range (max 100)");
    if (nsamp < 8 || nsamp > 4096) return fail_arg("nsamp out of range");
    int rc = ensure_device();
    if (rc) return rc;
    const int L = nlay;
    std::vector<double> hb(6 * (size_t)L);
    for (int i = 0; i < L; i++) {
        hb[i] = z[i]; hb[L + i] = vp[i]; hb[2 * L + i] = vs[i]; hb[3 * L + i] = rh[i];
        hb[4 * L + i] = qp[i]; hb[5 * L + i] = qs[i];
    }
    size_t off_out = hb.size() * sizeof(double);
    size_t off_zr = off_out + (size_t)nsamp * sizeof(double);
    size_t off_int = off_zr + 2 * (size_t)nsamp * sizeof(double);
    rc = g_scratch.ensure(off_int + sizeof(int));
    if (rc) return rc;
    char *d = (char *)g_scratch.p;
    BH_HIP(hipMemcpy(d, hb.data(), hb.size() * sizeof(double), hipMemcpyHostToDevice));
    BH_HIP(hipMemcpy(d + off_int, &nlay, sizeof(int), hipMemcpyHostToDevice));
    bh_rf_params par;
    std::memset(&par, 0, sizeof(par));
    par.p = p; par.gauss = a; par.fsamp = fsamp; par.tshift = tshift; par.nsv = nsv;
    par.nsamp = nsamp; par.waveno = waveno; par.nout = nsamp; par.out_off = 0;
    const double *dm = (const double *)d;
    rc = rf_launch_common(1, L, L, (const int *)(d + off_int), dm, dm + L, dm + 2 * L, dm + 3 * L,
                          dm + 4 * L, dm + 5 * L, &par, sigma, 1, (double *)(d + off_out), nsamp, nullptr,
                          fz ? (double *)(d + off_zr) : nullptr,
                          fz ? (double *)(d + off_zr) + nsamp : nullptr);
    if (rc) return rc;
    BH_HIP(hipDeviceSynchronize());
    BH_HIP(hipMemcpy(rf, d + off_out, (size_t)nsamp * sizeof(double), hipMemcpyDeviceToHost));
    if (fz) {
        BH_HIP(hipMemcpy(fz, d + off_zr, (size_t)nsamp * sizeof(double), hipMemcpyDeviceToHost));
        BH_HIP(hipMemcpy(fr, d + off_zr + (size_t)nsamp * sizeof(double), (size_t)nsamp * sizeof(double),
                         hipMemcpyDeviceToHost));
    }
    return BH_OK;
}

// ---- the reference's own FFI symbols ------------------------------------------------------------
// What f2py binds for `subroutine surfdisp96` (surfdisp96.f:55-56: every argument by reference, no
// return value) and what rfmini.pyx:74-114 binds (wrap.cpp:57-63: returns 1).  Neither has an error
// channel beyond `err`: when the library itself fails (no device, limits) the message goes to stderr
// and the caller sees the reference's failure convention -- err != 0 (SurfDisp.run_model then returns
// (nan, nan), surf96_modsw.py:119-126) or a NaN trace (rejected by Targets.py:204-214).
void surfdisp96_(const float *thkm, const float *vpm, const float *vsm, const float *rhom,
                 const int *nlayer, const int *iflsph, const int *iwave, const int *mode, const int *igr,
                 const int *kmax, const double *t, double *cg, int *err)
{
    int rc = (nlayer && iflsph && iwave && mode && igr && kmax && err)
                 ? bh_surfdisp96(thkm, vpm, vsm, rhom, *nlayer, *iflsph, *iwave, *mode, *igr, *kmax, t, cg, err)
                 : fail_arg("NULL pointer");
    if (rc != BH_OK) {
        std::fprintf(stderr, "libbayhunter_amd: surfdisp96_ failed (%d): %s\n", rc, g_err.c_str());
        if (err) *err = 100 + rc;
    }
}

int synrf_cwrap(int nsamp, double fsamp, double tshift, double p, double a, double nsv, double sigma,
                int waveno, int nlay, double *z, double *vp, double *vs, double *rh, double *qp, double *qs,
                double *fz, double *fr, double *rf)
{
    int rc = bh_synrf(nsamp, fsamp, tshift, p, a, nsv, sigma, waveno, nlay, z, vp, vs, rh, qp, qs, fz, fr, rf);
    if (rc == BH_OK) return 1;
    std::fprintf(stderr, "libbayhunter_amd: synrf_cwrap failed (%d): %s\n", rc, g_err.c_str());
    for (int i = 0; i < nsamp && nsamp <= 4096; i++) {
        if (rf) rf[i] = std::nan("");
        if (fz) fz[i] = std::nan("");
        if (fr) fr[i] = std::nan("");
    }
    return 0;
}

int bh_selftest_division(long n, unsigned seed, int max_exp, long *mismatches)
{
    if (!mismatches || n < 0 || max_exp < 0 || max_exp > 500) return fail_arg("bad argument");
    int rc = ensure_device();
    if (rc) return rc;
    unsigned long long *d = nullptr, h = 0;
    BH_HIP(hipMalloc((void **)&d, sizeof(h)));
    BH_HIP(hipMemset(d, 0, sizeof(h)));
    BH_HIP(bh::launch_division_selftest(n, seed, max_exp, d, nullptr));
    BH_HIP(hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost));
    BH_HIP(hipFree(d));
    *mismatches = (long)h;
    return BH_OK;
}

// ---- plumbing ----------------------------------------------------------------------------------
int bh_malloc(void **dptr, size_t bytes)
{
    if (!dptr) return fail_arg("dptr is NULL");
    int rc = ensure_device();
    if (rc) return rc;
    BH_HIP(hipMalloc(dptr, bytes));
    return BH_OK;
}
int bh_free(void *dptr)
{
    BH_HIP(hipFree(dptr));
    return BH_OK;
}
int bh_rf_cached_tables(void)
{
    std::lock_guard<std::mutex> lock(g_tw_mutex);
    return (int)g_ftab.size();
}
int bh_memcpy_h2d(void *dst, const void *src, size_t bytes, void *stream)
{
    BH_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return BH_OK;
}
int bh_memcpy_d2h(void *dst, const void *src, size_t bytes, void *stream)
{
    BH_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return BH_OK;
}
int bh_stream_synchronize(void *stream)
{
    BH_HIP(hipStreamSynchronize((hipStream_t)stream));
    return BH_OK;
}
int bh_stream_retire(void *stream) { return retire_stream((hipStream_t)stream); }
int bh_stream_create(void **stream)
{
    if (!stream) return fail_arg("stream is NULL");
    int rc = ensure_device();
    if (rc) return rc;
    hipStream_t s = nullptr;
    BH_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (void *)s;
    return BH_OK;
}
int bh_stream_destroy(void *stream)
{
    if (!stream) return fail_arg("the null stream cannot be destroyed");
    int rc = retire_stream((hipStream_t)stream);
    if (rc) return rc;
    BH_HIP(hipStreamDestroy((hipStream_t)stream));
    return BH_OK;
}

}  // extern "C"
