// Lock-step chain pool: the host side of the sampler for MANY chains advanced together
// (C ABI: bh_chains_*, include/bayhunter_amd.h).  No device work in here.
//
// What one chain does per iteration follows src/SingleChain.py of the reference:
//   iterate()                       :511-589   choose a move, build the proposal, check the priors,
//                                              [forward + likelihood], u = log(uniform), accept/store
//   _model_vschange/_zvnoi_move     :286-298   one nucleus, Gaussian step
//   _model_layerbirth/_layerdeath   :246-284   add / remove a nucleus, remember (delta vs)^2
//   _sort_modelproposal             :315-328   nuclei ordered by depth
//   _validmodel/_validnoise/_validvpvs :330-434
//   get_acceptance_probability      :463-497   likelihood ratio (+ the birth/death terms of Bodin 2012)
//   adjust_propdist                 :439-461   every 1000 iterations, +-5 % towards the window
//   draw_initvpvs/initmodel/initnoiseparams :94-157
//   append_currentmodel             :508-517   float32 rows + the iteration of acceptance
// and Model.get_vp_vs_h / get_vp (src/Models.py:26-52) for nuclei -> layers.
//
// The random numbers are numpy.random.RandomState's: MT19937 seeded with init_genrand(seed), doubles
// from two outputs (a>>5, b>>6), the polar Box-Muller `legacy_gauss` with its cached second value,
// and `randint` by masked rejection on 32-bit outputs.  The order of draws is the reference's, so a
// chain here and a reference chain with the same seed see the same numbers.
#include <sched.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "../../include/bayhunter_amd.h"

namespace bh { int fail_arg_(const char *what); }

namespace {

// ------------------------------------------------------------------------------------------------
// numpy.random.RandomState
struct Rng {
    uint32_t key[624];
    int pos;
    int has_gauss;
    double gauss;

    void seed(uint32_t s)
    {
        for (int i = 0; i < 624; i++) {
            key[i] = s;
            s = 1812433253u * (s ^ (s >> 30)) + (uint32_t)i + 1u;
        }
        pos = 624;
        has_gauss = 0;
        gauss = 0.0;
    }
    void refill()
    {
        const uint32_t A = 0x9908b0dfu, UP = 0x80000000u, LO = 0x7fffffffu;
        int i;
        uint32_t y;
        for (i = 0; i < 624 - 397; i++) {
            y = (key[i] & UP) | (key[i + 1] & LO);
            key[i] = key[i + 397] ^ (y >> 1) ^ ((y & 1u) ? A : 0u);
        }
        for (; i < 623; i++) {
            y = (key[i] & UP) | (key[i + 1] & LO);
            key[i] = key[i + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? A : 0u);
        }
        y = (key[623] & UP) | (key[0] & LO);
        key[623] = key[396] ^ (y >> 1) ^ ((y & 1u) ? A : 0u);
        pos = 0;
    }
    uint32_t next32()
    {
        if (pos >= 624) refill();
        uint32_t y = key[pos++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return y;
    }
    double next_double()
    {
        int32_t a = (int32_t)(next32() >> 5), b = (int32_t)(next32() >> 6);
        return (a * 67108864.0 + b) / 9007199254740992.0;
    }
    double uniform(double lo, double hi)
    {
        double width = hi - lo;
        return lo + width * next_double();
    }
    double std_gauss()
    {
        if (has_gauss) {
            double g = gauss;
            has_gauss = 0;
            gauss = 0.0;
            return g;
        }
        double x1, x2, r2;
        do {
            x1 = 2.0 * next_double() - 1.0;
            x2 = 2.0 * next_double() - 1.0;
            r2 = x1 * x1 + x2 * x2;
        } while (r2 >= 1.0 || r2 == 0.0);
        double f = std::sqrt(-2.0 * std::log(r2) / r2);
        gauss = f * x1;
        has_gauss = 1;
        return f * x2;
    }
    double normal(double loc, double scale) { return loc + scale * std_gauss(); }
    long randint(long lo, long hi)      // [lo, hi)
    {
        uint64_t span = (uint64_t)(hi - 1 - lo);
        if (span == 0) return lo;
        uint64_t mask = span;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4;
        mask |= mask >> 8; mask |= mask >> 16; mask |= mask >> 32;
        if (span <= 0xffffffffull) {
            if (span == 0xffffffffull) return lo + (long)next32();
            uint32_t v;
            do { v = next32() & (uint32_t)mask; } while (v > span);
            return lo + (long)v;
        }
        uint64_t v;
        do {
            uint64_t hi32 = next32(), lo32 = next32();
            v = ((hi32 << 32) | lo32) & mask;
        } while (v > span);
        return lo + (long)v;
    }
};

enum Move { VSMOD = 0, ZVMOD = 1, BIRTH = 2, DEATH = 3, NOISE = 4, VPVS = 5 };
const int PAR_OF_MOVE[6] = {0, 1, 2, 2, 3, 4};          // PAR_MAP, SingleChain.py:22-23
const int MAXN = 2 * BH_MAX_TARGETS;

// the four nuclei arrays of a chain live side by side in one arena of the pool
struct Span {
    double *p = nullptr;
    double *data() { return p; }
    const double *data() const { return p; }
    double *begin() { return p; }
    double &operator[](long i) { return p[i]; }
    const double &operator[](long i) const { return p[i]; }
};

// One node of a chain's look-ahead tree: a proposal, the iteration it belongs to and where it hangs.
// With a look-ahead of one node per chain (the default) the tree is its root: the proposal of the chain's
// next iteration.
struct Proposal {
    int pn, move, valid, slot, took;     // took: the walk over the results made this proposal the current model
    int parent, src, child[2];           // src: the node whose proposal is the current model here (-1: the chain's);
                                         // child[0]: next iteration if this one is rejected (or invalid), [1]: accepted
    long iter;
    Span pvs, pz;
    double pnoise[MAXN], pvpvs, dvs2, logu, prob;
};

struct Current {                         // the model a proposal starts from
    int n;
    const double *vs, *z, *noise;
    double vpvs;
};

struct Chain {
    Rng rng;
    int n;                                   // nuclei of the current model
    Span vs, z;
    double noise[MAXN], vpvs, like, misfits[BH_MAX_TARGETS + 1];
    double propdist[5], accepted[5], proposed[5];
    long nstored, lastmoditer, iter;         // iter: the next iteration of this chain
    int nnodes;                              // proposals of the running call
    Proposal *nodes;                         // [lookahead]
    Rng *after;                              // [lookahead]: the stream behind each node's draws (NULL: one node, rng itself)
};

// Persistent helper threads: propose/accept run once per iteration over a few thousand chains, too
// short to pay for thread creation -- or even for a futex wake-up (hundreds of microseconds on a
// virtualised host) -- every time.  A worker therefore spins on the job counter for a while after
// its last job (the next call normally follows within the time the GPU needs for the other group)
// and only then goes to sleep on the condition variable.
class Workers {
public:
    Workers()
    {
        if (const char *e = std::getenv("BH_CHAIN_SPIN_US")) spin_us_ = std::max(0, std::atoi(e));
    }
    ~Workers()
    {
        stop_.store(true);
        gen_.fetch_add(1);
        { std::lock_guard<std::mutex> lk(m_); }
        start_.notify_all();
        for (auto &t : th_) t.join();
    }
    // f(i) for i in [0, nitems), split into `parts` contiguous blocks; the caller works on block 0
    void run(int nitems, int parts, const std::function<void(int)> &f)
    {
        if (parts <= 1) {
            for (int i = 0; i < nitems; i++) f(i);
            return;
        }
        // A worker created now must only react to jobs published after its creation: it starts
        // with the current generation as "seen" (gen_ only changes in this function and in the
        // destructor, so the value cannot move between the load and the thread's first poll).
        // Starting at 0 let a late-created worker fall through its wait loop, read the job fields
        // while they were being written below and acknowledge a job it was never counted for.
        const long g0 = gen_.load();
        while ((int)th_.size() < parts - 1) {
            int id = (int)th_.size() + 1;
            th_.emplace_back([this, id, g0]() { loop(id, g0); });
        }
        // every worker acknowledges every job, also those it has no block in: the job fields below
        // are never rewritten while a straggler could still be reading them
        f_ = &f; nitems_ = nitems; parts_ = parts;
        pending_.store((int)th_.size());
        gen_.fetch_add(1);                       // publishes the job (seq_cst)
        { std::lock_guard<std::mutex> lk(m_); }  // a worker about to sleep has either seen it or waits
        start_.notify_all();
        block(0);
        while (pending_.load(std::memory_order_acquire) != 0) cpu_relax();
    }

private:
    static void cpu_relax()
    {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#else
        std::this_thread::yield();
#endif
    }
    void block(int id)
    {
        int per = (nitems_ + parts_ - 1) / parts_;
        int lo = id * per, hi = std::min(nitems_, lo + per);
        for (int i = lo; i < hi; i++) (*f_)(i);
    }
    void loop(int id, long seen)
    {
        for (;;) {
            auto t0 = std::chrono::steady_clock::now();
            int polls = 0;
            while (gen_.load(std::memory_order_acquire) == seen) {
                cpu_relax();
                if (++polls == 256) {
                    polls = 0;
                    auto waited = std::chrono::duration_cast<std::chrono::microseconds>(
                        std::chrono::steady_clock::now() - t0).count();
                    if (waited > spin_us_) {
                        std::unique_lock<std::mutex> lk(m_);
                        start_.wait(lk, [&]() { return gen_.load() != seen; });
                    }
                }
            }
            seen = gen_.load(std::memory_order_acquire);
            if (stop_.load()) return;
            if (id < parts_) block(id);
            pending_.fetch_sub(1, std::memory_order_release);
        }
    }
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable start_;
    const std::function<void(int)> *f_ = nullptr;
    int nitems_ = 0, parts_ = 0;
    std::atomic<int> pending_{0};
    std::atomic<long> gen_{0};
    std::atomic<bool> stop_{false};
    long spin_us_ = 2000;
};

int default_threads()
{
    // three quarters of the cores this process may run on (at most 16 of them: a GPU box gives one GPU a 16-core
    // share of a much larger machine, as a CPU quota): the helpers spin between the jobs of an iteration, and with
    // one spinning thread per core of the quota the HIP runtime's own threads pushed the process over it -- stalls
    // of 10-80 ms while the quota period ran out, 2.0 against 2.2e6 chain iterations/s at 4096 chains (12 or 8
    // threads; larger pools are device-bound and do not care; profiles/r03_chain_threads.txt).
    // BH_CHAIN_THREADS overrides.
    if (const char *e = std::getenv("BH_CHAIN_THREADS")) {
        int v = std::atoi(e);
        if (v > 0) return v;
    }
    cpu_set_t set;
    int n = 0;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
    if (n < 1) n = (int)std::max(1u, std::thread::hardware_concurrency());
    n = std::min(n, 16);
    return std::max(1, n - n / 4);
}

}  // namespace

struct bh_chain_pool {
    bh_chain_config cfg;
    bh_chain_storage st;
    int nchains, maxl, nthreads;
    long iiter, iterations;                  // iiter: the iteration every chain has reached (the slowest chain's next)
    int stage;                               // 0: initial models next, 1: their results awaited,
                                             // 2: proposals next, 3: their results awaited
    int count, failed;
    int lookahead;                           // proposals per chain and call (nodes of its look-ahead tree)
    long calls, advanced, rows;              // propose/accept pairs after the initial one, chain iterations they
                                             // completed, models they handed out
    std::vector<int> noiseinds;
    std::vector<Chain> chains;
    std::vector<double> arena, node_arena;
    std::vector<Proposal> nodes;
    std::vector<Rng> snaps;
    std::vector<int> first_slot;
};

namespace {

void for_chains(bh_chain_pool *p, const std::function<void(int)> &f)
{
    // One set of helper threads for the whole process: the groups of a ChainPool take turns on the
    // host, and two sets of spinning workers would fight for the same cores.
    // The set lives on the heap and is never destroyed; after a fork() (the broker forks chain
    // processes) the child has none of the parent's threads and starts a set of its own.
    static Workers *workers = nullptr;
    static pid_t owner = 0;
    static std::mutex turn;
    std::lock_guard<std::mutex> lk(turn);
    if (!workers || owner != getpid()) {
        workers = new Workers();
        owner = getpid();
    }
    // (a block of fewer than ~128 proposals is not worth a thread)
    workers->run(p->nchains, std::max(1, std::min(std::min(p->nthreads, p->nchains), p->nchains * p->lookahead / 128)), f);
}

// Model.get_vp_vs_h: interfaces midway between neighbouring nuclei, half space h = 0
void layers_of(const bh_chain_config &c, int n, const double *vs, const double *z, double vpvs,
               double *h, double *vp)
{
    double prev = 0.0;
    for (int i = 0; i + 1 < n; i++) {
        double mid = (z[i] + z[i + 1]) / 2.;
        h[i] = mid - prev;
        prev = mid;
    }
    h[n - 1] = 0.0;
    int deep = n;
    if (c.has_mantle)
        for (int i = 0; i < n; i++)
            if (vs[i] >= c.mantle_vs) { deep = i; break; }
    for (int i = 0; i < n; i++) vp[i] = vs[i] * (i < deep ? vpvs : c.mantle_vpvs);
}

bool valid_model(const bh_chain_config &c, int n, const double *vs, const double *z)
{
    int layers = n - 1;
    if (!(layers >= c.layers_min && layers <= c.layers_max)) return false;
    double prev = 0.0, depth = 0.0;
    for (int i = 0; i < n; i++) {
        double h = 0.0;
        if (i + 1 < n) {
            double mid = (z[i] + z[i + 1]) / 2.;
            h = mid - prev;
            prev = mid;
            if (h < c.thickmin) return false;
        }
        depth = (i == 0) ? h : depth + h;            // np.cumsum
        if (depth < c.z_min || depth > c.z_max) return false;
    }
    for (int i = 0; i < n; i++)
        if (vs[i] < c.vs_min || vs[i] > c.vs_max) return false;
    if (c.has_lvz)
        for (int i = 0; i + 1 < n; i++)
            if (!(vs[i + 1] - vs[i] * (1 - c.lvz) > 0)) return false;
    if (c.has_hvz)
        for (int i = 0; i + 1 < n; i++)
            if (!(vs[i] * (1 + c.hvz) - vs[i + 1] > 0)) return false;
    return true;
}

// the checks above return at the first violated rule; the reference tests them in the order layer
// count, thickness, vs, depth, lvz, hvz -- the verdict is the same, and nothing random depends on
// which rule fired.

void sort_by_depth(int n, double *vs, double *z)
{
    bool ordered = true;
    for (int i = 0; i + 1 < n; i++)
        if (!(z[i + 1] - z[i] > 0)) { ordered = false; break; }
    if (ordered) return;
    int idx[BH_MAX_LAYERS + 2];
    double tv[BH_MAX_LAYERS + 2], tz[BH_MAX_LAYERS + 2];
    for (int i = 0; i < n; i++) idx[i] = i;
    std::stable_sort(idx, idx + n, [&](int a, int b) { return z[a] < z[b]; });
    for (int i = 0; i < n; i++) { tv[i] = vs[idx[i]]; tz[i] = z[idx[i]]; }
    std::memcpy(vs, tv, n * sizeof(double));
    std::memcpy(z, tz, n * sizeof(double));
}

int nearest(int n, const double *z, double z0)
{
    int best = 0;
    double d = std::fabs(z[0] - z0);
    for (int i = 1; i < n; i++) {
        double di = std::fabs(z[i] - z0);
        if (di < d) { d = di; best = i; }
    }
    return best;
}

void initial_model(const bh_chain_pool *p, Rng &rng, Proposal &c)
{
    const bh_chain_config &g = p->cfg;
    // draw_initvpvs
    c.pvpvs = g.vpvs_fixed ? g.vpvs_min : rng.uniform(g.vpvs_min, g.vpvs_max);
    // draw_initmodel: the minimum number of layers, redrawn until the priors hold
    int n = g.layers_min + 1;
    for (;;) {
        for (int i = 0; i < n; i++) c.pvs[i] = rng.uniform(g.vs_min, g.vs_max);
        std::sort(c.pvs.data(), c.pvs.data() + n);
        if (g.has_mohoest && n > 1) {
            double moho = rng.normal(g.moho_mean, g.moho_std);
            double half = rng.uniform(1, std::min(5.0, moho));
            c.pz[0] = moho - half;
            c.pz[1] = moho + half;
            for (int i = 2; i < n; i++) c.pz[i] = rng.uniform(g.z_min, g.z_max);
        } else {
            for (int i = 0; i < n; i++) c.pz[i] = rng.uniform(g.z_min, g.z_max);
        }
        std::sort(c.pz.data(), c.pz.data() + n);
        if (valid_model(g, n, c.pvs.data(), c.pz.data())) break;
    }
    c.pn = n;
    // draw_initnoiseparams
    for (int i = 0; i < 2 * g.ntargets; i++)
        c.pnoise[i] = g.noise_fixed[i] ? g.noise_lo[i] : rng.uniform(g.noise_lo[i], g.noise_hi[i]);
    c.valid = 1;
    c.move = -1;
}

// The proposal of iteration `iter` for a chain whose current model is `cur` and whose random stream stands at
// `rng` (SingleChain.iterate up to the forward call, :511-552)
void propose(const bh_chain_pool *p, long iter, Rng &rng, const Current &cur, const double *propdist, Proposal &c)
{
    const bh_chain_config &g = p->cfg;
    const int nnoise = p->noiseinds.empty() ? 0 : 1, nvpvs = g.vpvs_fixed ? 0 : 1;
    // only vs and depth moves (and the hyper-parameters) during the first 1 % of the iterations
    bool early = (double)iter < (double)(-g.iter_burnin) + (double)p->iterations * 0.01;
    int k = (int)rng.randint(0, (early ? 2 : 4) + nnoise + nvpvs);
    int nmodel = early ? 2 : 4;
    int move = k < nmodel ? k : (k - nmodel < nnoise ? NOISE : VPVS);
    c.move = move;
    int n = cur.n;
    std::memcpy(c.pvs.data(), cur.vs, n * sizeof(double));
    std::memcpy(c.pz.data(), cur.z, n * sizeof(double));
    std::memcpy(c.pnoise, cur.noise, 2 * g.ntargets * sizeof(double));
    c.pvpvs = cur.vpvs;
    c.pn = n;
    c.valid = 1;
    switch (move) {
    case VSMOD: {
        long i = rng.randint(0, n);
        double step = rng.normal(0, propdist[0]);
        c.pvs[i] = c.pvs[i] + step;
        break;
    }
    case ZVMOD: {
        long i = rng.randint(n, 2 * n) - n;
        double step = rng.normal(0, propdist[1]);
        c.pz[i] = c.pz[i] + step;
        break;
    }
    case BIRTH: {
        double zb = rng.uniform(g.z_min, g.z_max);
        double before = cur.vs[nearest(n, cur.z, zb)];
        double vb = before + rng.normal(0, propdist[2]);
        c.pz[n] = zb;
        c.pvs[n] = vb;
        c.pn = n + 1;
        double d = vb - before;
        c.dvs2 = d * d;
        break;
    }
    case DEATH: {
        long gone = rng.randint(0, n);
        double zg = cur.z[gone], vg = cur.vs[gone];
        for (int i = (int)gone; i + 1 < n; i++) { c.pvs[i] = c.pvs[i + 1]; c.pz[i] = c.pz[i + 1]; }
        c.pn = n - 1;
        if (c.pn < 1) { c.valid = 0; break; }     // the reference cannot remove the last nucleus
        double d = c.pvs[nearest(c.pn, c.pz.data(), zg)] - vg;
        c.dvs2 = d * d;
        break;
    }
    case NOISE: {
        int i = p->noiseinds[rng.randint(0, (long)p->noiseinds.size())];
        double step = rng.normal(0, propdist[3]);
        c.pnoise[i] = c.pnoise[i] + step;
        for (int j : p->noiseinds)
            if (c.pnoise[j] < g.noise_lo[j] || c.pnoise[j] > g.noise_hi[j]) c.valid = 0;
        return;
    }
    case VPVS: {
        double step = rng.normal(0, propdist[4]);
        c.pvpvs = c.pvpvs + step;
        if (c.pvpvs < g.vpvs_min || c.pvpvs > g.vpvs_max) c.valid = 0;
        return;
    }
    }
    if (!c.valid) return;
    sort_by_depth(c.pn, c.pvs.data(), c.pz.data());
    c.valid = valid_model(g, c.pn, c.pvs.data(), c.pz.data()) ? 1 : 0;
}

bool store(bh_chain_pool *p, int ci, Chain &c, long iter)
{
    const bh_chain_storage &s = p->st;
    if (c.nstored >= s.nmodels) return false;
    const int T = p->cfg.ntargets, W = 2 * p->maxl;
    long row = (long)ci * s.nmodels + c.nstored;
    float *m = s.models + row * W;
    for (int i = 0; i < c.n; i++) { m[i] = (float)c.vs[i]; m[c.n + i] = (float)c.z[i]; }
    for (int i = 0; i <= T; i++) s.misfits[row * (T + 1) + i] = (float)c.misfits[i];
    s.likes[row] = (float)c.like;
    for (int i = 0; i < 2 * T; i++) s.noise[row * 2 * T + i] = (float)c.noise[i];
    s.vpvs[row] = (float)c.vpvs;
    s.iter[row] = (double)iter;
    c.nstored++;
    return true;
}

void take_proposal(bh_chain_pool *p, Chain &c, const Proposal &q, const double *logL, const double *misfits)
{
    const int T = p->cfg.ntargets;
    c.like = logL[q.slot];
    std::memcpy(c.misfits, misfits + (long)q.slot * (T + 1), (T + 1) * sizeof(double));
    c.n = q.pn;
    std::memcpy(c.vs.data(), q.pvs.data(), q.pn * sizeof(double));
    std::memcpy(c.z.data(), q.pz.data(), q.pn * sizeof(double));
    std::memcpy(c.noise, q.pnoise, 2 * T * sizeof(double));
    c.vpvs = q.pvpvs;
    c.lastmoditer = q.iter;
}

void adjust_propdist(const bh_chain_config &g, Chain &c)
{
    for (int i = 0; i < 5; i++)
        if (c.proposed[i] == 0) return;
    for (int i = 0; i < 5; i++) {
        double rate = c.accepted[i] / c.proposed[i] * 100;
        if (rate < g.acceptance[0]) {
            double w = c.propdist[i] * 0.95;
            c.propdist[i] = w < 0.001 ? 0.001 : w;
        } else if (rate > g.acceptance[1]) {
            c.propdist[i] = c.propdist[i] * 1.05;
        }
    }
}

bool adjusts(long iter) { return ((iter % 1000) + 1000) % 1000 == 0; }

Current current_of(const Chain &c, int src)
{
    if (src < 0) return Current{c.n, c.vs.data(), c.z.data(), c.noise, c.vpvs};
    const Proposal &q = c.nodes[src];
    return Current{q.pn, q.pvs.data(), q.pz.data(), q.pnoise, q.pvpvs};
}

// Look-ahead ("prefetching", Brockwell 2006): the proposal of the chain's next iteration and, when the pool
// evaluates more than one model per chain and call, the proposals of the iterations after it for the outcomes
// that are most likely -- a tree whose node (k rejections, m acceptances down from the root) is what
// SingleChain.iterate would propose after exactly those outcomes.  Nothing about an outcome is needed to draw a
// proposal except the current model: the stream position after an iteration is the same whether it was accepted
// or not (one `uniform` after the forward call, :554), and the proposal widths change only every 1000th iteration
// (:578) -- the tree never grows past one of those.  When the likelihoods are back the chain walks the tree from
// its root, deciding each iteration exactly as the reference does, for as far as the tree has the node of the
// outcome: the chain advances by the length of that walk (>= 1), the other nodes were evaluated for nothing.  The
// samples are those of the one-iteration-per-call chain, draw for draw.
struct Edge {
    double prob;
    int node, outcome;
    bool operator<(const Edge &o) const { return prob < o.prob; }
};

void grow(const bh_chain_pool *p, Chain &c)
{
    const bh_chain_config &g = p->cfg;
    c.nnodes = 0;
    if (c.iter >= g.iter_main) return;
    auto fresh = [&](int y, int parent, int src, long iter, double prob) -> Proposal & {
        Proposal &q = c.nodes[y];
        q.parent = parent; q.src = src; q.iter = iter; q.prob = prob;
        q.child[0] = q.child[1] = -1;
        q.slot = -1; q.took = 0; q.logu = 0.0; q.dvs2 = 0.0;
        return q;
    };
    Proposal &root = fresh(0, -1, -1, c.iter, 1.0);
    propose(p, c.iter, c.rng, current_of(c, -1), c.propdist, root);
    if (root.valid) root.logu = std::log(c.rng.uniform(0, 1));
    c.nnodes = 1;
    const int N = p->lookahead;
    if (N == 1) return;
    c.after[0] = c.rng;
    Edge heap[2 * BH_CHAIN_MAX_LOOKAHEAD + 2];
    int nheap = 0;
    auto offer = [&](int x) {
        const Proposal &q = c.nodes[x];
        if (q.iter + 1 >= g.iter_main || (q.valid && adjusts(q.iter))) return;
        double a = 0.0;
        if (q.valid) {
            // the expected outcome: the chain's own acceptance rate for this kind of move so far.  (Rates of the last
            // ~32 outcomes per move, or per move and size class of the step, predict no better: 6.6 iterations per
            // call of 32 proposals either way on the tutorial inversion -- whether a step goes uphill is the coin.)
            int par = PAR_OF_MOVE[q.move];
            a = (c.accepted[par] + 1.) / (c.proposed[par] + 3.);
            heap[nheap++] = Edge{q.prob * a, x, 1};
            std::push_heap(heap, heap + nheap);
        }
        heap[nheap++] = Edge{q.prob * (1. - a), x, 0};
        std::push_heap(heap, heap + nheap);
    };
    offer(0);
    while (c.nnodes < N && nheap > 0) {
        std::pop_heap(heap, heap + nheap);
        Edge e = heap[--nheap];
        int y = c.nnodes++;
        Proposal &q = fresh(y, e.node, e.outcome ? e.node : c.nodes[e.node].src, c.nodes[e.node].iter + 1, e.prob);
        c.nodes[e.node].child[e.outcome] = y;
        Rng &rng = c.after[y];
        rng = c.after[e.node];
        propose(p, q.iter, rng, current_of(c, q.src), c.propdist, q);
        if (q.valid) q.logu = std::log(rng.uniform(0, 1));
        offer(y);
    }
}

// The decisions of the iterations the tree covers (SingleChain.iterate from :554 on), root first
bool walk(bh_chain_pool *p, int ci, Chain &c, const double *logL, const double *misfits)
{
    const bh_chain_config &g = p->cfg;
    bool ok = true;
    int x = 0;
    while (c.nnodes > 0) {
        Proposal &q = c.nodes[x];
        int outcome = 0;
        if (q.valid) {                                  // (otherwise iterate() returned before anything else)
            int par = PAR_OF_MOVE[q.move];
            c.proposed[par] += 1;
            double ratio = logL[q.slot] - c.like, alpha = ratio;
            if (q.move == BIRTH || q.move == DEATH) {
                double theta = c.propdist[2], dv = g.vs_max - g.vs_min;
                double B = q.dvs2 / (2. * (theta * theta));
                if (q.move == BIRTH) {
                    double A = (theta * std::sqrt(2 * M_PI)) / dv;
                    alpha = std::log(A) + B + ratio;
                } else {
                    double A = dv / (theta * std::sqrt(2 * M_PI));
                    alpha = std::log(A) - B + ratio;
                }
            }
            if (q.logu < alpha) {
                outcome = 1;
                q.took = 1;
                take_proposal(p, c, q, logL, misfits);
                if (!store(p, ci, c, q.iter)) ok = false;
                c.accepted[par] += 1;
            }
            if (adjusts(q.iter)) adjust_propdist(g, c);
        }
        c.iter = q.iter + 1;
        int next = ok ? q.child[outcome] : -1;
        if (next < 0) {
            if (c.after) c.rng = c.after[x];
            break;
        }
        x = next;
    }
    return ok;
}

void point_nodes(bh_chain_pool *p)
{
    const long W = p->maxl + 2;
    const int N = p->lookahead;
    p->nodes.assign((size_t)p->nchains * N, Proposal());
    p->node_arena.assign((size_t)p->nchains * N * 2 * W, 0.0);
    p->snaps.clear();
    p->snaps.shrink_to_fit();
    if (N > 1) p->snaps.resize((size_t)p->nchains * N);
    for (int i = 0; i < p->nchains; i++) {
        Chain &c = p->chains[i];
        c.nodes = p->nodes.data() + (size_t)i * N;
        c.after = N > 1 ? p->snaps.data() + (size_t)i * N : nullptr;
        c.nnodes = 0;
        for (int k = 0; k < N; k++) {
            Proposal &q = c.nodes[k];
            double *a = p->node_arena.data() + ((size_t)i * N + k) * 2 * W;
            q.pvs.p = a; q.pz.p = a + W;
            std::memset(q.pnoise, 0, sizeof(q.pnoise));
            q.pn = 0; q.move = -1; q.valid = 0; q.slot = -1; q.took = 0;
            q.parent = q.src = q.child[0] = q.child[1] = -1;
            q.iter = 0; q.pvpvs = q.dvs2 = q.logu = q.prob = 0.0;
        }
    }
}

}  // namespace

extern "C" {

int bh_chains_create(const bh_chain_config *cfg, int nchains, const unsigned *seeds,
                     const bh_chain_storage *st, bh_chain_pool **out)
{
    if (!cfg || !seeds || !st || !out || nchains < 1) return bh::fail_arg_("bh_chains_create: NULL argument or no chains");
    if (cfg->ntargets < 1 || cfg->ntargets > BH_MAX_TARGETS) return bh::fail_arg_("bh_chains_create: ntargets out of range");
    if (cfg->layers_min < 0 || cfg->layers_max < cfg->layers_min || cfg->layers_max + 1 > BH_MAX_LAYERS)
        return bh::fail_arg_("bh_chains_create: layer prior out of range");
    if (!st->models || !st->misfits || !st->likes || !st->noise || !st->vpvs || !st->iter || st->nmodels < 1)
        return bh::fail_arg_("bh_chains_create: storage arrays missing");
    if (cfg->iter_burnin < 0 || cfg->iter_main < 0) return bh::fail_arg_("bh_chains_create: negative iteration count");
    bh_chain_pool *p = new (std::nothrow) bh_chain_pool();
    if (!p) return bh::fail_arg_("bh_chains_create: out of memory");
    p->cfg = *cfg;
    p->st = *st;
    p->nchains = nchains;
    p->maxl = cfg->layers_max + 1;
    p->nthreads = default_threads();
    p->iterations = cfg->iter_burnin + cfg->iter_main;
    p->iiter = -cfg->iter_burnin;
    p->stage = 0;
    p->count = 0;
    p->failed = 0;
    p->lookahead = 1;
    p->calls = p->advanced = p->rows = 0;
    for (int i = 0; i < 2 * cfg->ntargets; i++)
        if (!cfg->noise_fixed[i]) p->noiseinds.push_back(i);
    p->chains.resize(nchains);
    const long W = p->maxl + 2;
    p->arena.assign((size_t)nchains * 2 * W, 0.0);
    p->first_slot.assign(nchains + 1, 0);
    for (int i = 0; i < nchains; i++) {
        Chain &c = p->chains[i];
        c.rng.seed(seeds[i]);
        c.n = 0;
        double *a = p->arena.data() + (size_t)i * 2 * W;
        c.vs.p = a; c.z.p = a + W;
        std::memset(c.noise, 0, sizeof(c.noise));
        std::memset(c.misfits, 0, sizeof(c.misfits));
        c.vpvs = 0.0; c.like = 0.0;
        for (int k = 0; k < 5; k++) { c.propdist[k] = cfg->propdist[k]; c.accepted[k] = c.proposed[k] = 0.0; }
        c.nstored = 0; c.lastmoditer = p->iiter; c.iter = p->iiter;

    }
    point_nodes(p);
    // start the helper threads now rather than inside the first timed iteration (creating 15 threads
    // in a process that has the GPU's address ranges mapped costs ~0.1 s on a GPU box)
    for_chains(p, [](int) {});
    *out = p;
    return BH_OK;
}

void bh_chains_destroy(bh_chain_pool *p) { delete p; }

int bh_chains_set_threads(bh_chain_pool *p, int n)
{
    if (!p || n < 1) return bh::fail_arg_("bh_chains_set_threads: bad argument");
    p->nthreads = n;
    for_chains(p, [](int) {});
    return BH_OK;
}

int bh_chains_set_lookahead(bh_chain_pool *p, int nodes)
{
    if (!p || nodes < 1 || nodes > BH_CHAIN_MAX_LOOKAHEAD)
        return bh::fail_arg_("bh_chains_set_lookahead: 1 .. BH_CHAIN_MAX_LOOKAHEAD proposals per chain");
    if (p->stage != 0 && p->stage != 2) return bh::fail_arg_("bh_chains_set_lookahead: results of the last proposals are still due");
    if (nodes == p->lookahead) return BH_OK;
    p->lookahead = nodes;
    point_nodes(p);
    for_chains(p, [](int) {});
    return BH_OK;
}

int bh_chains_lookahead(const bh_chain_pool *p) { return p ? p->lookahead : 0; }
long bh_chains_rows(const bh_chain_pool *p) { return p ? (long)p->nchains * p->lookahead : 0; }

int bh_chains_advance(const bh_chain_pool *p, long *calls, long *iterations, long *rows)
{
    if (!p) return bh::fail_arg_("bh_chains_advance: NULL pool");
    if (calls) *calls = p->calls;
    if (iterations) *iterations = p->advanced;
    if (rows) *rows = p->rows;
    return BH_OK;
}

int bh_chains_done(const bh_chain_pool *p) { return (p && p->stage >= 2 && p->iiter >= p->cfg.iter_main) ? 1 : 0; }
long bh_chains_iteration(const bh_chain_pool *p) { return p ? p->iiter : 0; }

int bh_chains_propose(bh_chain_pool *p, int Lmax, double *packed, int *nlay, double *noise,
                      int *chain, int *count)
{
    if (!p || !packed || !nlay || !noise || !chain || !count) return bh::fail_arg_("bh_chains_propose: NULL argument");
    if (p->failed) return bh::fail_arg_("bh_chains_propose: the pool stopped after an error");
    if (p->stage != 0 && p->stage != 2) return bh::fail_arg_("bh_chains_propose: results of the last proposals are still due");
    if (Lmax < p->maxl) return bh::fail_arg_("bh_chains_propose: Lmax smaller than layers_max + 1");
    if (p->stage == 2 && p->iiter >= p->cfg.iter_main) { *count = 0; return BH_OK; }
    const bool init = p->stage == 0;
    int *first = p->first_slot.data();
    for_chains(p, [=](int i) {
        Chain &c = p->chains[i];
        if (init) {
            Proposal &q = c.nodes[0];
            q.parent = q.src = q.child[0] = q.child[1] = -1;
            q.iter = c.iter; q.prob = 1.0; q.took = 0; q.slot = -1; q.logu = q.dvs2 = 0.0;
            initial_model(p, c.rng, q);
            if (c.after) c.after[0] = c.rng;
            c.nnodes = 1;
        } else {
            grow(p, c);
        }
        int valid = 0;
        for (int k = 0; k < c.nnodes; k++) valid += c.nodes[k].valid ? 1 : 0;
        first[i + 1] = valid;
    });
    first[0] = 0;
    for (int i = 0; i < p->nchains; i++) first[i + 1] += first[i];
    const int T2 = 2 * p->cfg.ntargets;
    for_chains(p, [=](int i) {
        Chain &c = p->chains[i];
        int k = first[i];
        for (int x = 0; x < c.nnodes; x++) {
            Proposal &q = c.nodes[x];
            q.slot = -1;
            if (!q.valid) continue;
            q.slot = k++;
            chain[q.slot] = i;
            double *row = packed + (long)q.slot * 4 * Lmax;
            std::memset(row, 0, 4 * Lmax * sizeof(double));
            double *h = row, *vp = row + Lmax, *vs = row + 2 * Lmax, *rho = row + 3 * Lmax;
            layers_of(p->cfg, q.pn, q.pvs.data(), q.pz.data(), q.pvpvs, h, vp);
            for (int l = 0; l < q.pn; l++) { vs[l] = q.pvs[l]; rho[l] = vp[l] * 0.32 + 0.77; }
            nlay[q.slot] = q.pn;
            std::memcpy(noise + (long)q.slot * T2, q.pnoise, T2 * sizeof(double));
        }
    });
    p->count = first[p->nchains];
    *count = p->count;
    if (!init) { p->calls += 1; p->rows += p->count; }
    p->stage += 1;
    return BH_OK;
}

int bh_chains_accept(bh_chain_pool *p, const double *logL, const double *misfits)
{
    if (!p || ((!logL || !misfits) && p->count > 0)) return bh::fail_arg_("bh_chains_accept: NULL argument");
    if (p->stage != 1 && p->stage != 3) return bh::fail_arg_("bh_chains_accept: no proposals outstanding");
    std::vector<char> bad(p->nchains, 0);
    char *badp = bad.data();
    if (p->stage == 1) {
        for_chains(p, [=](int i) {
            Chain &c = p->chains[i];
            Proposal &q = c.nodes[0];
            take_proposal(p, c, q, logL, misfits);
            q.took = 1;
            if (!store(p, i, c, q.iter)) badp[i] = 1;
        });
        p->stage = 2;
    } else {
        for_chains(p, [=](int i) { if (!walk(p, i, p->chains[i], logL, misfits)) badp[i] = 1; });
        long slowest = p->cfg.iter_main, sum = 0;
        for (const Chain &c : p->chains) { slowest = std::min(slowest, c.iter); sum += c.iter; }
        p->advanced = sum + (long)p->nchains * p->cfg.iter_burnin;
        p->iiter = slowest;
        p->stage = 2;
    }
    for (int i = 0; i < p->nchains; i++)
        if (bad[i]) {
            p->failed = 1;
            return bh::fail_arg_("bh_chains_accept: a chain accepted more models than its storage holds "
                                 "(nmodels = iterations * max(acceptance) / 100, like the reference's arrays)");
        }
    return BH_OK;
}

int bh_chains_moves(const bh_chain_pool *p, int *move)
{
    if (!p || !move) return bh::fail_arg_("bh_chains_moves: NULL argument");
    for (const Chain &c : p->chains)
        for (int k = 0; k < c.nnodes; k++)
            if (c.nodes[k].slot >= 0) move[c.nodes[k].slot] = c.nodes[k].move;
    return BH_OK;
}

int bh_chains_accepted(const bh_chain_pool *p, int *flag)
{
    if (!p || !flag) return bh::fail_arg_("bh_chains_accepted: NULL argument");
    for (const Chain &c : p->chains)
        for (int k = 0; k < c.nnodes; k++)
            if (c.nodes[k].slot >= 0) flag[c.nodes[k].slot] = c.nodes[k].took;
    return BH_OK;
}

int bh_chains_iterations(const bh_chain_pool *p, long *iter)
{
    if (!p || !iter) return bh::fail_arg_("bh_chains_iterations: NULL argument");
    for (int i = 0; i < p->nchains; i++) iter[i] = p->chains[i].iter;
    return BH_OK;
}

int bh_chains_counters(const bh_chain_pool *p, long *naccepted, double *propdist, double *accepted,
                       double *proposed)
{
    if (!p) return bh::fail_arg_("bh_chains_counters: NULL pool");
    for (int i = 0; i < p->nchains; i++) {
        const Chain &c = p->chains[i];
        if (naccepted) naccepted[i] = c.nstored;
        for (int k = 0; k < 5; k++) {
            if (propdist) propdist[i * 5 + k] = c.propdist[k];
            if (accepted) accepted[i * 5 + k] = c.accepted[k];
            if (proposed) proposed[i * 5 + k] = c.proposed[k];
        }
    }
    return BH_OK;
}

int bh_chains_current(const bh_chain_pool *p, int ci, int *nnuclei, double *model, double *noise,
                      double *vpvs, double *like, double *misfits)
{
    if (!p || ci < 0 || ci >= p->nchains) return bh::fail_arg_("bh_chains_current: bad chain index");
    const Chain &c = p->chains[ci];
    if (nnuclei) *nnuclei = c.n;
    if (model)
        for (int i = 0; i < c.n; i++) { model[i] = c.vs[i]; model[c.n + i] = c.z[i]; }
    if (noise) std::memcpy(noise, c.noise, 2 * p->cfg.ntargets * sizeof(double));
    if (vpvs) *vpvs = c.vpvs;
    if (like) *like = c.like;
    if (misfits) std::memcpy(misfits, c.misfits, (p->cfg.ntargets + 1) * sizeof(double));
    return BH_OK;
}

int bh_chains_get_rng(const bh_chain_pool *p, int ci, unsigned *key, int *pos, int *has_gauss, double *gauss)
{
    if (!p || ci < 0 || ci >= p->nchains || !key || !pos || !has_gauss || !gauss)
        return bh::fail_arg_("bh_chains_get_rng: bad argument");
    const Rng &r = p->chains[ci].rng;
    std::memcpy(key, r.key, sizeof(r.key));
    *pos = r.pos; *has_gauss = r.has_gauss; *gauss = r.gauss;
    return BH_OK;
}

int bh_chains_set_rng(bh_chain_pool *p, int ci, const unsigned *key, int pos, int has_gauss, double gauss)
{
    if (!p || ci < 0 || ci >= p->nchains || !key || pos < 0 || pos > 624)
        return bh::fail_arg_("bh_chains_set_rng: bad argument");
    Rng &r = p->chains[ci].rng;
    std::memcpy(r.key, key, sizeof(r.key));
    r.pos = pos; r.has_gauss = has_gauss ? 1 : 0; r.gauss = gauss;
    return BH_OK;
}

int bh_chains_draw(bh_chain_pool *p, int ci, int kind, double a, double b, int n, double *out)
{
    if (!p || ci < 0 || ci >= p->nchains || !out || n < 0 || kind < 0 || kind > 2)
        return bh::fail_arg_("bh_chains_draw: bad argument");
    Rng &r = p->chains[ci].rng;
    for (int i = 0; i < n; i++)
        out[i] = kind == 0 ? r.uniform(a, b) : kind == 1 ? r.normal(a, b) : (double)r.randint((long)a, (long)b);
    return BH_OK;
}

}  // extern "C"
