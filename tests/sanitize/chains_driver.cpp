// Sanitizer driver for the host side of the chain pool (tests/test_sanitizers.py): chains.cpp is
// compiled with g++ -fsanitize=address,undefined or -fsanitize=thread together with this file and
// run on the CPU: several pools of different size stepped alternately (different thread counts per
// job), many iterations, a made-up likelihood.  Prints a checksum of the stored samples.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/bayhunter_amd.h"

namespace bh {
static thread_local std::string g_err;
int fail_arg_(const char *what) { g_err = what; return BH_ERR_ARG; }
}
extern "C" const char *bh_last_error(void) { return bh::g_err.c_str(); }

struct Pool {
    int n, L;
    long nm;
    std::vector<float> models, misfits, likes, noise, vpvs;
    std::vector<double> iter, packed, pnoise, logL, mis;
    std::vector<int> nlay, chain;
    std::vector<unsigned> seeds;
    bh_chain_pool *p = nullptr;
    Pool(int nchains, long iters, int lookahead = 1) : n(nchains), L(12), nm(iters + 2)
    {
        const int n1 = nchains;                       // chains; n: rows of the staging arrays
        n = nchains * lookahead;
        bh_chain_config c;
        std::memset(&c, 0, sizeof(c));
        c.ntargets = 2; c.layers_min = 1; c.layers_max = 10;
        c.vs_min = 2; c.vs_max = 5; c.z_min = 0; c.z_max = 60;
        c.vpvs_fixed = 0; c.vpvs_min = 1.5; c.vpvs_max = 2.0;
        c.has_mohoest = 1; c.moho_mean = 35; c.moho_std = 4;
        c.thickmin = 0.1; c.has_lvz = 1; c.lvz = 0.2; c.has_hvz = 1; c.hvz = 0.5;
        const double pd[5] = {0.05, 1.0, 0.1, 0.005, 0.01};
        std::memcpy(c.propdist, pd, sizeof(pd));
        c.acceptance[0] = 40; c.acceptance[1] = 45; c.iter_burnin = iters / 2; c.iter_main = iters - iters / 2;
        c.noise_fixed[0] = 1; c.noise_lo[0] = c.noise_hi[0] = 0.0;
        c.noise_lo[1] = 1e-5; c.noise_hi[1] = 0.05;
        c.noise_lo[2] = 0.3; c.noise_hi[2] = 0.9;
        c.noise_lo[3] = 1e-5; c.noise_hi[3] = 0.05;
        const int W = 2 * (c.layers_max + 1), T = c.ntargets;
        models.assign((size_t)n1 * nm * W, NAN); misfits.assign((size_t)n1 * nm * (T + 1), NAN);
        likes.assign((size_t)n1 * nm, NAN); noise.assign((size_t)n1 * nm * 2 * T, NAN); vpvs.assign((size_t)n1 * nm, NAN);
        iter.assign((size_t)n1 * nm, NAN);
        packed.assign((size_t)n * 4 * L, 0); pnoise.assign((size_t)n * 2 * T, 0); logL.assign(n, 0); mis.assign((size_t)n * (T + 1), 0);
        nlay.assign(n, 0); chain.assign(n, 0); seeds.resize(n1);
        for (int i = 0; i < n1; i++) seeds[i] = (unsigned)(i * 7 + 1) % 1000;
        bh_chain_storage st = {nm, models.data(), misfits.data(), likes.data(), noise.data(), vpvs.data(), iter.data()};
        if (bh_chains_create(&c, n1, seeds.data(), &st, &p) != BH_OK) { std::printf("create: %s\n", bh_last_error()); std::exit(2); }
        if (bh_chains_set_lookahead(p, lookahead) != BH_OK || bh_chains_rows(p) != n) { std::printf("lookahead: %s\n", bh_last_error()); std::exit(2); }
    }
    ~Pool() { bh_chains_destroy(p); }
    bool step()
    {
        if (bh_chains_done(p)) return false;
        int count = 0;
        if (bh_chains_propose(p, L, packed.data(), nlay.data(), pnoise.data(), chain.data(), &count) != BH_OK) {
            std::printf("propose: %s\n", bh_last_error()); std::exit(3);
        }
        for (int k = 0; k < count; k++) {
            const double *row = packed.data() + (size_t)k * 4 * L;
            double d = row[2 * L] - 3.2 + 0.01 * nlay[k] + 0.001 * row[0];
            logL[k] = -60.0 * d * d - 2.0 * pnoise[(size_t)k * 4 + 3];
            mis[(size_t)k * 3] = mis[(size_t)k * 3 + 1] = std::fabs(d); mis[(size_t)k * 3 + 2] = 2 * std::fabs(d);
        }
        if (bh_chains_accept(p, logL.data(), mis.data()) != BH_OK) { std::printf("accept: %s\n", bh_last_error()); std::exit(4); }
        return true;
    }
    double checksum() const
    {
        double s = 0;
        for (float v : likes) if (v == v) s += v;
        for (double v : iter) if (v == v) s += v;
        return s;
    }
};

int main(int argc, char **argv)
{
    int threads = argc > 1 ? std::atoi(argv[1]) : 8;
    // "ascending": the smallest pool steps first, so the helper-thread set GROWS after the first
    // job has been published (a worker created then must not react to the earlier generations)
    const bool ascending = argc > 2 && std::string(argv[2]) == "ascending";
    // "lookahead" (third argument): the pools draw 2, 5 and 16 proposals per chain and call (bh_chains_set_lookahead) --
    // the checksums are those of one proposal per call
    const bool ahead = argc > 3 && std::string(argv[3]) == "lookahead";
    Pool a(3000, 60, ahead ? 2 : 1), b(700, 40, ahead ? 5 : 1), c(130, 90, ahead ? 16 : 1);        // 8-, 5- and 1-part jobs interleave
    for (Pool *q : {&a, &b, &c}) bh_chains_set_threads(q->p, threads);
    std::vector<Pool *> order = {&a, &b, &c};
    if (ascending) order = {&c, &b, &a};
    bool more = true;
    while (more) {
        more = false;
        for (Pool *q : order) more = q->step() || more;
    }
    std::printf("checksum %.6f %.6f %.6f\n", a.checksum(), b.checksum(), c.checksum());
    return 0;
}
