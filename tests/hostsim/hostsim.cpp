// hostsim.cpp -- TEST INFRASTRUCTURE ONLY.
// Compiles the device solver cores (bayhunter_amd/csrc/*_core.h) with g++ and runs them lane by
// lane on the CPU, so that the control-flow transformation of the kernels (state machine around a
// single period-equation call site; task-parallel reflectivity + LDS FFT) can be checked bit for
// bit against the oracle without a GPU.  Never loaded by the bayhunter_amd package.
#define BH_HOSTSIM 1
#include <cstring>
#include <vector>
#include "../../bayhunter_amd/csrc/swd_core.h"
#include "../../bayhunter_amd/csrc/swd_team.h"
#include "../../bayhunter_amd/csrc/rf_core.h"
#include "../../bayhunter_amd/csrc/rf_host.h"

namespace {
struct HostLay {
    float *pd, *pa, *pb, *pr;
    float d(int i) const { return pd[i]; }
    float a(int i) const { return pa[i]; }
    float b(int i) const { return pb[i]; }
    float rho(int i) const { return pr[i]; }
    void set_d(int i, float v) { pd[i] = v; }
    void set_a(int i, float v) { pa[i] = v; }
    void set_b(int i, float v) { pb[i] = v; }
    void set_rho(int i, float v) { pr[i] = v; }
};
}  // namespace

namespace {
// task source of the host replay: hands out its one model once (the GPU's is an atomic queue)
struct OneTask {
    HostLay src;
    int nlayer, taken = 0, err = -1;
    double *out, *cws, *cbws;
    int next(HostLay &lay, double *&o, double *&c, double *&cb)
    {
        if (taken) return 0;
        taken = 1;
        lay = src;
        o = out; c = cws; cb = cbws;
        return nlayer;
    }
    void done(int e) { err = e; }
    void sphere(HostLay &lay, int mmax, int ifunc) { bh::swd_sphere(lay, mmax, ifunc); }
    void put(bh::SwdState &S, int k, int kmax, float v) { bh::swd_put_direct(S, k, kmax, v); }
    void fill_zero(bh::SwdState &S, int k, int kmax) { bh::swd_zero_direct(S, k, kmax); }
};
// task source of a consistency re-run that must not store anything
struct NullSrc {
    int next(HostLay &, double *&, double *&, double *&) { return 0; }
    void done(int) {}
    void sphere(HostLay &, int, int) {}
    void put(bh::SwdState &, int, int, float) {}
    void fill_zero(bh::SwdState &, int, int) {}
};
}  // namespace

extern "C" int hs_surfdisp96(const float *thkm, const float *vpm, const float *vsm, const float *rhom,
                             int nlayer, int iflsph, int iwave, int mode, int igr, int kmax,
                             const double *t, double *cg, long *ncalls)
{
    std::vector<float> d(thkm, thkm + nlayer), a(vpm, vpm + nlayer), b(vsm, vsm + nlayer),
        r(rhom, rhom + nlayer);
    HostLay lay{nullptr, nullptr, nullptr, nullptr};
    bh::SwdTargetDev tg{iwave, igr, mode, iflsph, kmax, 0, 0, 0};
    std::vector<double> cws(kmax > 0 ? kmax : 1), cbws(kmax > 0 ? kmax : 1);
    OneTask src{HostLay{d.data(), a.data(), b.data(), r.data()}, nlayer, 0, -1, cg, cws.data(), cbws.data()};
    bh::swd_lane(lay, src, tg, t, 1, ncalls);
    return src.err;
}

// CPU replay of the team kernel (swd_team.h): `nlanes` virtual lanes per search, phases separated
// like the __syncthreads() of swd_team_kernel.  *ncalls counts consumed (= reference) evaluations,
// *nspec all evaluations incl. discarded speculation, *nrounds the rounds.
extern "C" int hs_surfdisp96_team(const float *thkm, const float *vpm, const float *vsm, const float *rhom,
                                  int nlayer, int iflsph, int iwave, int mode, int igr, int kmax,
                                  const double *t, double *cg, int nlanes, long *ncalls, long *nspec,
                                  long *nrounds)
{
    std::vector<float> d(thkm, thkm + nlayer), a(vpm, vpm + nlayer), b(vsm, vsm + nlayer),
        r(rhom, rhom + nlayer);
    HostLay lay{nullptr, nullptr, nullptr, nullptr};
    bh::SwdTargetDev tg{iwave, igr, mode, iflsph, kmax, 0, 0, 0};
    std::vector<double> cws(kmax > 0 ? kmax : 1), cbws(kmax > 0 ? kmax : 1);
    OneTask src{HostLay{d.data(), a.data(), b.data(), r.data()}, nlayer, 0, -1, cg, cws.data(), cbws.data()};
    std::vector<double> lds(bh::swd_team_lds_doubles(nlayer, nlanes));
    double *mats = lds.data(), *trials = mats + lds.size() - 2 * bh::SWD_TEAM_NT,
           *dels = trials + bh::SWD_TEAM_NT;
    bh::SwdState S;
    bh::swd_state_init(S);
    bh::NevRegs nv;
    bh::swd_nev_init(nv);
    long nc = 0, ns = 0, nr = 0;
    for (;;) {
        bh::swd_driver(S, lay, src, tg, t, 1);
        if (S.st == bh::SWD_ST_DONE) break;
        int nt = bh::swd_team_plan(S, nlanes, trials);
        for (int lane = 0; lane < nlanes; lane++)
            bh::swd_team_assemble(lay, lane, nlanes, iwave, S, nt, trials, mats);
        for (int lane = 0; lane < nlanes; lane++)
            bh::swd_team_chain(lay, lane, iwave, S, nt, trials, mats, dels);
        nc += bh::swd_team_consume(S, nv, nt, trials, dels);
        ns += nt;
        nr++;
    }
    if (ncalls) *ncalls = nc;
    if (nspec) *nspec = ns;
    if (nrounds) *nrounds = nr;
    return src.err;
}

// Optional statistics of the wide-team replay: hist[st][nev][min(used, 15)] counts rounds by the state the
// search was in when the round was planned (ST_A .. ST_MID), nevill's `nev`, and the values consumed.
static long *g_round_hist = nullptr;
extern "C" void hs_teamw_set_histogram(long *hist) { g_round_hist = hist; }

// CPU replay of the wide-team kernel (swd_teamw_kernel): speculation across the end of a root search,
// values consumed by (omega, c) match.  `nlanes` = 64 * W virtual lanes.
namespace {
struct ArrayVals {
    const double *tc, *tom, *dl;
    const bh::TeamwRound *R;
    int find(double om, double c) const
    {
        for (int j = 0; j < R->nt; j++)
            if (tom[j] == om && tc[j] == c) return j;
        return -1;
    }
    double del(int j) const { return dl[j]; }
    double c(int j) const { return tc[j]; }
    void probe(int) const {}
    void count(int, int) const {}
    int go(const bh::SwdState &S, int j) const { return bh::swd_teamw_node(S, *R, *this, j).go; }
    bh::TeamwNode node(const bh::SwdState &S, int j) const { return bh::swd_teamw_node(S, *R, *this, j); }
    int chain_run(const bh::SwdState &S, int first, int dir, int max) const
    {
        int n = 0;
        while (n < max && go(S, first + 2 * n) == dir) n++;
        return n;
    }
    int run(int first, int stride, int count, bool neg) const
    {
        int m = 0;
        for (; m < count; m++) {
            const int j = first + m * stride;
            if (j >= R->nt || tc[j] != tc[j] || std::signbit(dl[j]) != neg) break;
        }
        return m;
    }
};
}  // namespace
extern "C" int hs_surfdisp96_teamw(const float *thkm, const float *vpm, const float *vsm, const float *rhom,
                                   int nlayer, int iflsph, int iwave, int mode, int igr, int kmax,
                                   const double *t, double *cg, int nlanes, long *ncalls, long *nspec,
                                   long *nrounds)
{
    std::vector<float> d(thkm, thkm + nlayer), a(vpm, vpm + nlayer), b(vsm, vsm + nlayer),
        r(rhom, rhom + nlayer);
    HostLay lay{nullptr, nullptr, nullptr, nullptr};
    bh::SwdTargetDev tg{iwave, igr, mode, iflsph, kmax, 0, 0, 0};
    std::vector<double> cws(kmax > 0 ? kmax : 1), cbws(kmax > 0 ? kmax : 1);
    OneTask src{HostLay{d.data(), a.data(), b.data(), r.data()}, nlayer, 0, -1, cg, cws.data(), cbws.data()};
    const int NT = bh::SWD_TEAMW_NT;
    std::vector<double> mats((size_t)bh::swd_mat_off(nlanes <= 64 ? nlanes * nlayer : nlanes > nlayer ? nlanes : nlayer)), tc(NT), tom(NT), dl(NT);
    bh::SwdState S;
    bh::swd_state_init(S);
    double nx[12], ny[12];
    bh::NevMem nv{nx, ny};
    bh::TeamwNext nxt{-1, -1, 0.0, 0, 0, 0.0, 0.0, 0.0};
    bh::swd_nev_init(nv);
    long nc = 0, ns = 0, nr = 0;
    for (;;) {
        bh::swd_driver(S, lay, src, tg, t, 1);
        if (S.st == bh::SWD_ST_DONE) break;
        const int nlm = S.mmax - S.llw;
        // (the device's slots per round: a wide team's by its layers; a team of up to 64 lanes -- one trial per lane,
        // kernels.hip swd_tpl_body and the one-wave team -- has one slot per lane)
        const int cap = nlanes <= 64 ? nlanes : bh::swd_teamw_cap(nlm, nlanes / 64, iwave);
        const bh::TeamwRound R = bh::swd_teamw_round(S, tg, t, cap, nxt);
        const int nt = R.nt;
        if (nt < 1 || nt > NT || nt > (cap > 1 ? cap : 1)) return -100;      // layout invariants
        for (int j = 0; j < nt; j++) bh::swd_teamw_trial(R, S, j, &tc[j], &tom[j]);
        auto same = [](double a, double b) { return a == b || (a != a && b != b); };
        if (!same(tc[0], S.ceval) || !same(tom[0], S.omega)) return -101;
        for (int gi = 0; gi < R.ngrp; gi++) {             // scan trials: consecutive grid points
            const bh::TeamwScan &g = gi ? R.g1 : R.g0;
            if (g.entry >= 0 && (g.entry <= R.nhalf || g.entry >= nt || tc[g.entry] != g.base)) return -103;
            for (int i = 0; i + 1 < g.nscan; i++) {
                const int j0 = g.scan0 + i * g.stride, j1 = j0 + g.stride;
                if (j0 <= R.nhalf - (R.nhalf == 0) || j1 >= nt + g.stride) return -104;
                if (j1 < nt && tc[j1] == tc[j1] && tc[j1] != tc[j0] + (double)0.005f) return -102;
            }
        }
        if (R.ngrp > 1 && (R.g0.scan0 + R.g0.nscan * R.g0.stride > R.g1.entry + (R.g0.stride - 1))) return -105;
        for (int j = 0; j < nt; j++) {
            if (tc[j] != tc[j]) { dl[j] = 0.0; continue; }                    // NaN slot: not evaluated
            for (int rr = 0; rr < nlm; rr++)
                bh::swd_teamw_assemble_one(lay, iwave, S, rr, tc[j], tom[j], mats.data() + bh::swd_mat_off(j * nlm + rr));
            dl[j] = bh::swd_teamw_chain_one(lay, iwave, S, tc[j], tom[j], mats.data(), j * nlm);
        }
        ArrayVals vals{tc.data(), tom.data(), dl.data(), &R};
        // the tree walk must leave the state one swd_control call per node leaves (checked as long as no
        // root search ends inside the round: the driver stores results)
        bh::SwdState Sq = S;
        double qx[12], qy[12];
        std::memcpy(qx, nx, sizeof(qx)); std::memcpy(qy, ny, sizeof(qy));
        bh::NevMem qv{qx, qy};
        const int k0 = S.k, pass0 = S.pass, iq0 = S.iq;
        const int st0 = S.st, nev0 = S.nev;
        const int used_now = bh::swd_teamw_consume(S, nv, lay, src, tg, t, 1, R, vals);
        nc += used_now;
        if (g_round_hist && st0 >= 0 && st0 < 4 && nev0 >= 0 && nev0 < 3)
            g_round_hist[(st0 * 3 + nev0) * 16 + (used_now < 15 ? used_now : 15)]++;
        if (S.k == k0 && S.pass == pass0 && S.iq == iq0 && S.st != bh::SWD_ST_DONE && S.ev == bh::SWD_EV_NONE) {
            NullSrc nsrc;
            bh::swd_teamw_consume(Sq, qv, lay, nsrc, tg, t, 1, R, vals, false);
            auto same = [](double a, double b) { return a == b || (a != a && b != b); };
            if (Sq.st != S.st || Sq.nev != S.nev || Sq.m != S.m || Sq.nctrl != S.nctrl || Sq.idir != S.idir ||
                Sq.nbrk != S.nbrk || !same(Sq.c1, S.c1) || !same(Sq.c2, S.c2) || !same(Sq.c3, S.c3) ||
                !same(Sq.del1, S.del1) || !same(Sq.del2, S.del2) || !same(Sq.del3, S.del3) ||
                !same(Sq.ceval, S.ceval) || !same(Sq.omega, S.omega) ||
                std::memcmp(qx, nx, sizeof(qx)) || std::memcmp(qy, ny, sizeof(qy)))
                return -110;
        }
        ns += nt;
        nr++;
    }
    if (ncalls) *ncalls = nc;
    if (nspec) *nspec = ns;
    if (nrounds) *nrounds = nr;
    return src.err;
}

namespace bh {
// CPU replay of the RF workgroup program (P1..P4) for one model; mirrors rf_kernels.hip.
static int rf_hostsim_model(int nlay, const double *h, const double *vp, const double *vs,
                            const double *rho, const double *qp, const double *qs, double p,
                            double gauss, int nsamp, double fsamp, double tshift, double nsv,
                            int waveno, int nout, double *rf)
{
    RfLaunch P;
    std::memset(&P, 0, sizeof(P));
    rf_fill_launch(P, p, gauss, nsamp, fsamp, tshift, nsv, waveno, nout);
    P.sigma = std::nan("");
    P.Lmax = nlay;
    RfLayout lo = rf_layout(nlay, nsamp);
    std::vector<double> S(lo.per_model, 0.0), tw(2 * (size_t)nsamp, 0.0);
    rf_fill_twiddles(tw.data(), nsamp);
    for (int i = 0; i < nlay; i++) rf_phase1_layer(S.data(), lo, nlay, i, h, vp, vs, rho, qp, qs, 0);
    for (int i = 0; i < nlay; i++) rf_phase2_interface(S.data(), lo, P, nlay, i, vp[0], vs[0]);
    std::vector<cd> spec(P.nfreq);
    std::vector<double> ftab((size_t)RF_FTAB * P.nfreq);
    rf_fill_freq_table(P, ftab.data());
    for (int j = 0; j < P.nfreq; j++)
        spec[j] = j < P.nact ? rf_phase3_task(S.data(), lo, P, nlay, j, rf_freq_load(ftab.data(), j)) : mk(0., 0.);
    double *X = S.data();
    // layout invariant of the kernel: while phase 3 runs, spectrum stores (threads that are done) must
    // not touch the parameter / coefficient region other threads still read
    for (int j = 0; j < P.nfreq; j++)
        if (2 * rf_swz(j) + 1 >= lo.off_par) {
            for (int i = 0; i < nout; i++) rf[i] = std::nan("");
            return -1;
        }
    for (int j = 0; j < P.nfreq; j++) rf_xst(X, j, spec[j]);
    for (int i = nsamp / 2 + 1; i < nsamp; i++) rf_fft_hermitian(X, nsamp, i);
    for (int i = 0; i < nsamp; i++) rf_fft_bitrev_scale(X, nsamp, P.log2n, P.sc, i);
    for (int l = 1; l < nsamp; l <<= 1)
        for (int bf = 0; bf < nsamp / 2; bf++) rf_fft_butterfly(X, tw.data(), l, bf);
    for (int i = 0; i < nout; i++) rf[i] = P.qn * X[2 * rf_swz(i)];
    return 0;
}
}  // namespace bh

extern "C" int hs_rf(int nlay, const double *h, const double *vp, const double *vs, const double *rho,
                     const double *qp, const double *qs, double p, double gauss, int nsamp,
                     double fsamp, double tshift, double nsv, int waveno, int nout, double *rf)
{
    return bh::rf_hostsim_model(nlay, h, vp, vs, rho, qp, qs, p, gauss, nsamp, fsamp, tshift, nsv,
                                waveno, nout, rf);
}

// swd_scan_cell (swd_team.h): closed form against the reference's repeated addition
extern "C" void hs_scan_cell(int n, const double *base, const int *cell, double *b, double *cn)
{
    for (int k = 0; k < n; k++) bh::swd_scan_cell(base[k], cell[k], &b[k], &cn[k]);
}

// accuracy probes for bh_math.h (tests/test_hostsim.py::test_math_accuracy)
extern "C" void hs_sincos(int n, const double *x, double *s, double *c)
{
    for (int i = 0; i < n; i++) bh::bh_sincos(x[i], &s[i], &c[i]);
}
extern "C" void hs_exp(int n, const double *x, double *y)
{
    for (int i = 0; i < n; i++) y[i] = bh::bh_exp(x[i]);
}
