// Sanitizer driver for the DEVICE solver cores compiled for the host (tests/hostsim/hostsim.cpp
// includes swd_core.h, swd_team.h, rf_core.h under BH_HOSTSIM): built with
// g++ -fsanitize=address,undefined by tests/test_sanitizers.py and run over pseudo-random models --
// 1 to 100 layers, low-velocity zones, water layer, all wave types, modes 1-3, flat/spherical,
// 1 to 60 periods, the team replays at several widths (which must agree bitwise), receiver functions at
// four transform lengths
// and with layer-dependent Q.  Out-of-range indexing of the layer images, the Neville tables or the
// FFT buffers, and undefined arithmetic, would show here; GPU sanitizers are not available.
#include <cmath>
#include <cstdio>
#include <vector>

#define BH_HOSTSIM 1
#include "../hostsim/hostsim.cpp"

static unsigned long long g_state = 88172645463325252ull;
static double rnd()
{
    g_state ^= g_state << 13; g_state ^= g_state >> 7; g_state ^= g_state << 17;
    return (double)(g_state >> 11) / 9007199254740992.0;
}

int main()
{
    double acc = 0;
    long calls = 0;
    for (int trial = 0; trial < 220; trial++) {
        int L = trial < 8 ? (trial % 2 ? 100 : 1) : 1 + (int)(rnd() * (trial % 5 == 0 ? 60 : 14));
        std::vector<float> h(L), vp(L), vs(L), rho(L);
        std::vector<double> hd(L), vpd(L), vsd(L), rhod(L), qp(L), qs(L);
        double v = 2.0 + rnd();
        for (int i = 0; i < L; i++) {
            v += (rnd() - (trial % 3 == 0 ? 0.45 : 0.1)) * 0.6;          // every third model: LVZ
            if (v < 1.2) v = 1.2;
            if (v > 5.2) v = 5.2;
            vsd[i] = v; vpd[i] = 1.73 * v; rhod[i] = 0.77 + 0.32 * vpd[i];
            hd[i] = (i == L - 1) ? 0.0 : 0.1 + rnd() * (L > 30 ? 3.0 : 12.0);
            qp[i] = 400 + 30 * i; qs[i] = 200 - i;
        }
        if (trial % 17 == 0 && L > 2) { vsd[0] = 0.0; vpd[0] = 1.5; rhod[0] = 1.03; }   // water layer
        for (int i = 0; i < L; i++) { h[i] = (float)hd[i]; vp[i] = (float)vpd[i]; vs[i] = (float)vsd[i]; rho[i] = (float)rhod[i]; }
        int nper = trial % 11 == 0 ? 60 : 1 + (int)(rnd() * 25);
        std::vector<double> per(nper), cg(nper);
        for (int k = 0; k < nper; k++) per[k] = 0.5 + 45.0 * k / nper + rnd() * 0.3;
        for (int iw = 1; iw <= 2; iw++)
            for (int ig = 0; ig <= 1; ig++) {
                int mode = 1 + trial % 3, fl = (trial / 3) % 2;
                long nc = 0, ns = 0, nr = 0;
                int e = hs_surfdisp96(h.data(), vp.data(), vs.data(), rho.data(), L, fl, iw, mode, ig, nper, per.data(),
                                      cg.data(), &nc);
                for (double x : cg) acc += x;
                calls += nc + e;
                static const int widths[5] = {64, 32, 16, 8, 7};
                e = hs_surfdisp96_team(h.data(), vp.data(), vs.data(), rho.data(), L, fl, iw, mode, ig, nper, per.data(),
                                       cg.data(), widths[trial % 5], &nc, &ns, &nr);
                for (double x : cg) acc += x;
                calls += nr + e;
                // the wide-team replay (speculation across root searches, value-matched consumption,
                // NevMem tables, slot layout): 64, 128, 256, 512 virtual lanes
                static const int wide[4] = {64, 128, 256, 512};
                std::vector<double> cg2(nper);
                int e2 = hs_surfdisp96_teamw(h.data(), vp.data(), vs.data(), rho.data(), L, fl, iw, mode, ig, nper,
                                             per.data(), cg2.data(), wide[trial % 4], &nc, &ns, &nr);
                if (e2 != e) { std::printf("wide replay: err %d vs %d (trial %d)\n", e2, e, trial); return 1; }
                for (int k = 0; k < nper; k++)
                    if (!(cg2[k] == cg[k]) && !(cg2[k] != cg2[k] && cg[k] != cg[k])) {
                        std::printf("wide replay differs (trial %d, period %d)\n", trial, k);
                        return 1;
                    }
                calls += nr;
            }
        if (vsd[0] > 0 && trial % 2 == 0) {
            static const int ns[4] = {64, 256, 512, 2048};
            int nsamp = ns[trial % 4];
            std::vector<double> rf(nsamp);
            hs_rf(L, hd.data(), vpd.data(), vsd.data(), rhod.data(), trial % 4 ? nullptr : qp.data(),
                  trial % 4 ? nullptr : qs.data(), 4.0 + 10.0 * rnd(), 0.8 + 2.0 * rnd(), nsamp, 5.0, 5.0,
                  trial % 6 ? -1.0 : 3.0, trial % 8 == 0 ? 1 : 0, nsamp / 2, rf.data());
            for (int i = 0; i < nsamp / 2; i++) if (rf[i] == rf[i]) acc += rf[i];
        }
    }
    std::printf("cores ok %ld %.9g\n", calls, acc);
    return 0;
}
