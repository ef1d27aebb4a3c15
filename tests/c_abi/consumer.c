/* consumer.c -- a plain C99 client of libbayhunter_amd.so (tests/test_c_abi.py).
 * Proves that include/bayhunter_amd.h is a C header (no C++, no HIP, no torch types) and that the
 * two drop-in entry points are callable from the language a cgo/JNI/FFI stub would use.
 * usage: consumer            -> version + argument validation only (no GPU needed)
 *        consumer run        -> tutorial model through bh_surfdisp96 and bh_synrf, values on stdout */
#include <stdio.h>
#include <string.h>
#include "bayhunter_amd.h"

int main(int argc, char **argv)
{
    int ndev = -1, err = -1, rc, k;
    printf("%s\n", bh_version());
    if (bh_device_count(&ndev) != BH_OK) return 2;
    printf("devices %d\n", ndev);
    {   /* argument validation happens before any device work */
        float m[4] = {5.f, 23.f, 8.f, 0.f};
        double t[1] = {1.0}, cg[1];
        rc = bh_surfdisp96(m, m, m, m, 101, 0, 2, 1, 0, 1, t, cg, &err);
        if (rc != BH_ERR_ARG) { printf("expected BH_ERR_ARG, got %d\n", rc); return 3; }
        printf("arg check ok: %s\n", bh_last_error());
        {   /* the same through the reference's own Fortran symbol (surfdisp96.f:55-56: all by reference):
             * a failure of the library shows as err = 100 + BH_ERR_* (and a line on stderr) */
            int nl = 101, zero = 0, one = 1, two = 2;
            err = -1;
            surfdisp96_(m, m, m, m, &nl, &zero, &two, &one, &zero, &one, t, cg, &err);
            if (err != 100 + BH_ERR_ARG) { printf("surfdisp96_: expected err %d, got %d\n", 100 + BH_ERR_ARG, err); return 3; }
            printf("literal names ok\n");
        }
    }
    {   /* the chain-pool loop from C (host code only): propose -> [likelihoods] -> accept.  The
         * "likelihood" here is a made-up function of the proposal; real callers run bh_swd_batch /
         * bh_rf_batch / bh_likelihood_batch on `packed` in between. */
        enum { NC = 4, LMAX = 8, NM = 64 };
        static float models[NC * NM * 2 * 6], misfits[NC * NM * 2], likes[NC * NM], noise[NC * NM * 2], vpvs[NC * NM];
        static double iter[NC * NM], packed[NC * 4 * LMAX], pnoise[NC * 2], logL[NC], mis[NC * 2];
        int nlay[NC], chain[NC], count = 0, it = 0, i;
        long nacc[NC];
        unsigned seeds[NC] = {1, 2, 3, 4};
        bh_chain_config cfg;
        bh_chain_storage st;
        bh_chain_pool *pool = NULL;
        memset(&cfg, 0, sizeof(cfg));
        cfg.ntargets = 1; cfg.layers_min = 1; cfg.layers_max = 5;
        cfg.vs_min = 2; cfg.vs_max = 5; cfg.z_min = 0; cfg.z_max = 60;
        cfg.vpvs_fixed = 0; cfg.vpvs_min = 1.5; cfg.vpvs_max = 2.0;
        cfg.propdist[0] = .05; cfg.propdist[1] = 1.; cfg.propdist[2] = .1; cfg.propdist[3] = .005; cfg.propdist[4] = .01;
        cfg.acceptance[0] = 40; cfg.acceptance[1] = 45; cfg.iter_burnin = 30; cfg.iter_main = 30;
        cfg.noise_fixed[0] = 1; cfg.noise_lo[0] = cfg.noise_hi[0] = 0.;          /* corr fixed at 0    */
        cfg.noise_fixed[1] = 0; cfg.noise_lo[1] = 1e-5; cfg.noise_hi[1] = 0.05;   /* sigma free        */
        st.nmodels = NM; st.models = models; st.misfits = misfits; st.likes = likes; st.noise = noise;
        st.vpvs = vpvs; st.iter = iter;
        if (bh_chains_create(&cfg, NC, seeds, &st, &pool) != BH_OK) { printf("create: %s\n", bh_last_error()); return 5; }
        while (!bh_chains_done(pool)) {
            if (bh_chains_propose(pool, LMAX, packed, nlay, pnoise, chain, &count) != BH_OK) return 6;
            for (i = 0; i < count; i++) {                /* toy target: 3.5 km/s in the top layer */
                double d = packed[i * 4 * LMAX + 2 * LMAX] - 3.5;
                logL[i] = -50. * d * d; mis[2 * i] = mis[2 * i + 1] = d < 0 ? -d : d;
            }
            if (bh_chains_accept(pool, logL, mis) != BH_OK) { printf("accept: %s\n", bh_last_error()); return 7; }
            it++;
        }
        if (bh_chains_counters(pool, nacc, NULL, NULL, NULL) != BH_OK) return 8;
        printf("chains ok: %d rounds, iteration %ld, accepted %ld %ld %ld %ld\n", it, bh_chains_iteration(pool),
               nacc[0], nacc[1], nacc[2], nacc[3]);
        bh_chains_destroy(pool);
        {   /* the same chains with a look-ahead of LA proposals per chain and call: staging arrays of bh_chains_rows()
             * rows, fewer rounds, the same accepted models */
            enum { LA = 6, NR = NC * LA };
            static float models2[NC * NM * 2 * 6], misfits2[NC * NM * 2], likes2[NC * NM], noise2[NC * NM * 2], vpvs2[NC * NM];
            static double iter2[NC * NM], packed2[NR * 4 * LMAX], pnoise2[NR * 2], logL2[NR], mis2[NR * 2];
            int nlay2[NR], chain2[NR], it2 = 0;
            long nacc2[NC], calls = 0, iters = 0, rows = 0;
            st.models = models2; st.misfits = misfits2; st.likes = likes2; st.noise = noise2; st.vpvs = vpvs2; st.iter = iter2;
            if (bh_chains_create(&cfg, NC, seeds, &st, &pool) != BH_OK) return 9;
            if (bh_chains_set_lookahead(pool, LA) != BH_OK || bh_chains_rows(pool) != NR || bh_chains_lookahead(pool) != LA) return 10;
            while (!bh_chains_done(pool)) {
                if (bh_chains_propose(pool, LMAX, packed2, nlay2, pnoise2, chain2, &count) != BH_OK || count > NR) return 11;
                for (i = 0; i < count; i++) {
                    double d = packed2[i * 4 * LMAX + 2 * LMAX] - 3.5;
                    logL2[i] = -50. * d * d; mis2[2 * i] = mis2[2 * i + 1] = d < 0 ? -d : d;
                }
                if (bh_chains_accept(pool, logL2, mis2) != BH_OK) { printf("accept: %s\n", bh_last_error()); return 12; }
                it2++;
            }
            if (bh_chains_counters(pool, nacc2, NULL, NULL, NULL) != BH_OK || bh_chains_advance(pool, &calls, &iters, &rows) != BH_OK) return 13;
            if (memcmp(nacc, nacc2, sizeof(nacc)) || memcmp(likes, likes2, sizeof(likes)) || memcmp(models, models2, sizeof(models)) ||
                memcmp(iter, iter2, sizeof(iter)) || iters != NC * 60 || calls != it2 - 1 || it2 >= it) {
                printf("look-ahead changed the chains (%d rounds, %ld iterations)\n", it2, iters);
                return 14;
            }
            printf("look-ahead ok: %d rounds instead of %d, %ld rows\n", it2, it, rows);
            bh_chains_destroy(pool);
        }
    }
    if (argc > 1 && strcmp(argv[1], "run") == 0) {
        /* tutorial model st3 (tutorial/create_testdata.py:13-17), vp = 1.73 vs, rho = .77 + .32 vp */
        float h[4] = {5.f, 23.f, 8.f, 0.f}, vs[4] = {2.7f, 3.6f, 3.8f, 4.4f}, vp[4], rho[4];
        double hd[4], vsd[4] = {2.7, 3.6, 3.8, 4.4}, vpd[4], rhod[4], z[4] = {0., 5., 28., 36.};
        double qp[4] = {500., 500., 500., 500.}, qs[4] = {225., 225., 225., 225.};
        double per[21], cg[21], rf[512];
        for (k = 0; k < 4; k++) {
            vpd[k] = vsd[k] * 1.73; rhod[k] = vpd[k] * 0.32 + 0.77; hd[k] = h[k];
            vp[k] = (float)vpd[k]; rho[k] = (float)rhod[k];
        }
        for (k = 0; k < 21; k++) per[k] = 1.0 + 2.0 * k;
        rc = bh_surfdisp96(h, vp, vs, rho, 4, 0, 2, 1, 0, 21, per, cg, &err);
        if (rc != BH_OK) { printf("bh_surfdisp96 failed: %s\n", bh_last_error()); return 4; }
        printf("err %d\n", err);
        for (k = 0; k < 21; k++) printf("cg %.17g\n", cg[k]);
        {
            double nsvp = vpd[0], nsvs = vsd[0], kk = nsvp / nsvs, sigma = (2 - kk * kk) / (2 - 2 * kk * kk);
            rc = bh_synrf(512, 5.0, 5.0, 6.4, 1.0, nsvs, sigma, 0, 4, z, vpd, vsd, rhod, qp, qs, NULL, NULL, rf);
        }
        if (rc != BH_OK) { printf("bh_synrf failed: %s\n", bh_last_error()); return 5; }
        for (k = 0; k < 201; k++) printf("rf %.17g\n", rf[k]);
        (void)hd;
        {   /* the literal FFI names the reference's f2py / Cython glue binds */
            int nl = 4, flsph = 0, iwave = 2, mode = 1, igr = 0, kmax = 21;
            double cg2[60], fz[512], fr[512], rf2[512];
            double kk = vpd[0] / vsd[0], sigma = (2 - kk * kk) / (2 - 2 * kk * kk);
            err = -1;
            surfdisp96_(h, vp, vs, rho, &nl, &flsph, &iwave, &mode, &igr, &kmax, per, cg2, &err);
            printf("err2 %d\n", err);
            for (k = 0; k < 21; k++) printf("cg2 %.17g\n", cg2[k]);
            rc = synrf_cwrap(512, 5.0, 5.0, 6.4, 1.0, vsd[0], sigma, 0, 4, z, vpd, vsd, rhod, qp, qs, fz, fr, rf2);
            printf("synrf_cwrap returned %d\n", rc);
            for (k = 0; k < 201; k++) printf("rf2 %.17g\n", rf2[k]);
            printf("fz0 %.17g fr0 %.17g\n", fz[25], fr[25]);
        }
    }
    return 0;
}
