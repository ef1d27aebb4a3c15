/*
 * oracle_port.h -- CPU restatement ("port") of the reference's native forward solvers.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under bayhunter_amd/ may include, link or call this.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
 *
 * Pinned against (a) the reference's own native sources compiled into oracle/_ref/ in the
 * development container (bit-identical, see tests/test_oracle.py::test_live_against_reference and
 * tests/golden/make_golden.py) and (b) the reference's shipped tutorial data
 * (tests/golden/tutorial_observed/, 4-decimal files).
 *
 * Plain C (gnu11), no FMA contraction, no fast-math: build with oracle/Makefile.
 */
#ifndef BH_ORACLE_PORT_H
#define BH_ORACLE_PORT_H

#ifdef __cplusplus
extern "C" {
#endif

#define BHO_NL 100 /* surfdisp96.f:60  NL  */
#define BHO_NP 60  /* surfdisp96.f:62  NP  */

/* Restates `subroutine surfdisp96` (surfdisp96.f:55-360).  Same argument meaning as the f2py
 * symbol (surf96_modsw.py:116): model arrays real*4[nlayer], periods/velocities real*8[kmax].
 * Returns err (0 ok, 1 = fundamental-mode root not found; failed and later periods zero-filled).
 * n_dltar (may be NULL) receives the number of period-equation evaluations (SURVEY 8d N_dltar). */
int bho_surfdisp96(const float *thkm, const float *vpm, const float *vsm, const float *rhom,
                   int nlayer, int iflsph, int iwave, int mode, int igr, int kmax,
                   const double *t, double *cg, long *n_dltar);

/* Batched convenience over fp64 SoA inputs [B][Lmax] (row-major), casting to fp32 like f2py does
 * (surf96_modsw.py:68-82,116).  out[B][kmax], err[B].  OpenMP over models. */
void bho_surfdisp96_batch(int B, int Lmax, const int *nlay, const double *h, const double *vp,
                          const double *vs, const double *rho, int iflsph, int iwave, int mode,
                          int igr, int kmax, const double *t, double *out, int *err,
                          long *n_dltar_total, int nthreads);

/* Restates `synrf_cwrap` (rfmini/wrap.cpp:57-80) and everything below it
 * (synrf.cpp:16-55, model.cpp:223-251, greens.cpp non-PD branch, fork.cpp).
 * fz/fr may be NULL (BayHunter discards them, rfmini_modrf.py:134-142). Always returns 1. */
int bho_synrf(int nsamp, double fsamp, double tshift, double p, double a, double nsv,
              double sigma, int waveno, int nlay, const double *z, const double *vp,
              const double *vs, const double *rh, const double *qp, const double *qs,
              double *fz, double *fr, double *rf);

/* Batched: restates RFminiModRF.compute_rf (rfmini_modrf.py:99-142) per model: z = shifted
 * cumsum(h), poisson from the top layer, nsv = vs[0] unless nsv_override > 0, qp=500, qs=225.
 * h,vp,vs,rho fp64 [B][Lmax]; rf out [B][nout] (first nout samples). OpenMP over models. */
void bho_rf_batch(int B, int Lmax, const int *nlay, const double *h, const double *vp,
                  const double *vs, const double *rho, double p, double gauss, int nsamp,
                  double fsamp, double tshift, double nsv_override, int waveno, int nout,
                  double *rf, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
