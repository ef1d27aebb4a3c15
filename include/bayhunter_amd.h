/*
 * bayhunter_amd.h -- C ABI of libbayhunter_amd.so, the MI355X (gfx950) forward-modelling engine
 * for BayHunter's per-proposal likelihood hot path.
 *
 * Boundary being replaced (reference file:line, paths relative to the BayHunter tree):
 *   - f2py symbol   err = surfdisp96(thkm,vpm,vsm,rhom,nlayer,iflsph,iwave,mode,igr,kmax,t,cg)
 *                   src/extensions/surfdisp96.f:55-56, called at src/surf96_modsw.py:116-117
 *   - Cython/C      synrf_cwrap(nsamp,fsamp,tshift,p,a,nsv,sigma,waveno,nlay,z,vp,vs,rh,qp,qs,fz,fr,rf)
 *                   src/extensions/rfmini/wrap.cpp:57-63, called through rfmini.pyx:74-114 from
 *                   src/rfmini_modrf.py:134-137
 *
 * Conventions
 *   - plain pointers and sizes only; the caller owns every buffer; no allocation crosses the ABI
 *   - `*_batch` entry points take DEVICE pointers and a HIP stream (hipStream_t passed as void*;
 *     NULL = the default stream) and are asynchronous
 *   - `bh_surfdisp96` / `bh_synrf` take HOST pointers, copy, launch, and synchronise: they are the
 *     drop-in for the two reference symbols above (same argument meaning, same failure semantics)
 *   - return value: BH_OK, or an error code for API misuse / HIP failures.  A model for which the
 *     solver finds no root is NOT an error of the call: it is reported per model in `err[]`
 *     exactly like the reference's `err` (surfdisp96.f:313-316,348-354)
 *   - every function fails with BH_ERR_NO_DEVICE when no gfx950 device is usable: there is no
 *     CPU fallback in this library
 */
#ifndef BAYHUNTER_AMD_H
#define BAYHUNTER_AMD_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BH_OK            0
#define BH_ERR_ARG       1 /* bad argument (sizes, NULL pointers, limits)           */
#define BH_ERR_HIP       2 /* a HIP runtime call failed; see bh_last_error()        */
#define BH_ERR_NO_DEVICE 3 /* no usable HIP device                                  */
#define BH_ERR_WORKSPACE 4 /* workspace too small; see the *_workspace_bytes helper */

#define BH_MAX_LAYERS  100 /* NL, surfdisp96.f:60 */
#define BH_MAX_PERIODS 60  /* NP, surfdisp96.f:62 */
#define BH_MAX_TARGETS 16

/* One surface-wave target = one (wave type, velocity type) dispersion curve, i.e. one SurfDisp
 * plugin instance of the reference (surf96_modsw.py:24-34,48-59). */
typedef struct bh_swd_target {
    int iwave;   /* 1 = Love, 2 = Rayleigh                      (surfdisp96.f:75)      */
    int igr;     /* 0 = phase velocity, >0 = group velocity     (surfdisp96.f:77)      */
    int mode;    /* 1 = fundamental, 2 = first higher, ...      (surfdisp96.f:76)      */
    int iflsph;  /* 0 = flat earth, 1 = earth-flattening        (surfdisp96.f:74)      */
    int nper;    /* number of periods, <= BH_MAX_PERIODS        (kmax, surfdisp96.f:78) */
    int per_off; /* offset of this target's periods in `periods`                       */
    int out_off; /* first column of this target in a model's output row                */
    int _pad;
} bh_swd_target;

/* Receiver-function parameters = RFminiModRF.modelparams + the values derived from the observed
 * time axis (rfmini_modrf.py:26-62,99-142). */
typedef struct bh_rf_params {
    double p;       /* slowness in s/deg                                  ('p')     */
    double gauss;   /* Gauss filter parameter a                           ('gauss') */
    double fsamp;   /* sampling frequency in Hz                                      */
    double tshift;  /* time shift in s                                               */
    double nsv;     /* near-surface S velocity; <= 0: take vs of the top layer ('nsv' None) */
    int    nsamp;   /* FFT length, power of two, 8..4096                             */
    int    waveno;  /* 0 = P, 1 = SV                                       ('wtype') */
    int    nout;    /* number of leading samples returned (= obsx.size)              */
    int    out_off; /* first column of the RF in a model's output row                */
} bh_rf_params;

/* ---- library / device ------------------------------------------------------------------- */
const char *bh_version(void);
const char *bh_last_error(void);          /* text of the last BH_ERR_HIP on this thread     */
int  bh_device_count(int *count);
int  bh_set_device(int device);

/* ---- batched surface-wave dispersion (device pointers) ---------------------------------- */
/* Models: fp64 arrays h, vp, vs, rho holding Lmax values per model, model b at element offset
 * b*model_stride of each pointer, and int32 nlay[B] (layers incl. the half-space,
 * 1 <= nlay <= Lmax <= BH_MAX_LAYERS).  model_stride = Lmax for four separate row-major [B][Lmax]
 * arrays; model_stride = 4*Lmax with vp = h+Lmax, vs = h+2*Lmax, rho = h+3*Lmax for the packed
 * [B][4][Lmax] layout (one contiguous 32*Lmax-byte block per model: what a lane fetches when it
 * pulls a model from the work queue).  Values are rounded to fp32 on load exactly like f2py does
 * for the reference (surf96_modsw.py:68-82).  `targets` is a HOST array.
 * out[b*out_stride + out_off + k], k < nper: phase/group velocity; failed and later periods are 0.
 * err[b*ntargets + t]: the reference's err flag for that (model, target) -- 0, or 1 when no root of
 * the fundamental mode was found -- or BH_MODEL_BAD_DEPTH (2) when nlay[b] is outside 1..Lmax: such a
 * model is not evaluated on a truncated layer stack; its dispersion row (and receiver function) is NaN.
 * workspace: only needed when some target has mode > 1 (bh_swd_workspace_bytes). */
#define BH_MODEL_BAD_DEPTH 2
size_t bh_swd_workspace_bytes(int B, int ntargets, const bh_swd_target *targets);
/* Several kernels compute the same values, bit for bit:
 *   BH_SWD_LANE     one search per lane (persistent lanes, work queue): the throughput form,
 *                   ~1e7 ten-layer searches/s, ~11 ms latency
 *   BH_SWD_TEAM     one wave (64 lanes) per search: speculative bracketing, layer-parallel matrix
 *                   assembly, the bisection tree of a refinement and the start of the next root
 *                   search evaluated ahead (swd_team.h): ~1.0 ms for up to ~1000 ten-layer searches
 *   TEAM128/256/512 2, 4, 8 waves per search: deeper speculation for deep models / few searches
 *   TEAM32/16/8     2, 4, 8 searches per wave: less speculation, more searches resident
 * BH_SWD_AUTO (default) picks by a measured latency / saturation-rate table on searches per call,
 * deepest model and CU count (capi.hip; profiles/r02_team_widths.txt).  Process-wide setting,
 * read once per call. */
#define BH_SWD_AUTO 0
#define BH_SWD_LANE 1
#define BH_SWD_TEAM 2   /* 64 lanes per search: lowest latency                                     */
#define BH_SWD_TEAM32 3 /* 32 lanes per search, two searches per wave                               */
#define BH_SWD_TEAM16 4 /* 16 lanes per search, four per wave: less speculation, more searches/s   */
#define BH_SWD_TEAM8 5  /* 8 lanes per search, eight per wave: layer-parallel assembly only         */
#define BH_SWD_TEAM128 6 /* 2 waves per search: deeper speculation for deep models / few searches   */
#define BH_SWD_TEAM256 7 /* 4 waves per search                                                      */
#define BH_SWD_TEAM512 8 /* 8 waves per search: a handful of deep models                            */
int bh_swd_set_kernel(int mode);
/* The kernel form the last bh_swd_batch / bh_swd_batch_ordered of the calling thread launched: 0 = the
 * lane kernel, otherwise the lanes per search of the team kernel (8 .. 512); -1 before the first call. */
int bh_swd_last_form(void);
/* With several targets BH_SWD_AUTO may give the targets of one call to different forms (the heaviest targets
 * of a latency-bound call to a faster form, launched beside the others on a second stream; the caller's stream
 * continues when all of them are done): forms[t] = the form target t ran on in the calling thread's last call;
 * bh_swd_last_form is then the form of the heaviest target. */
int bh_swd_last_forms(int *forms, int ntargets);
/* Tests and experiments: the calling thread's next calls with exactly `ntargets` targets run target t on
 * forms[t] (at most three different forms; wide forms that do not fit the LDS fall back to narrower ones as
 * with bh_swd_set_kernel).  ntargets = 0 or forms = NULL: back to the library's choice. */
int bh_swd_set_forms(const int *forms, int ntargets);
/* The same choice without a launch (host only, no device needed): forms[t] for a call with B models of at
 * most Lmax layers on a device with `cus` compute units (<= 0: 256). */
int bh_swd_plan_forms(int B, int Lmax, int ntargets, const bh_swd_target *targets, int cus, int *forms);
/* What the caller knows about its NEXT dispersion call on this thread (bh_swd_batch*, bh_swd_plan_forms) and the
 * library cannot see in device memory: the mean layer count of the batch's models (0: unknown -- the deepest model
 * then stands for all, which is right for batches of one depth and pessimistic for a sampler's ragged ones) and the
 * number of such calls in flight on the device together (>= 1; the chain groups of a pool alternate).  The kernel
 * form is chosen with them; results never depend on it.  The evaluation plan and ForwardEngine.upload give the hint
 * themselves. */
int bh_swd_hint(double mean_layers, int concurrent_calls);
int bh_swd_batch(int B, int Lmax, int model_stride, const int *nlay, const double *h,
                 const double *vp, const double *vs, const double *rho, int ntargets,
                 const bh_swd_target *targets,
                 const double *periods, double *out, int out_stride, int *err,
                 void *workspace, size_t workspace_bytes, void *stream);
/* Same, with a processing order: order[i] (device, int[B], a permutation of 0..B-1) is the model
 * the i-th search slot works on; results still land in row order[i].  The lanes of a wave run in
 * lock step, so putting models of similar depth and similar search length next to each other
 * saves 10 % at half a million models (bayhunter_amd/engine.py orders by layer count, then by the
 * S-wave travel time through the stack).  order == NULL is bh_swd_batch.  `order` is trusted: it
 * must be a permutation of 0..B-1 (a repeated index evaluates that model twice and leaves another
 * row unwritten; an index outside the batch reads and writes outside the caller's buffers). */
int bh_swd_batch_ordered(int B, int Lmax, int model_stride, const int *nlay, const double *h,
                         const double *vp, const double *vs, const double *rho, int ntargets,
                         const bh_swd_target *targets, const double *periods, double *out,
                         int out_stride, int *err, const int *order, void *workspace,
                         size_t workspace_bytes, void *stream);

/* Sort keys for that order (device pointers; keys: int[B]): sorting the models by ascending key puts
 * the deepest first, within a depth the predicted longest searches first (by_length != 0; a launch
 * ends with lanes waiting for the last searches, which should be short ones) and within such a class
 * the models by their S-wave travel time.  longest_period: the longest period of the targets [s].
 * order = argsort(keys) is left to the caller (any device sort; bayhunter_amd/engine.py has one). */
int bh_swd_order_keys(int B, int Lmax, int model_stride, const int *nlay, const double *h, const double *vs,
                      double longest_period, int by_length, int *keys, void *stream);

/* ---- batched receiver functions (device pointers) --------------------------------------- */
/* qp/qs may be NULL: 500 / 225 like rfmini_modrf.py:119-120.  Output: the first nout samples of
 * the RF trace at out[b*out_stride + out_off + i].  NaN propagates like in the reference. */
size_t bh_rf_workspace_bytes(int B, int Lmax, const bh_rf_params *par);
/* Number of frequencies (of nsamp/2 + 1) the kernel computes for these parameters: bins whose Gauss
 * filter weight exp(-(w/a)^2/4) (greens.cpp:389-392) is below 3e-19 of its value at w = 0 are set to
 * zero instead (rf_host.h).  0 for invalid parameters.  Host only, no device needed. */
int bh_rf_active_frequencies(const bh_rf_params *par);
int bh_rf_batch(int B, int Lmax, int model_stride, const int *nlay, const double *h,
                const double *vp, const double *vs, const double *rho, const double *qp,
                const double *qs, /* qp/qs: [B][Lmax] (stride Lmax) or NULL */
                const bh_rf_params *par, double *out, int out_stride,
                void *workspace, size_t workspace_bytes, void *stream);

/* ---- Voronoi nuclei -> layered models + prior checks (device pointers) ------------------- */
/* The step in front of the forward path: Model.get_vp_vs_h (src/Models.py:26-52) and
 * SingleChain._validmodel (src/SingleChain.py:330-392) for a batch of proposals.  Input per model:
 * nlay nuclei (vs[i], z_vnoi[i]) sorted by depth, rows of [B][Lmax] arrays, and the chain's vpvs.
 * Output: the packed model block [B][4][Lmax] (h, vp, vs, rho = 0.77 + 0.32*vp, src/Targets.py:319;
 * zero padded) that bh_swd_batch / bh_rf_batch take with model_stride = 4*Lmax, and valid[b] = 1
 * iff the proposal passes every prior check.  NaN fields of the struct mean "None". */
typedef struct bh_model_priors {
    int    layers_min, layers_max; /* priors['layers'], counted without the half-space        */
    double vs_min, vs_max;         /* priors['vs']                                             */
    double z_min, z_max;           /* priors['z']                                              */
    double thickmin;               /* initparams['thickmin']                                   */
    double lowvelperc;             /* initparams['lvz'] or NaN                                 */
    double highvelperc;            /* initparams['hvz'] or NaN                                 */
    double mantle_vs, mantle_vpvs; /* priors['mantle'] = (vs threshold, vp/vs) or NaN, NaN     */
} bh_model_priors;
int bh_voronoi_to_layers(int B, int Lmax, const int *nlay, const double *vs_nuclei,
                         const double *z_nuclei, const double *vpvs, const bh_model_priors *pri,
                         double *model, int *valid, void *stream);

/* ---- fused likelihood (device pointers) ------------------------------------------------- */
/* JointTarget.evaluate's tail (src/Targets.py:322-347) for a batch: per target RMS misfit and
 * Gaussian log-likelihood with one of the four covariance models of Valuation
 * (src/Targets.py:105-173), summed over targets.  Consumes the output rows of bh_swd_batch /
 * bh_rf_batch in place.  Per model b:
 *   logL[b]                      sum over targets of -0.5*(n*log(2*pi) + logdet) - madist/2
 *   misfits[b*(ntargets+1) + t]  RMS of target t;  [.. + ntargets] their sum
 * A model with a non-zero err flag in any of its flag columns gets logL = -1e15 and all misfits
 * 1e15 (src/Targets.py:325-328); NaN data propagates exactly like in the reference. */
#define BH_COV_NOCORR        0 /* get_covariance_nocorr            C^-1 = I/sigma^2                   */
#define BH_COV_NOCORR_SCALED 1 /* get_covariance_nocorr_scalederr  C^-1 = diag(1/(scaled_err*sigma^2)) */
#define BH_COV_EXP           2 /* get_covariance_exp               tridiagonal closed form             */
#define BH_COV_GAUSS         3 /* get_covariance_gauss             fixed dense R^-1 / sigma^2          */
typedef struct bh_like_target {
    int    n;        /* number of data points                                                  */
    int    off;      /* first column of the target in the output row (= offset in yobs too)    */
    int    cov;      /* BH_COV_*                                                               */
    int    aux_off;  /* element offset in `aux`: scaled_err[n] (SCALED) or R^-1[n*n] row-major (GAUSS) */
    double logdet_extra; /* SCALED: log(prod(scaled_err)); GAUSS: log det R; else 0           */
} bh_like_target;
/* yobs: observed data laid out like an output row [row_len]; noise: [B][2*ntargets] (corr, sigma)
 * pairs per target (src/Targets.py:338); err: [B][nflags] int32 flags (may be NULL with nflags 0). */
/* workspace: bh_likelihood_workspace_bytes() bytes of device memory when a BH_COV_GAUSS target is
 * present (its n x n product runs on the FP64 matrix cores and parks d^T R^-1 d there); with a NULL
 * / too small workspace the product falls back to the vector units inside the same call. */
size_t bh_likelihood_workspace_bytes(int B, int ntargets, const bh_like_target *targets);
int bh_likelihood_batch(int B, int ntargets, const bh_like_target *targets, const double *out,
                        int out_stride, const int *err, int nflags, const double *yobs,
                        const double *noise, const double *aux, double *logL, double *misfits,
                        void *workspace, size_t workspace_bytes, void *stream);

/* The same in two stages, for a caller that overlaps them with other work: BH_LIKE_STAGE_GAUSS launches only the
 * dense products of the BH_COV_GAUSS targets into the workspace (it reads the columns of those targets, yobs and
 * aux: nothing else of the arguments needs to be ready), BH_LIKE_STAGE_REST the rest, which consumes them -- the
 * caller orders the two (an evaluation plan runs the first behind the receiver-function kernel on its side
 * stream, beside the dispersion searches).  Needs the workspace.  Both stages at once = bh_likelihood_batch. */
#define BH_LIKE_STAGE_GAUSS 1
#define BH_LIKE_STAGE_REST  2
int bh_likelihood_stage(int stages, int B, int ntargets, const bh_like_target *targets, const double *out,
                        int out_stride, const int *err, int nflags, const double *yobs,
                        const double *noise, const double *aux, double *logL, double *misfits,
                        void *workspace, size_t workspace_bytes, void *stream);

/* ---- single-model drop-ins (host pointers, synchronous) --------------------------------- */
/* Same argument list as the f2py wrapper of `subroutine surfdisp96`; model arrays are real*4 with
 * at least nlayer valid entries, t/cg real*8 with at least kmax entries.  *err as the reference. */
int bh_surfdisp96(const float *thkm, const float *vpm, const float *vsm, const float *rhom,
                  int nlayer, int iflsph, int iwave, int mode, int igr, int kmax,
                  const double *t, double *cg, int *err);
/* Same argument list as synrf_cwrap (wrap.cpp:57-63).  fz and fr (vertical / radial traces, which
 * BayHunter discards, rfmini_modrf.py:134-142) may both be NULL.  Returns BH_OK, not 1. */
int bh_synrf(int nsamp, double fsamp, double tshift, double p, double a, double nsv, double sigma,
             int waveno, int nlay, const double *z, const double *vp, const double *vs,
             const double *rh, const double *qp, const double *qs,
             double *fz, double *fr, double *rf);

/* ---- the reference's literal FFI symbols ------------------------------------------------ */
/* Thin aliases of the two drop-ins above under the names the reference's generated glue binds, so that
 * the f2py wrapper of surfdisp96.f and rfmini.pyx link against this library unchanged:
 *   surfdisp96_   the Fortran symbol of `subroutine surfdisp96` (surfdisp96.f:55-56): thirteen
 *                 arguments, all by reference, no return value; *err as the reference, or 100 + BH_ERR_*
 *                 when the library itself failed (message on stderr and in bh_last_error())
 *   synrf_cwrap   wrap.cpp:57-63, exact signature; returns 1 like the reference (0 and NaN traces when
 *                 the library itself failed) */
void surfdisp96_(const float *thkm, const float *vpm, const float *vsm, const float *rhom,
                 const int *nlayer, const int *iflsph, const int *iwave, const int *mode, const int *igr,
                 const int *kmax, const double *t, double *cg, int *err);
int synrf_cwrap(int nsamp, double fsamp, double tshift, double p, double a, double nsv, double sigma,
                int waveno, int nlay, double *z, double *vp, double *vs, double *rh, double *qp, double *qs,
                double *fz, double *fr, double *rf);

/* ---- self-test --------------------------------------------------------------------------- */
/* The surface-wave kernels divide with a shared-reciprocal form of the compiler's own IEEE sequence
 * (bh_common.h, Recip/qdiv).  This runs n pseudo-random fp64 quotients a/b with exponents in
 * [-max_exp, max_exp] on the device both ways and counts bitwise differences (expected: 0). */
int bh_selftest_division(long n, unsigned seed, int max_exp, long *mismatches);

/* ---- lock-step chain pool (host side of the sampler) ------------------------------------ */
/* Replaces the per-process loop `while iiter < iter_phase2: SingleChain.iterate()`
 * (src/SingleChain.py:511-589,591-606) for MANY chains advanced together, so that every iteration
 * hands one batch of proposals to bh_swd_batch / bh_rf_batch / bh_likelihood_batch.  Host code, no
 * device work: each chain owns a numpy-compatible MT19937 stream (RandomState(seed) with the legacy
 * uniform / normal / randint algorithms) and replays draw_initvpvs / draw_initmodel /
 * draw_initnoiseparams (:94-157), the five moves (:246-300,:405-428), _validmodel /_validnoise /
 * _validvpvs (:317-403,:413-434), get_acceptance_probability (:463-497), adjust_propdist (:439-461)
 * and append_currentmodel (:508-517) move for move: with the same seed and the same forward values a
 * chain accepts exactly the models the reference chain accepts.
 *
 *   bh_chains_create -> { bh_chains_propose -> [forward + likelihood of `count` models]
 *                         -> bh_chains_accept } until bh_chains_done
 *
 * The first propose/accept pair is the initial model of every chain (_init_model_and_currentvalues,
 * :71-92: accepted unconditionally); each later pair is one iterate() of every chain. */
typedef struct bh_chain_pool bh_chain_pool;

typedef struct bh_chain_config {
    int    ntargets;
    int    layers_min, layers_max;        /* priors['layers'] (half space not counted)          */
    double vs_min, vs_max;                /* priors['vs']                                         */
    double z_min, z_max;                  /* priors['z']                                          */
    int    vpvs_fixed;                    /* priors['vpvs'] is a float: vpvs_min holds it         */
    double vpvs_min, vpvs_max;
    int    has_mantle;                    /* priors['mantle'] = (vs threshold, mantle vp/vs)      */
    double mantle_vs, mantle_vpvs;
    int    has_mohoest;                   /* priors['mohoest'] = (mean, std)                      */
    double moho_mean, moho_std;
    double thickmin;                      /* initparams['thickmin']                               */
    int    has_lvz, has_hvz;              /* initparams['lvz'], ['hvz'] (None -> 0)               */
    double lvz, hvz;
    double propdist[5];                   /* vs, z, birth/death, noise, vpvs                      */
    double acceptance[2];                 /* target acceptance window in percent                  */
    long   iter_burnin, iter_main;
    int    noise_fixed[2 * BH_MAX_TARGETS]; /* per target (corr, sigma): prior is a number        */
    double noise_lo[2 * BH_MAX_TARGETS];  /* the number, or the lower bound                       */
    double noise_hi[2 * BH_MAX_TARGETS];
} bh_chain_config;

/* Caller-owned sample storage, the layout of the reference's shared arrays
 * (src/mcmcOptimizer.py:77-125): float32, NaN-filled by the caller, nmodels rows per chain. */
typedef struct bh_chain_storage {
    long    nmodels;                      /* int(iterations * max(acceptance) / 100)              */
    float  *models;                       /* [nchains][nmodels][2*(layers_max+1)]: vs.., z..      */
    float  *misfits;                      /* [nchains][nmodels][ntargets+1]                       */
    float  *likes;                        /* [nchains][nmodels]                                   */
    float  *noise;                        /* [nchains][nmodels][2*ntargets]                       */
    float  *vpvs;                         /* [nchains][nmodels]                                   */
    double *iter;                         /* [nchains][nmodels]: iteration of acceptance          */
} bh_chain_storage;

int  bh_chains_create(const bh_chain_config *cfg, int nchains, const unsigned *seeds,
                      const bh_chain_storage *storage, bh_chain_pool **pool);
void bh_chains_destroy(bh_chain_pool *pool);
int  bh_chains_set_threads(bh_chain_pool *pool, int nthreads);
/* Draw the next proposal of every chain.  Chains whose proposal passed the prior checks are written
 * compacted, k = 0..count-1: packed[k] = h, vp, vs, rho rows of Lmax doubles each (the layout
 * bh_swd_batch takes with model_stride = 4*Lmax; rho = vp*0.32 + 0.77, src/Targets.py:319),
 * nlay[k], noise[k][2*ntargets], chain[k] = the chain it belongs to.  Arrays hold nchains rows
 * (bh_chains_rows() with a look-ahead, below). */
int  bh_chains_propose(bh_chain_pool *pool, int Lmax, double *packed, int *nlay, double *noise,
                       int *chain, int *count);
/* logL[count], misfits[count][ntargets+1] of the models handed out by the last propose: draw u,
 * accept or reject, store, adapt the proposal widths, advance the iteration counter(s). */
int  bh_chains_accept(bh_chain_pool *pool, const double *logL, const double *misfits);
/* Per model handed out by the last propose (k = 0..count-1): the move that produced it -- 0 vs of a
 * nucleus, 1 depth of a nucleus, 2 birth, 3 death, 4 noise parameter, 5 vp/vs, -1 initial model.
 * A noise move (4) leaves the layered model as it was.  After accept: whether the proposal replaced
 * the chain's current model. */
int  bh_chains_moves(const bh_chain_pool *pool, int *move);
int  bh_chains_accepted(const bh_chain_pool *pool, int *flag);
int  bh_chains_done(const bh_chain_pool *pool);            /* 1 when every chain reached iter_main */
long bh_chains_iteration(const bh_chain_pool *pool);       /* the slowest chain's next iteration; starts at -iter_burnin */
/* Look-ahead.  A Metropolis chain is sequential -- iteration i+1 starts from the model iteration i leaves
 * behind (src/SingleChain.py:554-589) -- so a pool of few chains hands the GPU batches far below the size
 * at which a launch costs more than its latency.  With `nodes` > 1 bh_chains_propose also draws, per chain,
 * the proposals of the FOLLOWING iterations for the most likely outcomes of the ones before (a tree: "if
 * this proposal is rejected, the next one is ...; if accepted, ..."; likelihood of an outcome from the
 * chain's own acceptance rates), up to `nodes` proposals per chain and call, and bh_chains_accept walks the
 * tree along the outcomes that the likelihoods decide: a chain advances by 1 .. nodes iterations per call,
 * chains of a pool no longer stand at the same iteration, and the samples are EXACTLY those of nodes = 1
 * (same random draws in the same order; tests/test_chains.py).  The staging arrays of bh_chains_propose must
 * then hold bh_chains_rows() = nchains * nodes rows.  Allowed whenever no results are outstanding. */
#define BH_CHAIN_MAX_LOOKAHEAD 64
int  bh_chains_set_lookahead(bh_chain_pool *pool, int nodes);
int  bh_chains_lookahead(const bh_chain_pool *pool);
long bh_chains_rows(const bh_chain_pool *pool);
/* totals since creation, initial models not counted: propose/accept pairs, chain iterations they completed
 * (all chains), models they handed out; any pointer may be NULL */
int  bh_chains_advance(const bh_chain_pool *pool, long *calls, long *iterations, long *rows);
int  bh_chains_iterations(const bh_chain_pool *pool, long *iter);   /* iter[nchains]: each chain's next iteration */
/* per-chain bookkeeping; any pointer may be NULL.  naccepted = rows stored so far (the reference's
 * self.n), propdist[nchains][5], accepted/proposed[nchains][5] as in adjust_propdist */
int  bh_chains_counters(const bh_chain_pool *pool, long *naccepted, double *propdist,
                        double *accepted, double *proposed);
/* current state of one chain: model = vs.., z.. (2*nnuclei doubles) */
int  bh_chains_current(const bh_chain_pool *pool, int chain, int *nnuclei, double *model,
                       double *noise, double *vpvs, double *like, double *misfits);
/* The random stream of one chain, for tests and for adopting a numpy state
 * (RandomState.get_state()): key[624], pos, has_gauss, cached_gaussian. */
int  bh_chains_get_rng(const bh_chain_pool *pool, int chain, unsigned *key, int *pos,
                       int *has_gauss, double *gauss);
int  bh_chains_set_rng(bh_chain_pool *pool, int chain, const unsigned *key, int pos,
                       int has_gauss, double gauss);
/* n draws from the stream of one chain: kind 0 = uniform(a, b), 1 = normal(a, b),
 * 2 = randint(a, b) (as doubles) -- numpy.random.RandomState's legacy algorithms */
int  bh_chains_draw(bh_chain_pool *pool, int chain, int kind, double a, double b, int n,
                    double *out);

/* ---- evaluation plan: a batch of proposals -> (logL, misfits) in one call ----------------- */
/* What the sampler does with the proposals of one iteration (reference: one JointTarget.evaluate per
 * chain and iteration, src/SingleChain.py:545-552, src/Targets.py:314-347): copy them to the device,
 * order them, run every dispersion target, every receiver function and the fused likelihood, copy
 * 8*(ntargets+2) bytes per model back.  A plan owns everything that needs -- device buffers, pinned host
 * staging, its own two streams (the receiver functions back-fill the tail of the dispersion kernel),
 * events, sort scratch -- so that bh_eval_submit is a handful of asynchronous HIP calls and nothing else:
 * no allocation, no synchronisation, no interpreter in the loop.
 *
 *   bh_eval_create(...)                         once per (targets, pool size); on the current device
 *   bh_eval_buffers(plan, &packed, &nlay, &noise, &chain, &results)   pinned host memory of the plan
 *   { bh_chains_propose(pool, Lmax, packed, nlay, noise, chain, &count);   writes the staging block
 *     bh_eval_submit(plan, count);                                          asynchronous
 *     bh_eval_wait(plan, NULL);                                             results[] is valid
 *     bh_chains_accept(pool, results, results + count); }
 *
 * Staging layout (max_models rows each): packed [rows][4][Lmax] doubles (h, vp, vs, rho: model_stride =
 * 4*Lmax), noise [rows][2*ntargets], nlay [rows] int32, chain [rows] int32 (not sent to the device).
 * results: logL[count] followed by misfits[count][ntargets+1] of the last submission.
 * row / out_off / off describe one output row exactly as for bh_swd_batch, bh_rf_batch and
 * bh_likelihood_batch; periods, yobs[row] and aux[naux] are HOST arrays, copied once.  A dispersion target
 * with more than 60 observed periods is solved on its 60 periods (columns out_off .. behind the visible
 * ones) and interpolated to obsx[n_dst] at columns dst_off .. with numpy.interp's formula
 * (surf96_modsw.py:35-43,106-122): one bh_eval_interp per such target. */
typedef struct bh_eval_plan bh_eval_plan;
typedef struct bh_eval_interp {
    int target;          /* index into the dispersion targets                                  */
    int dst_off, n_dst;  /* visible columns of the target in the output row, number of obsx    */
    int _pad;
    const double *obsx;  /* observed periods [n_dst] (host)                                    */
} bh_eval_interp;
int  bh_eval_create(int max_models, int Lmax, int row, int nswd, const bh_swd_target *swd,
                    const double *periods, int nperiods, int nrf, const bh_rf_params *rf, int ntargets,
                    const bh_like_target *like, int nflags, const double *yobs, const double *aux,
                    size_t naux, int ninterp, const bh_eval_interp *interp, int use_mfma,
                    bh_eval_plan **plan);
void bh_eval_destroy(bh_eval_plan *plan);
int  bh_eval_buffers(bh_eval_plan *plan, double **packed, int **nlay, double **noise, int **chain,
                     double **results);
int  bh_eval_submit(bh_eval_plan *plan, int count);
int  bh_eval_wait(bh_eval_plan *plan, int *count);   /* blocks until the last submission has landed */
/* How many plans take turns on the device (a pool's chain groups; default 1): passed on as bh_swd_hint's second
 * argument with every submission. */
int  bh_eval_set_concurrency(bh_eval_plan *plan, int plans_in_flight);

/* ---- plumbing for hosts without their own device allocator ------------------------------ */
int bh_malloc(void **dptr, size_t bytes);
int bh_free(void *dptr);
int bh_memcpy_h2d(void *dst, const void *src, size_t bytes, void *stream);
int bh_memcpy_d2h(void *dst, const void *src, size_t bytes, void *stream);
int bh_stream_synchronize(void *stream);
/* Diagnostic: per-frequency constant tables of the receiver-function kernel currently cached (one per
 * (device, nsamp, fsamp, gauss, tshift); bounded, least recently used first out). */
int bh_rf_cached_tables(void);
/* Streams are the caller's.  The batched entry points record library events on the stream they are given (the
 * guards of the dispersion kernels' work-queue slots); a HIP event refers to the stream it was last recorded on,
 * so before DESTROYING a stream that was ever passed to this library call bh_stream_retire(stream): it waits for
 * the stream's work and drops every event the library recorded on it.  (Not needed for the null stream nor for
 * streams that live as long as the process, e.g. the pooled streams of a tensor framework; evaluation plans retire their own
 * streams in bh_eval_destroy.) */
int bh_stream_retire(void *stream);
/* A non-blocking stream on the current device for hosts without a HIP binding of their own, and its end:
 * bh_stream_destroy = bh_stream_retire + hipStreamDestroy. */
int bh_stream_create(void **stream);
int bh_stream_destroy(void *stream);

#ifdef __cplusplus
}
#endif
#endif /* BAYHUNTER_AMD_H */
