// kernels.h -- kernel argument blocks and launchers shared by kernels.hip and capi.hip.
#pragma once
#include <hip/hip_runtime.h>
#include "rf_core.h"
#include "swd_core.h"

namespace bh {

constexpr int SWD_T = 64;   // one wave per workgroup: a lane's search length is data dependent, so
                            // single-wave groups retire (and free their LDS) independently
constexpr int RF_T = 256;
constexpr int BH_NP = 60;   // NP, surfdisp96.f:62
constexpr int BH_NL = 100;  // NL, surfdisp96.f:60
constexpr int BH_NT = 16;   // BH_MAX_TARGETS

struct SwdArgs {
    int B, Lmax, ntargets, out_stride;
    int mstride;             // elements between consecutive models in h/vp/vs/rho
    int vec2;                // rows are 16-byte aligned and Lmax is even: fetch two layers per load
    int stage;               // swd_kernel: > 0: a search's results wait in LDS (that many periods per lane)
                             // and are written as one row
    const int *nlay;
    const int *order;        // optional: the i-th search taken from the queue is model order[i]
    const double *h, *vp, *vs, *rho;
    const double *periods;
    double *out;
    int *err;
    double *ws;
    unsigned int *counters;  // [BH_NT] work-queue heads (one per target), zeroed by the caller of launch_swd*
    SwdTargetDev tg[BH_NT];
    // A launch covers the targets whose bit is set in tmask (nsel of them): the grid always spans all targets
    // in y and the workgroups of the others leave at once.  bh_swd_batch may give the targets of one call to
    // different kernel forms (capi.hip: plan_forms), one launch per form on concurrent streams.  (A list of
    // target indices looked up by blockIdx.y cost swd_kernel six VGPRs -- 195, allocated as 200 -- and with
    // them the co-residency with rf_kernel: 2 x 192 + 128 is all a SIMD has.)
    unsigned int tmask;
    int nsel;
    unsigned char tord[BH_NT];   // the launch's targets, heaviest first (narrow teams: the order in which every wave drains them)
};

struct RfArgs {
    int B;
    int mstride;             // elements between consecutive models in h/vp/vs/rho
    const int *nlay;
    const double *h, *vp, *vs, *rho, *qp, *qs;
    const double *tw;  // FFT twiddles, rf_host.h
    const double *ftab; // per-frequency constants [nfreq][3], rf_host.h (rf_fill_freq_table)
    double *out;
    double *out_fz, *out_fr;  // optional [B][nsamp] vertical / radial traces (synrf_cwrap's fz, fr)
    RfLaunch P;
};

struct ModelPriorsDev {
    int layers_min, layers_max;
    double vs_min, vs_max, z_min, z_max, thickmin, lowvelperc, highvelperc, mantle_vs, mantle_vpvs;
};
struct VoronoiArgs {
    int B, Lmax;
    const int *nlay;
    const double *vs, *z, *vpvs;
    double *model;
    int *valid;
    ModelPriorsDev pri;
};
hipError_t launch_voronoi(const VoronoiArgs &A, hipStream_t stream);
hipError_t launch_order_keys(int B, int Lmax, int mstride, const int *nlay, const double *h, const double *vs,
                             double reach, int by_length, int *keys, hipStream_t stream);
hipError_t launch_division_selftest(long n, unsigned seed, int max_exp, unsigned long long *bad, hipStream_t stream);

struct LikeTargetDev {
    int n, off, cov, aux_off;
    double logdet_extra;
};
constexpr int LIKE_T = 256;   // threads per workgroup
constexpr int LIKE_M = 8;     // models per workgroup (register tile of the dense R^-1 product)
constexpr int LIKE_NMAX = 1024;  // max data points per target (LDS: LIKE_M*n doubles)

struct LikeArgs {
    int B, ntargets, out_stride, nflags;
    const double *out;
    const int *err;
    const double *yobs, *noise, *aux;
    double *logL, *misfits;
    double *gq;              // [ntargets][B][gq_groups][2] partial (q, sum d^2) of dense-Gaussian targets, or null
    int gq_groups;           // column-tile groups per model in gq (the widest Gaussian target's)
    LikeTargetDev tg[BH_NT];
};
// The dense Gaussian product runs per group of GQ_NTG column tiles of 16 (like_kernel.hip: gauss_q_kernel)
constexpr int GQ_NTG = 4;
__host__ __device__ inline int gq_groups_of(int n) { return ((n + 15) / 16 + GQ_NTG - 1) / GQ_NTG; }

hipError_t launch_like(const LikeArgs &A, int nmax, hipStream_t stream, int stages = 3);
hipError_t launch_swd(const SwdArgs &A, int resident_waves, hipStream_t stream);
hipError_t launch_swd_team(const SwdArgs &A, int team_lanes, int resident_waves, hipStream_t stream);
size_t swd_team_lds_bytes(int Lmax, int team_lanes);
hipError_t launch_rf(const RfArgs &A, hipStream_t stream);
size_t rf_lds_bytes(int Lmax, int nsamp, int M, bool zr = false);

}  // namespace bh
