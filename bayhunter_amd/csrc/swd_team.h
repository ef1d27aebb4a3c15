// swd_team.h -- latency-oriented surface-wave search: one WAVE per (model, dispersion target).
//
// swd_lane (swd_core.h) gives every lane its own model: maximal throughput, but one search is a
// chain of ~700 period-equation evaluations x (L-1) layers, ~14 ms however many lanes idle around
// it.  For batches that cannot fill the chip (a few thousand chains and fewer) this file spends
// lanes on ONE search instead, without changing a single value of it:
//
//   * speculative bracketing -- the scan of getsol (surfdisp96.f:448-470) visits c1+dc, c1+2dc, ...
//     until the period equation changes sign; those trial velocities are known in advance, so a
//     round evaluates the next `nt` of them at once and then feeds the values to the unchanged
//     control code (swd_control) in order, discarding whatever lies beyond the first sign change;
//   * layer-parallel assembly -- inside one evaluation the Dunkin (or Love) layer matrices do not
//     depend on the propagated vector (surfdisp96.f:813-837 vs :838-848): lane (j, r) assembles the
//     matrix of layer r for trial j into LDS, then lane j runs the short sequential product chain.
//
// With L-1 = 9 layers a round carries 7 trials on 63 lanes; a root costs ~1 + 3 + 7 rounds of
// ~(1 layer assembly + 9 chain steps) instead of ~25 full evaluations of 9 layers each: ~10x less
// latency for ~5x the lane-cycles.  Every lane of the wave executes driver/control redundantly on
// identical (wave-uniform) state, so there is no broadcast step and the code is the one swd_lane uses.
#pragma once
#include "swd_core.h"

namespace bh {

enum { SWD_TEAM_NT = 16 };   // max speculative trials per round (LDS: trials[], dels[])

// Plan the trials of this round into `trials` (LDS; every lane writes the same values).
BH_DEV int swd_team_plan(const SwdState &S, int nlanes, double *trials)
{
    const double dc = (double)0.005f;
    const int nlm = S.mmax - S.llw;
    int cap = (nlm > 0) ? nlanes / nlm : SWD_TEAM_NT;
    if (cap < 1) cap = 1;
    if (cap > SWD_TEAM_NT) cap = SWD_TEAM_NT;
    if (cap > nlanes) cap = nlanes;          // one chain lane (or 8-lane group) per trial
    int nt = 1;
    trials[0] = S.ceval;
    if (S.st == SWD_ST_A || S.st == SWD_ST_B) {
        // the scan continues like this unless a sign change or a bound stops it (swd_control).
        // At the entry evaluation (ST_A) the direction is not known yet: assume upwards, which is
        // what getsol chooses unless the dispersion is reversed (then the guess is discarded).
        double c1s = S.c2;
        int idirs = S.idir;
        if (S.st == SWD_ST_A && cap > 1) {
            c1s = S.ceval;
            idirs = +1;
            double c2s = swd_bracket_next(c1s, idirs, S.clow, dc);
            trials[nt++] = c2s;
            c1s = c2s;
        }
        while (nt < cap) {
            if (c1s < S.cc || c1s >= S.cfail) break;
            double c2s = swd_bracket_next(c1s, idirs, S.clow, dc);
            trials[nt++] = c2s;
            c1s = c2s;
        }
    }
    return nt;
}

// Lane `lane` of `nlanes` assembles its share of the (trial, layer) matrices into `mats`.
template <class Lay>
BH_DEV void swd_team_assemble(const Lay &lay, int lane, int nlanes, int ifunc, const SwdState &S,
                              int nt, const double *trials, double *mats)
{
    const int nlm = S.mmax - S.llw;
    for (int idx = lane; idx < nt * nlm; idx += nlanes) {
        const int j = idx / nlm, r = idx - j * nlm;
        const int i0 = S.llw - 1 + r;        // 0-based layer; Fortran m = llw + r
        const double wvno = S.omega / trials[j];
        if (ifunc == 1) {
            LoveLayer o;
            swd_love_layer(lay, i0, wvno, S.omega, o);
            double *p = mats + (long)idx * SWD_NCA;
            p[0] = o.cosq; p[1] = o.y; p[2] = o.z; p[3] = o.xmu;
        } else {
            double omega = S.omega;
            if (omega < 1.0e-4) omega = 1.0e-4;
            Dunkin a;
            swd_ray_layer_matrix(lay, i0, wvno, wvno * wvno, omega, a);
            double *p = mats + (long)idx * SWD_NCA;
            p[0] = a.c11; p[1] = a.c12; p[2] = a.c13; p[3] = a.c14; p[4] = a.c15; p[5] = a.c21;
            p[6] = a.c22; p[7] = a.c23; p[8] = a.c24; p[9] = a.c31; p[10] = a.c32; p[11] = a.c33;
            p[12] = a.c34; p[13] = a.c35; p[14] = a.c41; p[15] = a.c42; p[16] = a.c43; p[17] = a.c51;
            p[18] = a.c53;
        }
    }
}

// Lane j < nt propagates the half-space vector through the assembled layers of trial j.
template <class Lay>
BH_DEV void swd_team_chain(const Lay &lay, int lane, int ifunc, const SwdState &S, int nt,
                           const double *trials, const double *mats, double *dels)
{
    if (lane >= nt) return;
    const int nlm = S.mmax - S.llw;
    const double wvno = S.omega / trials[lane];
    if (ifunc == 1) {
        double e1, e2;
        swd_love_halfspace(lay, S.mmax, wvno, S.omega, e1, e2);
        for (int r = nlm - 1; r >= 0; r--) {
            const double *p = mats + ((long)lane * nlm + r) * SWD_NCA;
            LoveLayer o;
            o.cosq = p[0]; o.y = p[1]; o.z = p[2]; o.xmu = p[3];
            swd_love_apply(e1, e2, o);
        }
        dels[lane] = e1;
    } else {
        double omega = S.omega;
        if (omega < 1.0e-4) omega = 1.0e-4;
        double e[5];
        swd_ray_halfspace(lay, S.mmax, wvno, wvno * wvno, omega, e);
        for (int r = nlm - 1; r >= 0; r--) {
            const double *p = mats + ((long)lane * nlm + r) * SWD_NCA;
            Dunkin a;
            a.c11 = p[0]; a.c12 = p[1]; a.c13 = p[2]; a.c14 = p[3]; a.c15 = p[4]; a.c21 = p[5];
            a.c22 = p[6]; a.c23 = p[7]; a.c24 = p[8]; a.c31 = p[9]; a.c32 = p[10]; a.c33 = p[11];
            a.c34 = p[12]; a.c35 = p[13]; a.c41 = p[14]; a.c42 = p[15]; a.c43 = p[16]; a.c51 = p[17];
            a.c53 = p[18];
            swd_dunkin_apply(e, a);
        }
        dels[lane] = (S.llw != 1) ? swd_ray_water(lay, wvno, omega, e) : e[0];
    }
}

#if !defined(BH_HOSTSIM)
// Rayleigh chain with the 5 components of the Dunkin vector on 5 lanes (device only; the host
// replay keeps swd_team_chain, whose arithmetic this reproduces operation for operation):
// lanes 8j .. 8j+4 serve trial j.  Lane i forms ee_i = sum_k e_k*ca(k,i) in the reference's order
// from its column of the layer matrix, the group max-reduces |ee_i| (normc), every lane divides its
// own component with the shared reciprocal and the new vector is re-gathered by shuffles.
// `lane` is the lane within the team (8 | team width, teams aligned in the wave); requires
// 8*nt <= team width and every lane of the team active.
template <class Lay>
__device__ __forceinline__ void swd_team_chain_ray5(const Lay &lay, int lane, const SwdState &S,
                                                    int nt, const double *trials, const double *mats,
                                                    double *dels)
{
    const int j = lane >> 3, i = lane & 7, gbase = (int)(threadIdx.x & 63) & ~7;
    const bool live = (j < nt) && (i < 5);
    const int nlm = S.mmax - S.llw;
    const double wvno = S.omega / trials[j < nt ? j : 0];
    double omega = S.omega;
    if (omega < 1.0e-4) omega = 1.0e-4;
    double e[5];
    swd_ray_halfspace(lay, S.mmax, wvno, wvno * wvno, omega, e);
    // column i of ca as indices into the 19 stored values (swd_team_assemble's order):
    // 0 c11 1 c12 2 c13 3 c14 4 c15 5 c21 6 c22 7 c23 8 c24 9 c31 10 c32 11 c33 12 c34 13 c35
    // 14 c41 15 c42 16 c43 17 c51 18 c53 ; c25=c14 c44=c22 c45=c12 c52=c41 c54=c21 c55=c11
    int k1, k2, k3, k4, k5;
    switch (i) {
    case 0: k1 = 0; k2 = 5; k3 = 9; k4 = 14; k5 = 17; break;
    case 1: k1 = 1; k2 = 6; k3 = 10; k4 = 15; k5 = 14; break;
    case 2: k1 = 2; k2 = 7; k3 = 11; k4 = 16; k5 = 18; break;
    case 3: k1 = 3; k2 = 8; k3 = 12; k4 = 6; k5 = 5; break;
    default: k1 = 4; k2 = 3; k3 = 13; k4 = 1; k5 = 0; break;
    }
    for (int r = nlm - 1; r >= 0; r--) {
        const double *p = mats + ((long)(j < nt ? j : 0) * nlm + r) * SWD_NCA;
        double ee = ((((0.0 + e[0] * p[k1]) + e[1] * p[k2]) + e[2] * p[k3]) + e[3] * p[k4]) + e[4] * p[k5];
        double t1 = live ? fabs(ee) : 0.0;
        t1 = dmax(t1, __shfl_xor(t1, 1, 64));
        t1 = dmax(t1, __shfl_xor(t1, 2, 64));
        t1 = dmax(t1, __shfl_xor(t1, 4, 64));
        if (t1 < 1.e-40) t1 = 1.0;
        const double en = qdiv(ee, recip_of(t1));
        e[0] = __shfl(en, gbase + 0, 64);
        e[1] = __shfl(en, gbase + 1, 64);
        e[2] = __shfl(en, gbase + 2, 64);
        e[3] = __shfl(en, gbase + 3, 64);
        e[4] = __shfl(en, gbase + 4, 64);
    }
    if (j < nt && i == 0) dels[j] = (S.llw != 1) ? swd_ray_water(lay, wvno, omega, e) : e[0];
}
#endif

// Feed the round's values to the search in order; stop at the first one that ends the scan.
// Returns the number of values the reference would have computed (the rest was speculation).
BH_DEV int swd_team_consume(SwdState &S, int nt, const double *trials, const double *dels)
{
    int used = 0;
    for (int j = 0; j < nt; j++) {
        swd_control(S, dels[j]);
        used++;
        if (S.ev != SWD_EV_NONE || S.st != SWD_ST_B) break;
        if (j + 1 < nt && S.ceval != trials[j + 1]) break;   // cannot happen; guards the replay
    }
    return used;
}

// LDS doubles needed by one team for models of up to Lmax layers
BH_HD int swd_team_lds_doubles(int Lmax, int nlanes)
{
    int nm = Lmax > nlanes ? Lmax : nlanes;
    return nm * SWD_NCA + 2 * SWD_TEAM_NT;
}

}  // namespace bh
