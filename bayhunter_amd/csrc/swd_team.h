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
    if (cap > nlanes) cap = nlanes;          // one chain lane per trial
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
