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
template <class Nev>
BH_DEV int swd_team_consume(SwdState &S, Nev &nv, int nt, const double *trials, const double *dels)
{
    int used = 0;
    for (int j = 0; j < nt; j++) {
        swd_control(S, dels[j], nv);
        used++;
        if (S.ev != SWD_EV_NONE || S.st != SWD_ST_B) break;
        if (j + 1 < nt && S.ceval != trials[j + 1]) break;   // cannot happen; guards the replay
    }
    return used;
}

// ================================================================================================
// Wide teams (64*W lanes per search, swd_teamw_kernel): the lowest-latency form.
//
// Same two ideas as above, plus:
//   * the bisection tree.  A root is refined by ~13 sequential evaluations (surfdisp96.f:582-673), nine
//     in ten of them bisections, and a bisection point is known before the value that selects it: a
//     refinement round carries the whole tree of candidates below the pending trial, as deep as the
//     slots allow, and consumes one value per level (swd_teamw_round, swd_teamw_node below).
//   * speculation across the end of a root search.  Once the tree reaches the halving at which the
//     stopping test fires, its leaves are the possible roots; behind each comes the START of the search
//     that follows: the entry evaluation of the next period (or of the second solve of a
//     group-velocity pair) at c - 1.5 dc and the first grid points of its bracketing scan
//     (surfdisp96.f:268-271,282-294,448-470).  In the round in which the root converges the next search
//     is therefore already bracketed (or well on its way).
//   * a value is only ever consumed for the (omega, c) it was computed at: the consuming loop
//     matches every speculative trial against what the unchanged control code (swd_control /
//     swd_driver) asks for next, bit for bit, and stops at the first mismatch.  The prediction can be
//     wrong (reversed scan direction, no convergence yet, no root) -- it can never change a result.
//   * the Dunkin chain of a trial runs on a QUAD of lanes (component i on lane i, the fifth on all
//     four) with DPP quad permutes instead of LDS shuffles; a lane picks its column out of the 19 stored
//     values of a layer matrix (kernels.hip: quad_load).
// Trials of a round are numbered 0..nt-1; trial j has its own (c, omega).
enum { SWD_TEAMW_NT = 64 };       // max trials per round (one per lane of the control wave)
#ifndef SWD_TEAMW_MIDROOM
#define SWD_TEAMW_MIDROOM 24   // scan slots from which on the cell midpoints ride along (replay: 16 and 24 within 1 %)
#endif

// A stored layer matrix is the 19 distinct values of Dunkin's 5x5 (+1 for Love's four: cosq, y, z, xmu), row by
// row: 0 c11 1 c12 2 c13 3 c14 4 c15 | 5 c21 6 c22 7 c23 8 c24 | 9 c31 10 c32 11 c33 12 c34 13 c35 |
// 14 c41 15 c42 16 c43 | 17 c51 18 c53   (c25 = c14, c44 = c22, c45 = c12, c52 = c41, c54 = c21, c55 = c11).
// Until round 3 it was the 25 entries column by column on 30 doubles (one aligned run per lane of the quad
// chain): 18.3 KB of LDS per 64-lane team -- eight teams per CU, two waves per SIMD where the registers allow
// three.  With 20 doubles (at a pitch of 21, below) a team takes 13.4 KB: twelve per CU.
enum { SWD_MAT = 20 };
// Slot s of the store begins at double swd_mat_off(s): a pitch of 21 doubles, one more than a matrix.  The lanes of
// a wave write one matrix each, 8 bytes per store, and the quads of the chain read the matrices of different trials
// side by side: with a pitch of 20 doubles (40 banks) lanes l and l + 8, and the trials of a five-layer model, fall
// on the same banks (8-way conflicts); 42 banks spread them over all 64 (one 512-lane team 0.602 -> 0.581 ms, four
// waves 0.693 -> 0.674, 1 024 five-layer searches 0.521 -> 0.516; same box, profiles/r03_ab_pitch.txt).  A 64-lane
// team then takes 13.4 KB: twelve per CU up to 28 layers, eleven beyond.  (An irregular padding -- one double per
// eight slots -- took the kernels from 142 to 189 VGPRs; sizing the per-trial arrays by the launch, which would keep
// twelve teams at any depth, to 168 + scratch.)
enum { SWD_PITCH = 21 };
BH_HD int swd_mat_off(int s) { return s * SWD_PITCH; }

// Slots per round of a team of W waves for models with nlm layers above the half-space: cell (j, r) -- layer r of
// trial j -- is assembled by lane j * nlm + r of the whole team, quad q of wave w chains trial 16 w + q.
// (Round 4 tried dealing the trials wave by wave -- wave w assembles AND chains trials w * tpw ..., which removes
// the workgroup barrier between the two phases: cfg4 0.600 -> 0.681 ms, cfg5 8.09 -> 9.28 ms.  With 64 / nlm trials
// per wave every wave walks through the chain at a quarter of its quads, eight waves on four SIMDs, where the
// dealing above fills three waves and lets five skip it.)
BH_HD int swd_teamw_cap(int nlm, int W, int iwave)
{
    const int NL = 64 * W;
    int cap = nlm > 0 ? NL / nlm : (int)SWD_TEAMW_NT;
    if (iwave == 2 && cap > 16 * W) cap = 16 * W;             // one quad per Rayleigh trial
    if (cap > SWD_TEAMW_NT) cap = SWD_TEAMW_NT;
    return cap < 1 ? 1 : cap;
}

// floor(log2(x)) for x >= 1
BH_DEV int swd_ilog2(int x) { return 31 - __builtin_clz((unsigned)x); }

// Cell i of a bracketing scan that starts at `base`: its base b_i and its end c_i = b_i + dc, where
// b_0 = base and b_{i+1} = c_i -- the reference adds dc once per step (surfdisp96.f:448-452), and so
// must every replay, rounding included.  But those additions do not round while the values stay in one
// binade: dc = dble(0.005) carries 24 significant bits down to 2^-31 and a velocity in [2^e, 2^(e+1))
// has an ulp of 2^(e-52) <= 2^-31, so x + dc is exactly representable unless it reaches 2^(e+1); then
// b_i = base + i dc and c_i = base + (i+1) dc hold exactly (the products are exact: i + 1 <= 64).  A
// scan that crosses a power of two (2 or 4 km/s) takes the additions one by one from the start.
BH_DEV void swd_scan_cell(double base, int i, double *b, double *cn)
{
    const double dc = (double)0.005f;
    const double hi = base + (double)(i + 1) * dc;
#if defined(BH_HOSTSIM)
    long long ub, uh;
    __builtin_memcpy(&ub, &base, 8); __builtin_memcpy(&uh, &hi, 8);
    const bool same_binade = ub > 0 && ((ub ^ uh) >> 52) == 0;
#else
    const int eb = __double2hiint(base), eh = __double2hiint(hi);
    const bool same_binade = eb > 0 && ((eb ^ eh) >> 20) == 0;       // sign and exponent bits agree
#endif
    if (same_binade) {
        *b = base + (double)i * dc;
        *cn = hi;
        return;
    }
    double x = base, c = base + dc;
    for (int n = i; n > 0; n--) { x = c; c = x + dc; }
    *b = x;
    *cn = c;
}

// ---- the plan of a round ----------------------------------------------------------------------
// Slot 0 is the evaluation the search is waiting for; the rest is speculation, most valuable first:
//   bracketing (ST_A / ST_B)   the scan continues on the grid: scan trial i = base + (i+1) dc by
//                              repeated addition (surfdisp96.f:448-470), assuming the upward
//                              direction getsol takes for normal dispersion
//   refinement (ST_TOP / MID)  (1) the bisection tree below the pending trial.  Nine refinement steps
//                              in ten are bisections (`half`, :620-640: forced while the end values of
//                              the bracket differ by > 100x, and 0.005 -> 1e-6 c takes 11 of them), and
//                              a bisection point is known before the value that selects it: the
//                              midpoint of one of the two sub-brackets the pending trial can leave.  A
//                              tree of depth D -- 2, 4, 8, ... candidates per level, 2^(D+1)-2 slots --
//                              lets a round consume D+1 values.  D is bounded by the slots and by the
//                              number of halvings left before the stopping test |c1-c2| <= 1e-6 c1
//                              (:614) fires, which the bracket width predicts.
//                              (2) once the tree reaches that last halving its leaves are the possible
//                              roots: behind each (at most two) comes the search that follows -- its
//                              entry evaluation at c - 1.5 dc with the next period's omega (or the
//                              second solve of a group-velocity pair, :268-271,:282-294) and its scan
//   with >= 16 slots to spare  the midpoint of every scan cell: nevill's first point (:583) if the
//                              sign change lies in that cell
// The plan is a pure function of the (wave-uniform) state: swd_teamw_round lays the slots out,
// swd_teamw_trial gives slot j's (c, omega) -- on the device lane j computes its own.  A slot whose
// scan has left the search bounds is NaN (never matched).  Every prediction here may be wrong (a
// Neville step instead of a bisection, a reversed scan, no root): values are consumed by matching
// (omega, c), so a wrong prediction costs idle lanes and nothing else.
struct TeamwScan {                // an entry evaluation (optional) and the scan that steps away from it
    int entry;                    // slot of the entry evaluation (c = base), or -1
    int scan0, stride, nscan;     // scan trial i in slot scan0 + i*stride, i < nscan (stride 2: cell
                                  // midpoints in between)
    double base, oms;             // the point the scan steps away from, omega of the scan
};
struct TeamwRound {
    int nt;                       // slots in use
    int nhalf;                    // bisection candidates in slots 1 .. nhalf
    int chains;                   // 0: slots 1 .. nhalf (0, 2, 6, 14, 30 or 62) are the complete tree in level
                                  //    order, left (c1 side) to right
                                  // 1: the two INNER CHAINS below a pending Neville estimate x (see swd_teamw_round):
                                  //    slot 2i-1 = A_i, slot 2i = B_i, i = 1 .. nhalf/2
    int ngrp;                     // scan groups in use (0..2); g1 follows g0
    TeamwScan g0, g1;
    int pm0, pmi, npm;            // predicted-cell midpoints of g0's scan: slots pm0 .. pm0+npm-1 hold the midpoints
                                  // of scan cells pmi .. pmi+npm-1 (swd_teamw_predict)
};
#ifndef SWD_TEAMW_CHAIN_MAX
#define SWD_TEAMW_CHAIN_MAX 14      // deepest inner chain (0.005 -> 1e-6 c takes 11-13 halvings)
#endif

// What the plan keeps across the rounds of a search: omega of the search that follows the current one (k, pass),
// and the roots of the last periods (first solves), from which the next root is extrapolated (swd_teamw_predict).
struct TeamwNext {
    int k, pass;
    double oms;
    int hk, nr;                   // period the history belongs to; roots held (0..3)
    double r1, r2, r3;            // c(hk-1), c(hk-2), c(hk-3)
};
#ifndef SWD_TEAMW_NPM
#define SWD_TEAMW_NPM 3           // predicted-cell midpoints per scan round (0: none)
#endif

// The number of halvings that bring a bracket of width w down to tol: the smallest i >= 0 with w / 2^i <= tol,
// at most SWD_TEAMW_CHAIN_MAX (w, tol > 0 and finite; anything else: 0).  Powers of two scale exactly, so this is
// a comparison of exponents and one of mantissas.
BH_DEV int swd_halvings(double w, double tol)
{
    if (!(w > tol) || !(tol > 0.0)) return 0;
#if defined(BH_HOSTSIM)
    int ew, et;
    const double mw = std::frexp(w, &ew), mt = std::frexp(tol, &et);
#else
    const int ew = __builtin_amdgcn_frexp_exp(w), et = __builtin_amdgcn_frexp_exp(tol);
    const double mw = __builtin_amdgcn_frexp_mant(w), mt = __builtin_amdgcn_frexp_mant(tol);
#endif
    int i = ew - et + (mw > mt ? 1 : 0);            // w = mw 2^ew, tol = mt 2^et, mantissas in [0.5, 1)
    return i < 0 ? 0 : i > SWD_TEAMW_CHAIN_MAX ? SWD_TEAMW_CHAIN_MAX : i;
}

// ---- where will this scan end?  -----------------------------------------------------------------
// The first bisection after a bracket (`nevill`'s opening `half`, surfdisp96.f:583) is an evaluation at the
// midpoint of the scan cell that holds the sign change, and it is alone in its round: its value decides the
// Neville estimate that follows (replay: one round in four consumes this one value).  The midpoint is known as
// soon as the CELL is, and dispersion curves are smooth: the root of period k extrapolated from the last three
// roots (quadratically in the period index; linearly from two) lands within one and a half cells of the true one
// three times in four (tools/replay_rounds.py).  A scan round therefore gives its last SWD_TEAMW_NPM slots to the
// midpoints of the predicted cell and its neighbours; when one of them is the bracketing cell the search is past
// the first bisection in the round in which it finds the bracket.  As everywhere in the plan a wrong prediction
// costs lanes, never a value: slots are consumed by matching (omega, c).
// History: called by the plan of every round; period S.k of a mode has seen the roots of S.k - 1, ... .
BH_DEV void swd_teamw_history(const SwdState &S, TeamwNext &nx)
{
    if (nx.hk == S.k) return;
    if (S.k == nx.hk + 1 && S.k > 1) {
        nx.r3 = nx.r2; nx.r2 = nx.r1; nx.r1 = S.cprev;
        nx.nr = nx.nr < 3 ? nx.nr + 1 : 3;
    } else {
        nx.nr = 0;                                  // first period of a mode (or periods skipped: no history)
        if (S.k > 1) { nx.r1 = S.cprev; nx.nr = 1; }
    }
    nx.hk = S.k;
}
// Predicted root of the search the state is in (NaN: no prediction).
BH_DEV double swd_teamw_predict(const SwdState &S, const SwdTargetDev &tg, const double *BH_RESTRICT per,
                                const TeamwNext &nx)
{
    if (tg.mode != 1 || S.iq != 1) return __builtin_nan("");
    if (S.pass == 1) {
        // second solve of a group-velocity pair, at T/(1-h) instead of T/(1+h): the root moves by the curve's
        // slope -- taken from the step since the previous period -- times the difference of the two periods
        if (nx.nr < 1 || S.k < 2) return __builtin_nan("");
        const double dt = (double)S.t1b - (double)S.t1a, step = per[S.k - 1] - per[S.k - 2];
        return S.ck + (S.ck - nx.r1) * (dt / step);
    }
    if (nx.nr >= 3) return 3.0 * (nx.r1 - nx.r2) + nx.r3;
    if (nx.nr == 2) return 2.0 * nx.r1 - nx.r2;
    return __builtin_nan("");
}
// Gives the last slots of a scan group (stride 1, `nscan` cells from slot `scan0`, ending at slot `end`) to the
// midpoints of the predicted cell and its neighbours; returns the group's new cell count.
BH_DEV int swd_teamw_reserve_mids(TeamwRound &R, const TeamwScan &g, int end, double cpred)
{
    R.npm = 0; R.pm0 = end; R.pmi = 0;
    const int npm = SWD_TEAMW_NPM;
    if (npm <= 0 || g.stride != 1 || g.nscan <= npm + 1 || !(cpred == cpred)) return g.nscan;
    const double pf = (cpred - g.base) * 200.0;     // cells of dc = 0.005 (an estimate: any rounding will do)
    if (!(pf > -1.0 && pf < 80.0)) return g.nscan;
    int lo = (int)pf - npm / 2;                     // first cell with a midpoint
    if (lo < 0) lo = 0;
    const int kept = g.nscan - npm;                 // cells the scan keeps
    if (lo >= kept) return g.nscan;                 // the predicted cells lie beyond this round: scan on
    R.npm = npm; R.pm0 = end - npm; R.pmi = lo;
    return kept;
}

// lays one scan group out in slots [first, first + room): entry (if has_entry) + scan
BH_DEV int swd_teamw_scan_layout(TeamwScan &g, int first, int room, bool has_entry, bool scan_ok)
{
    g.entry = -1; g.scan0 = first; g.stride = 1; g.nscan = 0;
    if (room <= 0) return first;
    if (has_entry) { g.entry = first; g.scan0 = first + 1; room--; }
    if (!scan_ok || room <= 0) return g.scan0;
    g.stride = room >= SWD_TEAMW_MIDROOM ? 2 : 1;
    g.nscan = (room + g.stride - 1) / g.stride;     // stride 2 and odd room: the last cell has no midpoint
    return g.scan0 + room;
}

// The search that follows the current one, if it is predictable (mirror of swd_driver; only the plain case --
// fundamental mode, no workspace): its omega into nx (kept across the rounds of a search), the lower bound of its
// scan, and whether it starts from the root about to be found (per_leaf) or from c(k) of the first solve.
BH_DEV bool swd_teamw_follow(const SwdState &S, const SwdTargetDev &tg, const double *BH_RESTRICT per,
                             TeamwNext &nx, bool *per_leaf, double *clows)
{
    const double TWOPI = 2.0 * 3.141592653589793, one = 1.0e-2, dc = (double)0.005f;
    const float h = 0.005f;
    *per_leaf = true;
    *clows = 0.0;
    if (tg.mode != 1 || S.iq != 1 || S.ceval > (double)S.betmx) return false;
    const bool fresh = nx.k != S.k || nx.pass != S.pass;
    if (tg.igr > 0 && S.pass == 0) {                // second solve of the pair, surfdisp96.f:282-294
        if (fresh) nx.oms = TWOPI / (double)S.t1b;
        *clows = 0.0 + one * dc;
    } else {                                        // next period, surfdisp96.f:231-239,268-271
        const int k2 = S.k + 1;
        if (k2 > tg.nper || k2 >= S.ift) return false;
        if (fresh) {
            double t1 = per[k2 - 1];
            if (tg.igr > 0) t1 = (double)(float)(t1 / (double)(1.f + h));
            nx.oms = TWOPI / t1;
        }
        *per_leaf = S.pass == 0;                    // (pass 1: it starts from c(k), the first root)
        *clows = S.cc;
    }
    nx.k = S.k; nx.pass = S.pass;
    return true;
}

BH_DEV TeamwRound swd_teamw_round(const SwdState &S, const SwdTargetDev &tg, const double *BH_RESTRICT per,
                                  int cap, TeamwNext &nx)
{
    const double dc = (double)0.005f;
    TeamwRound R;
    R.nt = 1; R.nhalf = 0; R.ngrp = 0; R.chains = 0;
    R.g0.entry = -1; R.g0.scan0 = 1; R.g0.stride = 1; R.g0.nscan = 0; R.g0.base = S.ceval; R.g0.oms = S.omega;
    R.g1 = R.g0;
    R.pm0 = 0; R.pmi = 0; R.npm = 0;
    swd_teamw_history(S, nx);
    if (cap > SWD_TEAMW_NT) cap = SWD_TEAMW_NT;
    if (cap <= 1) return R;
    if (S.st == SWD_ST_A) {
        // the scan is only laid out in its plain form: upwards, never turned around at clow
        // (swd_bracket_next resets c1 to clow when c1 + dc <= clow)
        R.ngrp = 1;
        R.nt = swd_teamw_scan_layout(R.g0, 1, cap - 1, false, R.g0.base + dc > S.clow);
        R.g0.nscan = swd_teamw_reserve_mids(R, R.g0, R.nt, swd_teamw_predict(S, tg, per, nx));
        return R;
    }
    if (S.st == SWD_ST_B) {
        // slot 0 is itself a scan point: the step from S.c1
        R.g0.base = S.c1;
        if (S.idir > 0 && S.c1 + dc == S.ceval && R.g0.base + dc > S.clow) {
            R.ngrp = 1;
            R.nt = swd_teamw_scan_layout(R.g0, 0, cap, false, true);
            R.g0.nscan = swd_teamw_reserve_mids(R, R.g0, R.nt, swd_teamw_predict(S, tg, per, nx));
        }
        return R;
    }
    // The first bisection after a bracket (`nevill`'s opening `half`, surfdisp96.f:583): its value decides the
    // Neville estimate that follows, nine times in ten nothing below it is ever used (replay: 329 of 359 such
    // rounds consume the one value) -- a round of one trial, not worth a tree.
#if !defined(BH_TEAMW_NO_LONE)
    if (S.nev == 1 && S.nctrl == 1 && S.st == SWD_ST_TOP) return R;
#endif
    const double onea = 1.5;
    R.chains = 0;
    const double wa = fabs(S.ceval - S.c1), wb = fabs(S.c2 - S.ceval), tol = 1.e-6 * fabs(S.ceval);
    bool per_leaf;                                  // the start of the search that follows depends on this root
    double clows;

    // ---- inner chains.  The pending trial x is a Neville estimate (nev == 2): it sits within ~1e-5 of the root,
    // so whichever side of it the root is on, the bisections that follow all move the FAR end of the bracket
    // towards x -- forced while the end values differ by > 100x (surfdisp96.f:620-640) -- until the next Neville
    // step or the stopping test.  Of the complete tree below x only two paths are ever taken (replay: 1 968
    // refinement rounds, 5 exceptions, all on low-velocity-zone models): L R R R ... and R L L L ....  Those two
    // chains cost 2 slots per level instead of 2^level: with 16 slots a round follows 7 halvings instead of 3.
    //   A_i = mid(A_{i-1}, x), A_0 = c1  (after x: c2 = x, then c1 = A_1, c1 = A_2, ...)
    //   B_i = mid(x, B_{i-1}), B_0 = c2  (after x: c1 = x, then c2 = B_1, c2 = B_2, ...)
    // A chain is no longer than the halvings left on its side before |c1 - c2| <= 1e-6 c1 fires (:614).
#if !defined(BH_TEAMW_NO_CHAINS)                  // (A/B switch: the complete tree everywhere)
    if (S.nev == 2 && cap >= 5) {
        // nodes of chain A / B evaluated until the stopping test: the smallest i with w / 2^i <= tol
        const int ia = swd_halvings(wa, tol), ib = swd_halvings(wb, tol);
        int D = ia > ib ? ia : ib;
        if (D > (cap - 1) / 2) D = (cap - 1) / 2;
        if (D >= 2) {
            R.chains = 1;
            R.nhalf = 2 * D;
            R.nt = 1 + R.nhalf;
            // behind a chain that reaches its stopping test within the round comes the search that follows: its
            // last node is then the root (the last estimate, not a midpoint of the final bracket)
            const int room = cap - R.nt;
            const bool enda = ia >= 1 && ia <= D, endb = ib >= 1 && ib <= D;
            if (room < 2 || !(enda || endb) || !swd_teamw_follow(S, tg, per, nx, &per_leaf, &clows)) return R;
            R.g0.oms = R.g1.oms = nx.oms;
            if (!per_leaf) {
                R.g0.base = S.ck - onea * dc;
                R.ngrp = 1;
                R.nt = swd_teamw_scan_layout(R.g0, R.nt, room, true, R.g0.base + dc > clows);
                return R;
            }
            double ra = S.c1, rb = S.c2;            // the chains' last nodes (one loop for both: the steps of a
            {                                       // chain depend on each other)
                const int na = enda ? ia : 0, nb = endb ? ib : 0, nmax = na > nb ? na : nb;
                for (int i = 0; i < nmax; i++) {
                    const double ta = 0.5 * (ra + S.ceval), tb = 0.5 * (S.ceval + rb);
                    ra = i < na ? ta : ra;
                    rb = i < nb ? tb : rb;
                }
            }
            if (enda && endb) {
                const int r0 = (room + 1) / 2;
                R.g0.base = ra - onea * dc;
                R.g1.base = rb - onea * dc;
                R.ngrp = 2;
                R.nt = swd_teamw_scan_layout(R.g0, R.nt, r0, true, R.g0.base + dc > clows);
                R.nt = swd_teamw_scan_layout(R.g1, R.nt, room - r0, true, R.g1.base + dc > clows);
                if (R.g1.entry < 0) R.ngrp = 1;
            } else if (enda || endb) {
                R.g0.base = (enda ? ra : rb) - onea * dc;
                R.ngrp = 1;
                R.nt = swd_teamw_scan_layout(R.g0, R.nt, room, true, R.g0.base + dc > clows);
            }
            return R;
        }
    }
#endif

    // ---- complete tree.  Evaluations left if every step from here on is a bisection: the pending one, plus
    // one per halving until the bracket is narrower than the stopping tolerance.
    int left = 1;
    {
        const double w = wa > wb ? wa : wb;
        left += (w > tol) + (w > 2.0 * tol) + (w > 4.0 * tol) + (w > 8.0 * tol) + (w > 16.0 * tol);
    }
    int depth = 0;
    while (depth < left - 1 && (4 << depth) - 1 <= cap) depth++;
    R.nhalf = (2 << depth) - 2;
    R.nt = 1 + R.nhalf;
    // the search that follows, behind the leaves if they are the last evaluations of this one
    if (depth != left - 1 || depth > 1 || R.nt >= cap || !swd_teamw_follow(S, tg, per, nx, &per_leaf, &clows))
        return R;
    R.g0.oms = R.g1.oms = nx.oms;
    const int room = cap - R.nt;
    if (depth == 0 || !per_leaf) {
        R.g0.base = ((S.pass == 0) ? S.ceval : S.ck) - onea * dc;
        R.ngrp = 1;
        R.nt = swd_teamw_scan_layout(R.g0, R.nt, room, true, R.g0.base + dc > clows);
    } else {
        // two leaves: the midpoints of the sub-brackets (slots 1 and 2)
        const int r0 = (room + 1) / 2;
        R.g0.base = 0.5 * (S.c1 + S.ceval) - onea * dc;
        R.g1.base = 0.5 * (S.ceval + S.c2) - onea * dc;
        R.ngrp = 2;
        R.nt = swd_teamw_scan_layout(R.g0, R.nt, r0, true, R.g0.base + dc > clows);
        R.nt = swd_teamw_scan_layout(R.g1, R.nt, room - r0, true, R.g1.base + dc > clows);
        if (R.g1.entry < 0) R.ngrp = 1;
    }
    return R;
}

// the scan group slot j (> nhalf) belongs to
BH_DEV const TeamwScan &swd_teamw_group(const TeamwRound &R, int j)
{
    return (R.ngrp > 1 && j >= R.g1.entry) ? R.g1 : R.g0;
}

// Slot j of the round: (c, omega).  NaN marks a scan slot beyond the search bounds.
BH_DEV void swd_teamw_trial(const TeamwRound &R, const SwdState &S, int j, double *c, double *om)
{
    const double dc = (double)0.005f;
    *c = S.ceval; *om = S.omega;
    if (j <= 0) return;
    if (j <= R.nhalf && R.chains) {
        // inner chains: A_i in slot 2i-1, B_i in slot 2i; midpoints taken exactly like nevill (:583,:661)
        const int i = (j + 1) >> 1;
        double a = (j & 1) ? S.c1 : S.c2;
        if (j & 1) { for (int t = 0; t < i; t++) a = 0.5 * (a + S.ceval); }
        else { for (int t = 0; t < i; t++) a = 0.5 * (S.ceval + a); }
        *c = a;
        return;
    }
    if (j <= R.nhalf) {
        // node j of the bisection tree in level order: level l = floor(log2(j+1)), position p in the
        // level; walk down from the pending trial, a bit of p per level (0: the value there has the
        // sign of del2 -> c2 = c3; 1: c1 = c3), taking midpoints exactly like nevill (:583,:661)
        const int l = swd_ilog2(j + 1);
        const int pth = j + 1 - (1 << l);
        double lo = S.c1, mid = S.ceval, hi = S.c2;
        for (int b = l - 1; b >= 0; b--) {
            if ((pth >> b) & 1) lo = mid; else hi = mid;
            mid = 0.5 * (lo + hi);
        }
        *c = mid;
        return;
    }
    const bool pmid = R.npm > 0 && j >= R.pm0;       // a predicted-cell midpoint of g0's scan
    const TeamwScan &g = pmid ? R.g0 : swd_teamw_group(R, j);
    *om = g.oms;
    if (!pmid && j == g.entry) { *c = g.base; return; }
    const int q = j - g.scan0;
    const int i = pmid ? R.pmi + (j - R.pm0) : g.stride == 2 ? q >> 1 : q;        // scan cell
    const bool mid = pmid || (g.stride == 2 && (q & 1));
    // base_0 = base, base_{n+1} = c_n = base_n + dc by repeated addition.  The scan stops at the first
    // base outside [cc, cfail) (swd_control: "c1 < cm or c1 >= betmx + dc -> no root"); the bases
    // increase, so it is enough to look at the first and at this cell's
    double b, cn;
    swd_scan_cell(g.base, i, &b, &cn);
    const bool ok = !(g.base < S.cc) && !(b >= S.cfail);
    *c = ok ? (mid ? 0.5 * (b + cn) : cn) : __builtin_nan("");
}

// ---- the bisection tree taken in one step -------------------------------------------------------
// What nevill does with the value of tree node j once the search arrives there (the `mid` part of
// swd_control, surfdisp96.f:599-640): the bracket update, the stopping test, and whether the next
// point is a bisection -- then the search moves on to one of the node's two children -- or not (a
// Neville step, or the root is found: the walk stops and swd_control itself takes over).  It only
// depends on values of the node's ancestors, so lane j evaluates node j for all nodes at once and the
// consuming loop follows the decisions down the tree instead of passing through swd_control once per
// level.  `del(slot)` returns the period-equation value of a slot.
enum { SWD_GO_LEFT = 0, SWD_GO_RIGHT = 1, SWD_GO_STOP = 2 };    // LEFT: c2 = c3 (child 2j+1), RIGHT: c1 = c3
// `out` says what swd_control does when it is handed the node's value after an arrival by bisection (j > 0) -- the
// lane of node j works that out too, for all nodes at once, so that the consuming loop can take the LAST node of a
// walk without a swd_control call (130 mostly scalar instructions on the control wave, 1.3 of them per round):
//   0  not worked out (node 0: its Neville table may hold more than the bracket ends): c1 .. d2 are the bracket
//      on ARRIVAL, swd_control takes the value
//   1  the root is found (surfdisp96.f:614)           c1 .. d2: the bracket AFTER the update (:605-612)
//   2  a bisection follows (:620-640, 661)            c3n = its point
//   3  a Neville step follows: from a fresh table of the two bracket ends it is the secant (:642-658), c3n
//   4  that step failed its guard (:651): the table is restarted all the same, a bisection follows at c3n
enum { SWD_OUT_CONTROL = 0, SWD_OUT_FINISH = 1, SWD_OUT_HALF = 2, SWD_OUT_NEVILLE = 3, SWD_OUT_NEVILLE_BAD = 4 };
struct TeamwNode {
    int go, out;
    double c1, d1, c2, d2;
    double c3n;
};
// `v.del(slot)`, `v.c(slot)`: period-equation value and trial velocity of a slot
template <class V>
BH_DEV TeamwNode swd_teamw_node(const SwdState &S, const TeamwRound &R, const V &v, int j)
{
    const double pct = (double)0.01f;
    TeamwNode n;
    n.c1 = S.c1; n.d1 = S.del1; n.c2 = S.c2; n.d2 = S.del2;
    double c3 = S.ceval;
    if (R.chains) {
        // node A_i (odd j): the search has set c2 = x and then c1 = A_1 .. A_{i-1}; B_i: the mirror image.
        // Only the pending trial and the chain's previous node enter.
        if (j > 0) {
            const double dx = v.del(0);
            const int p = j - 2;                    // the chain's previous node (slot <= 0: the bracket end itself)
            if (j & 1) {
                n.c2 = S.ceval; n.d2 = dx;
                if (p > 0) { n.c1 = v.c(p); n.d1 = v.del(p); }
            } else {
                n.c1 = S.ceval; n.d1 = dx;
                if (p > 0) { n.c2 = v.c(p); n.d2 = v.del(p); }
            }
            c3 = v.c(j);
        }
    } else {
        const int l = swd_ilog2(j + 1);
        const int pth = j + 1 - (1 << l);
        // values of the ancestors (level m: slot ((j + 1) >> (l - m)) - 1), fetched together: the walk
        // below is then arithmetic only
        double da[5];
        for (int m = 0; m < 5; m++) da[m] = v.del(m < l ? ((j + 1) >> (l - m)) - 1 : 0);
        for (int m = 0; m < 5; m++) {
            if (m < l) {
                const int bit = (pth >> (l - 1 - m)) & 1;
                if (bit) { n.c1 = c3; n.d1 = da[m]; } else { n.c2 = c3; n.d2 = da[m]; }
                c3 = 0.5 * (n.c1 + n.c2);
            }
        }
    }
    const double d3 = v.del(j);
    bool stop = false;
    int nev = 1;                                    // (a node below the root is reached by a bisection)
    if (j == 0) {
        nev = S.nev;
        if (S.st == SWD_ST_TOP)                     // label 100: step count, estimate outside the bracket
            stop = S.nctrl + 1 >= 100 || c3 < dmin(n.c1, n.c2) || c3 > dmax(n.c1, n.c2);
    }
    const double s13 = n.d1 - d3, s32 = d3 - n.d2;
    const bool neg = bh_signs_differ(d3, n.d1);
    const double c1 = neg ? n.c1 : c3, d1 = neg ? n.d1 : d3, c2 = neg ? c3 : n.c2, d2 = neg ? d3 : n.d2;
    if (fabs(c1 - c2) <= 1.e-6 * c1) stop = true;
    if (bh_signs_differ(s13, s32)) nev = 0;
    const double ss1 = fabs(d1), s1 = pct * ss1, ss2 = fabs(d2), s2 = pct * ss2;
    const bool do_half = s1 > ss2 || s2 > ss1 || nev == 0;
    const bool fin = stop;
    if (!do_half) stop = true;                                  // a Neville step comes next
    n.go = stop ? SWD_GO_STOP : neg ? SWD_GO_LEFT : SWD_GO_RIGHT;
    n.out = SWD_OUT_CONTROL;
    n.c3n = 0.0;
    if (j > 0) {
        // swd_neville with nev != 2: x(1), y(1) = c1, del1; x(2), y(2) = c2, del2; one step
        const double denom = d2 - d1, guard = 1.0e-10 * fabs(d2);
        const bool bad = fabs(denom) < guard;
        const double sec = (-d1 * c2 + d2 * c1) / denom, half = 0.5 * (c1 + c2);
        n.out = fin ? SWD_OUT_FINISH : do_half ? SWD_OUT_HALF : bad ? SWD_OUT_NEVILLE_BAD : SWD_OUT_NEVILLE;
        n.c3n = (n.out == SWD_OUT_NEVILLE) ? sec : half;
        n.c1 = c1; n.d1 = d1; n.c2 = c2; n.d2 = d2;
    }
    return n;
}

// Stores a Rayleigh layer matrix column by column (column i at p + 6 i: ca(1..5, i)).
BH_DEV void swd_teamw_store_dunkin(double *p, const Dunkin &a)
{
    p[0] = a.c11; p[1] = a.c12; p[2] = a.c13; p[3] = a.c14; p[4] = a.c15;
    p[5] = a.c21; p[6] = a.c22; p[7] = a.c23; p[8] = a.c24;
    p[9] = a.c31; p[10] = a.c32; p[11] = a.c33; p[12] = a.c34; p[13] = a.c35;
    p[14] = a.c41; p[15] = a.c42; p[16] = a.c43;
    p[17] = a.c51; p[18] = a.c53;
}
BH_DEV void swd_teamw_load_dunkin(const double *p, Dunkin &a)
{
    a.c11 = p[0]; a.c12 = p[1]; a.c13 = p[2]; a.c14 = p[3]; a.c15 = p[4];
    a.c21 = p[5]; a.c22 = p[6]; a.c23 = p[7]; a.c24 = p[8];
    a.c31 = p[9]; a.c32 = p[10]; a.c33 = p[11]; a.c34 = p[12]; a.c35 = p[13];
    a.c41 = p[14]; a.c42 = p[15]; a.c43 = p[16];
    a.c51 = p[17]; a.c53 = p[18];
}

// Layer matrix of layer r (0-based above llw) for the trial (c, omega) into slot `p`.
template <class Lay>
BH_DEV void swd_teamw_assemble_one(const Lay &lay, int ifunc, const SwdState &S, int r, double c, double om,
                                   double *p)
{
    const int i0 = S.llw - 1 + r;
    const double wvno = om / c;
    if (ifunc == 1) {
        LoveLayer o;
        swd_love_layer(lay, i0, wvno, om, o);
        p[0] = o.cosq; p[1] = o.y; p[2] = o.z; p[3] = o.xmu;
    } else {
        double omega = om;
        if (omega < 1.0e-4) omega = 1.0e-4;
        Dunkin a;
        swd_ray_layer_matrix(lay, i0, wvno, wvno * wvno, omega, a);
        swd_teamw_store_dunkin(p, a);
    }
}

// Period-equation value of the trial (c, omega) from its nlm assembled layer matrices, slots slot0 .. of the
// store `mats` (generic form: one lane per trial; the device runs Rayleigh trials on quads, kernels.hip).
template <class Lay>
BH_DEV double swd_teamw_chain_one(const Lay &lay, int ifunc, const SwdState &S, double c, double om,
                                  const double *mats, int slot0)
{
    const int nlm = S.mmax - S.llw;
    const double wvno = om / c;
    if (ifunc == 1) {
        double e1, e2;
        swd_love_halfspace(lay, S.mmax, wvno, om, e1, e2);
        for (int r = nlm - 1; r >= 0; r--) {
            const double *p = mats + swd_mat_off(slot0 + r);
            LoveLayer o;
            o.cosq = p[0]; o.y = p[1]; o.z = p[2]; o.xmu = p[3];
            swd_love_apply(e1, e2, o);
        }
        return e1;
    }
    double omega = om;
    if (omega < 1.0e-4) omega = 1.0e-4;
    double e[5];
    swd_ray_halfspace(lay, S.mmax, wvno, wvno * wvno, omega, e);
    for (int r = nlm - 1; r >= 0; r--) {
        Dunkin a;
        swd_teamw_load_dunkin(mats + swd_mat_off(slot0 + r), a);
        swd_dunkin_apply(e, a);
    }
    return (S.llw != 1) ? swd_ray_water(lay, wvno, omega, e) : e[0];
}

// Feeds the round's values to the search: a trial is consumed iff the search asks for exactly its
// (omega, c) next.  When a root search ends inside the round the driver runs here (results stored,
// next period / second solve / next mode set up) and matching goes on with the speculative trials;
// a task boundary (next model) ends the round.
//   vals   `int find(double om, double c)`: slot computed at exactly that point, or -1;
//          `double del(int j)`, `double c(int j)`: value and velocity of slot j;
//          `int run(int first, int stride, int count, bool neg)`: how many of the slots first,
//          first + stride, ... (at most count) in a row are valid scan trials whose value has sign
//          bit `neg`;
//          `int go(const SwdState &, int j)`, `TeamwNode node(const SwdState &, int j)`: swd_teamw_node
//          of tree node j for the state the round was planned with;
//          `int chain_run(const SwdState &, int first, int dir, int max)`: how many of the chain nodes first,
//          first + 2, ... (at most max) in a row decide `dir`;
//          `probe(int)`, `count(int, int)`: cycle probes of the diagnostic build, else empty.
// Scan trials without a sign change are the bulk of all evaluations (two thirds) and
// each costs a pass through swd_control although all it does is "c1 = c2, del1 = del2, next grid
// point" (surfdisp96.f:461-470): a run of them is taken in one step.
// A chain of bisections down the tree is taken in one step too (swd_teamw_node): the state is set to
// what swd_control would have left on arrival at the last node of the chain, and swd_control runs for
// that node only.  `tree` = false walks the tree one swd_control call per node (the replay checks that
// both leave the same state).
// Returns the number of values consumed (= evaluations of the reference).
template <class Lay, class Src, class Vals, class Nev>
BH_DEV int swd_teamw_consume(SwdState &S, Nev &nv, Lay &lay, Src &src, const SwdTargetDev &tg,
                             const double *BH_RESTRICT per, int wss, const TeamwRound &R, const Vals &vals,
                             bool tree = true)
{
    const double dc = (double)0.005f;
    int used = 0;
    while (used < R.nt) {
        // slot 0 IS the pending evaluation (no comparison: a NaN model has a NaN trial velocity, which
        // equals nothing, and the search must still advance -- to its bracketing step cap -- and end)
        int j = used == 0 ? 0 : vals.find(S.omega, S.ceval);
        vals.probe(4);
        if (j < 0) break;
        bool handled = false;
        if (tree && used == 0 && R.nhalf > 0 && S.nctrl < 80) {       // (refinement round: ST_TOP / ST_MID)
            int last = 0, lev = 0;
            if (R.chains) {
                // x, then down chain A (x went left: c2 = x) or B while the far end keeps moving towards x
                const int g0 = vals.go(S, 0);
                if (g0 != SWD_GO_STOP) {
                    const int inner = g0 == SWD_GO_LEFT ? SWD_GO_RIGHT : SWD_GO_LEFT;
                    last = g0 == SWD_GO_LEFT ? 1 : 2;
                    // nodes last, last + 2, ... of the chain, as long as each sends the search on inwards
                    const int steps = vals.chain_run(S, last, inner, (R.nhalf - last) >> 1);
                    last += 2 * steps;
                    lev = 1 + steps;
                }
            } else {
#if !defined(BH_HOSTSIM)
#pragma clang loop unroll(disable)
#endif
                while (2 * last + 2 <= R.nhalf) {                     // `last` has children in the tree
                    const int go = vals.go(S, last);
                    if (go == SWD_GO_STOP) break;
                    last = 2 * last + 1 + go;
                    lev++;
                }
            }
            if (lev > 0) {
                // (`last` > 0: the lane of that node has worked out what swd_control does with its value)
                const TeamwNode a = vals.node(S, last);
                S.nctrl += lev + 1 - (S.st == SWD_ST_MID ? 1 : 0);    // (ST_MID does not count its step)
                S.c1 = a.c1; S.del1 = a.d1; S.c2 = a.c2; S.del2 = a.d2;
                S.del3 = vals.del(last);
                S.c3 = vals.c(last);
                S.nev = 1; S.m = 1; S.st = SWD_ST_TOP;
                if (a.out == SWD_OUT_FINISH) {                        // label 1000 of nevill + getsol's tail
                    S.ceval = S.c3;
                    S.c1 = S.c3;
                    S.ev = (S.c1 > (double)S.betmx) ? SWD_EV_NOROOT : SWD_EV_SOLVED;
                } else {
                    if (a.out != SWD_OUT_HALF) {                      // the table the Neville step leaves behind
                        const bool ok = a.out == SWD_OUT_NEVILLE;
                        swd_nev_restart(nv, ok ? a.c3n : a.c1, a.d1, a.c2, a.d2);
                        if (ok) { S.nev = 2; S.m = 2; }
                    }
                    S.c3 = a.c3n;
                    S.ceval = S.c3;
                }
                used = lev + 1;
                handled = true;
                vals.count(12, lev);
                vals.count(11, 1);
            }
            vals.probe(15);
        }
        if (!handled && S.st == SWD_ST_B && S.idir > 0 && R.ngrp > 0 && (R.nhalf == 0 || j > R.nhalf)) {
            const TeamwScan &g = swd_teamw_group(R, j);
            const int q = j - g.scan0, i = q / g.stride;
            if (q >= 0 && q == i * g.stride && i < g.nscan) {
                int m = vals.run(j, g.stride, g.nscan - i, dsign1(S.del1) < 0.0);
                if (m > SWD_MAX_BRACKET_STEPS - S.nbrk) m = 0;          // (the hard step cap: one by one)
                if (m > 0) {
                    // m steps of label 1000 without a sign change.  Every base of the run but the
                    // last lies inside the bounds (else the following slots would be NaN).
                    const int last = j + (m - 1) * g.stride;
                    S.del2 = vals.del(last);
                    S.c1 = vals.c(last);
                    S.del1 = S.del2;
                    used += m;
                    if (S.c1 < S.cc || S.c1 >= S.cfail) {
                        S.nbrk += m - 1;
                        S.ev = SWD_EV_NOROOT;
                    } else {
                        S.nbrk += m;
                        S.c2 = swd_bracket_next(S.c1, S.idir, S.clow, dc);
                        S.ceval = S.c2;
                    }
                    handled = true;
                    vals.probe(5);
                    vals.count(10, 1);
                }
            }
        }
        if (!handled) {
            swd_control(S, vals.del(j), nv);
            used++;
            vals.probe(6);
            vals.count(11, 1);
        }
        // the one place of the loop where a pending event is taken (root found / no root: results stored, next
        // period or second solve set up) -- three inlined copies of the driver were 1 200 instructions of code
        if (S.ev != SWD_EV_NONE) {
            swd_events(S, lay, src, tg, per, wss, false);
            vals.probe(7);
            if (S.st == SWD_ST_DONE) break;          // task finished: the next model is fetched by the caller
        }
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
