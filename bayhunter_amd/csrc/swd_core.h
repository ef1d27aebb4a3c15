// swd_core.h -- one lane = one (layered model, dispersion target): exact replay of the reference's
// surface-wave search (surfdisp96.f:55-360 driver, :390-482 getsol, :557-686 nevill/half) around a
// SINGLE period-equation call site.
//
// Why a state machine.  The reference is a nest of routines that call the period equation `dltar`
// from six places (getsol x2, half x3 call sites in nevill, Neville step).  Executed as written, 64
// lanes holding 64 different models would sit at different call sites and the wave would run the
// expensive part (dltar: the whole layer stack, ~95 % of the work) once per call site.  Here every
// lane carries its search state explicitly, the loop body evaluates dltar exactly once at one
// program point for all lanes, and the (cheap, divergent) control logic only decides which trial
// velocity a lane evaluates next.  The sequence of trial velocities per lane is the reference's,
// value for value: same bracketing grid (repeated addition of dc), same Neville / bisection
// decisions, same real*4 / real*8 mixture.
//
// dltar4 / dltar1 / var / dnka / normc follow surfdisp96.f:710-1068 expression by expression
// (build with -ffp-contract=off).
#pragma once
#include "bh_common.h"
#include "bh_math.h"

namespace bh {

// ---- period equations --------------------------------------------------------------------------
struct VarProd {
    double a0, cpcq, cpy, cpz, cqw, cqx, xy, xz, wy, wz;
};

// surfdisp96.f:874-991.  Returns w and cosp through pointers (needed by the water-layer tail).
BH_DEV void swd_var(double p, double q, double ra, double rb, double wvno, double xka, double xkb,
                    double dpth, double *w_out, double *cosp_out, VarProd &o)
{
    double w, x, y, z, cosp, cosq, sinp, sinq, fac;
    double pex = 0.0, sex = 0.0;
    if (wvno < xka) {
        bh_sincos(p, &sinp, &cosp);
        w = xdiv(sinp, ra);
        x = -ra * sinp;
    } else if (wvno == xka) {
        cosp = 1.0; w = dpth; x = 0.0;
    } else {
        pex = p;
        fac = 0.0;
        if (p < 16) fac = bh_exp_bounded(-2.0 * p);
        cosp = (1.0 + fac) * 0.5;
        sinp = (1.0 - fac) * 0.5;
        w = xdiv(sinp, ra);
        x = ra * sinp;
    }
    if (wvno < xkb) {
        bh_sincos(q, &sinq, &cosq);
        y = xdiv(sinq, rb);
        z = -rb * sinq;
    } else if (wvno == xkb) {
        cosq = 1.0; y = dpth; z = 0.0;
    } else {
        sex = q;
        fac = 0.0;
        if (q < 16) fac = bh_exp_bounded(-2.0 * q);
        cosq = (1.0 + fac) * 0.5;
        sinq = (1.0 - fac) * 0.5;
        y = xdiv(sinq, rb);
        z = rb * sinq;
    }
    double exa = pex + sex;
    double a0 = 0.0;
    if (exa < 60.0) a0 = bh_exp_bounded(-exa);
    o.a0 = a0;
    o.cpcq = cosp * cosq; o.cpy = cosp * y; o.cpz = cosp * z;
    o.cqw = cosq * w;     o.cqx = cosq * x;
    o.xy = x * y; o.xz = x * z; o.wy = w * y; o.wz = w * z;
    *w_out = w;
    *cosp_out = cosp;
}

// Dunkin's 5x5 layer matrix (dnka, surfdisp96.f:1024-1068).  Its 25 entries are 19 distinct values
// (ca25=ca14, ca44=ca22, ca45=ca12, ca52=ca41, ca54=ca21, ca55=ca11); they are kept as scalars.
struct Dunkin {
    double c11, c12, c13, c14, c15, c21, c22, c23, c24, c31, c32, c33, c34, c35, c41, c42, c43, c51, c53;
};
enum { SWD_NCA = 19 };

BH_DEV void swd_dnka(Dunkin &a, double wvno2, double gam, double gammk, double rho, const VarProd &v)
{
    const double one = 1.0, two = 2.0;
    double gamm1 = gam - one, twgm1 = gam + gamm1, gmgmk = gam * gammk, gmgm1 = gam * gamm1,
           gm1sq = gamm1 * gamm1, rho2 = rho * rho, a0pq = v.a0 - v.cpcq;
    a.c11 = v.cpcq - two * gmgm1 * a0pq - gmgmk * v.xz - wvno2 * gm1sq * v.wy;
    const Recip by_rho = recip_of(rho);
    a.c12 = qdiv(wvno2 * v.cpy - v.cqx, by_rho);
    a.c13 = qdiv(-(twgm1 * a0pq + gammk * v.xz + wvno2 * gamm1 * v.wy), by_rho);
    a.c14 = qdiv(v.cpz - wvno2 * v.cqw, by_rho);
    a.c15 = xdiv(-(two * wvno2 * a0pq + v.xz + wvno2 * wvno2 * v.wy), rho2);
    a.c21 = (gmgmk * v.cpz - gm1sq * v.cqw) * rho;
    a.c22 = v.cpcq;
    a.c23 = gammk * v.cpz - gamm1 * v.cqw;
    a.c24 = -v.wz;
    a.c41 = (gm1sq * v.cpy - gmgmk * v.cqx) * rho;
    a.c42 = -v.xy;
    a.c43 = gamm1 * v.cpy - gammk * v.cqx;
    a.c51 = -(two * gmgmk * gm1sq * a0pq + gmgmk * gmgmk * v.xz + gm1sq * gm1sq * v.wy) * rho2;
    a.c53 = -(gammk * gamm1 * twgm1 * a0pq + gam * gammk * gammk * v.xz + gamm1 * gm1sq * v.wy) * rho;
    double t = -two * wvno2;
    a.c31 = t * a.c53;
    a.c32 = t * a.c43;
    a.c33 = v.a0 + two * (v.cpcq - a.c11);
    a.c34 = t * a.c23;
    a.c35 = t * a.c13;
}

// ee = e * ca (surfdisp96.f:838-844, accumulated from 0.0 in j order) followed by normc (:995-1020):
// divide by the max-abs (the log of the scale is dead in the reference's caller).
BH_DEV void swd_dunkin_apply(double e[5], const Dunkin &a)
{
    double e1 = e[0], e2 = e[1], e3 = e[2], e4 = e[3], e5 = e[4];
    double ee1 = ((((0.0 + e1 * a.c11) + e2 * a.c21) + e3 * a.c31) + e4 * a.c41) + e5 * a.c51;
    double ee2 = ((((0.0 + e1 * a.c12) + e2 * a.c22) + e3 * a.c32) + e4 * a.c42) + e5 * a.c41;
    double ee3 = ((((0.0 + e1 * a.c13) + e2 * a.c23) + e3 * a.c33) + e4 * a.c43) + e5 * a.c53;
    double ee4 = ((((0.0 + e1 * a.c14) + e2 * a.c24) + e3 * a.c34) + e4 * a.c22) + e5 * a.c21;
    double ee5 = ((((0.0 + e1 * a.c15) + e2 * a.c14) + e3 * a.c35) + e4 * a.c12) + e5 * a.c11;
    // t1 = max |ee_i| (normc's chain of `if (abs(ee_i) > t1) t1 = abs(ee_i)` from t1 = 0: a NaN entry is
    // skipped there, and fmax returns its other argument for a NaN: same value in every case)
    double t1 = bh_fmax(bh_fmax(bh_fmax(bh_fmax(bh_fmax(0.0, fabs(ee1)), fabs(ee2)), fabs(ee3)), fabs(ee4)), fabs(ee5));
    if (t1 < 1.e-40) t1 = 1.0;
    const Recip by_t1 = recip_of(t1);
    e[0] = qdiv(ee1, by_t1); e[1] = qdiv(ee2, by_t1); e[2] = qdiv(ee3, by_t1);
    e[3] = qdiv(ee4, by_t1); e[4] = qdiv(ee5, by_t1);
}

// Dunkin matrix of layer i0 (0-based) at (wvno, omega): the loop body of surfdisp96.f:813-837.
// Independent of the vector e: layers can be assembled in any order / in parallel.
template <class Lay>
BH_DEV void swd_ray_layer_matrix(const Lay &lay, int i0, double wvno, double wvno2, double omega,
                                 Dunkin &a)
{
    double am = (double)lay.a(i0), bm = (double)lay.b(i0);
    double xka = xdiv(omega, am);
    double xkb = xdiv(omega, bm);
    double t = xdiv(bm, omega);
    double gammk = 2.0 * t * t;
    double gam = gammk * wvno2;
    double wvnop = wvno + xka, wvnom = fabs(wvno - xka);
    double ra = xsqrt(wvnop * wvnom);
    wvnop = wvno + xkb; wvnom = fabs(wvno - xkb);
    double rb = xsqrt(wvnop * wvnom);
    double dpth = (double)lay.d(i0);
    double rho1 = (double)lay.rho(i0);
    double p = ra * dpth, q = rb * dpth, w, cosp;
    VarProd v;
    swd_var(p, q, ra, rb, wvno, xka, xkb, dpth, &w, &cosp, v);
    swd_dnka(a, wvno2, gam, gammk, rho1, v);
}

// E vector of the bottom half-space, surfdisp96.f:785-808 (omega already clamped).  (Lean quotient /
// root sequences: bit-identical for these normal-range operands, bh_selftest_division.)
template <class Lay>
BH_DEV void swd_ray_halfspace(const Lay &lay, int mmax, double wvno, double wvno2, double omega,
                              double e[5])
{
    const double am = (double)lay.a(mmax - 1), bm = (double)lay.b(mmax - 1);
    double xka = xdiv(omega, am);
    double xkb = xdiv(omega, bm);
    double wvnop = wvno + xka, wvnom = fabs(wvno - xka);
    double ra = xsqrt(wvnop * wvnom);
    wvnop = wvno + xkb; wvnom = fabs(wvno - xkb);
    double rb = xsqrt(wvnop * wvnom);
    double t = xdiv(bm, omega);
    double gammk = 2.0 * t * t, gam = gammk * wvno2, gamm1 = gam - 1.0;
    double rho1 = (double)lay.rho(mmax - 1);
    e[0] = rho1 * rho1 * (gamm1 * gamm1 - gam * gammk * ra * rb);
    e[1] = -rho1 * ra;
    e[2] = rho1 * (gamm1 - gammk * ra * rb);
    e[3] = rho1 * rb;
    e[4] = wvno2 - ra * rb;
}

// water layer on top, surfdisp96.f:850-867
template <class Lay>
BH_DEV double swd_ray_water(const Lay &lay, double wvno, double omega, const double e[5])
{
    double xka = omega / (double)lay.a(0);
    double wvnop = wvno + xka, wvnom = fabs(wvno - xka);
    double ra = sqrt(wvnop * wvnom);
    double dpth = (double)lay.d(0);
    double rho1 = (double)lay.rho(0);
    double p = ra * dpth, znul = 1.0e-05, w, cosp;
    VarProd v;
    swd_var(p, znul, ra, znul, wvno, xka, znul, dpth, &w, &cosp, v);
    double w0 = -rho1 * w;
    return cosp * e[0] + w0 * e[1];
}

// Rayleigh / P-SV period equation, surfdisp96.f:773-871.  Lay: d(i),a(i),b(i),rho(i), 0-based.
template <class Lay>
BH_DEV double swd_dltar4(const Lay &lay, int mmax, int llw, double wvno, double omga)
{
    double e[5];
    double omega = omga;
    if (omega < 1.0e-4) omega = 1.0e-4;
    double wvno2 = wvno * wvno;
    swd_ray_halfspace(lay, mmax, wvno, wvno2, omega, e);
    for (int m = mmax - 1; m >= llw; m--) {  // Fortran index m; 0-based layer m-1
        Dunkin a;
        swd_ray_layer_matrix(lay, m - 1, wvno, wvno2, omega, a);
        swd_dunkin_apply(e, a);
    }
    if (llw != 1) return swd_ray_water(lay, wvno, omega, e);
    return e[0];
}

// ---- Love / SH -----------------------------------------------------------------------------------
// Per-layer quantities of the Thomson-Haskell recursion (surfdisp96.f:733-757), independent of (e1,e2)
struct LoveLayer {
    double cosq, y, z, xmu;
};
template <class Lay>
BH_DEV void swd_love_layer(const Lay &lay, int i0, double wvno, double omega, LoveLayer &o)
{
    double beta1 = (double)lay.b(i0);
    double rho1 = (double)lay.rho(i0);
    double dm = (double)lay.d(i0);
    o.xmu = rho1 * beta1 * beta1;
    double xkb = xdiv(omega, beta1);
    double wvnop = wvno + xkb, wvnom = fabs(wvno - xkb);
    double rb = xsqrt(wvnop * wvnom);
    double q = dm * rb, sinq, fac;
    if (wvno < xkb) {
        bh_sincos(q, &sinq, &o.cosq);
        o.y = xdiv(sinq, rb);
        o.z = -rb * sinq;
    } else if (wvno == xkb) {
        o.cosq = 1.0; o.y = dm; o.z = 0.0;
    } else {
        fac = 0.0;
        if (q < 16) fac = bh_exp_bounded(-2.0 * q);
        o.cosq = (1.0 + fac) * 0.5;
        sinq = (1.0 - fac) * 0.5;
        o.y = xdiv(sinq, rb);
        o.z = rb * sinq;
    }
}
// surfdisp96.f:758-765
BH_DEV void swd_love_apply(double &e1, double &e2, const LoveLayer &o)
{
    double e10 = e1 * o.cosq + e2 * o.xmu * o.z;
    double e20 = xdiv(e1 * o.y, o.xmu) + e2 * o.cosq;
    double xnor = fabs(e10), ynor = fabs(e20);
    if (ynor > xnor) xnor = ynor;                     // (surfdisp96.f:760-763, as written: xnor = |e10| may be NaN)
    if (xnor < 1.e-40) xnor = 1.0;
    const Recip by_nor = recip_of(xnor);
    e1 = qdiv(e10, by_nor);
    e2 = qdiv(e20, by_nor);
}
// surfdisp96.f:723-730
template <class Lay>
BH_DEV void swd_love_halfspace(const Lay &lay, int mmax, double wvno, double omega, double &e1, double &e2)
{
    double beta1 = (double)lay.b(mmax - 1), rho1 = (double)lay.rho(mmax - 1);
    double xkb = omega / beta1;
    double wvnop = wvno + xkb, wvnom = fabs(wvno - xkb);
    double rb = sqrt(wvnop * wvnom);
    e1 = rho1 * rb;
    e2 = 1.0 / (beta1 * beta1);
}

// Love / SH period equation, surfdisp96.f:710-769.
template <class Lay>
BH_DEV double swd_dltar1(const Lay &lay, int mmax, int llw, double wvno, double omega)
{
    double e1, e2;
    swd_love_halfspace(lay, mmax, wvno, omega, e1, e2);
    for (int m = mmax - 1; m >= llw; m--) {
        LoveLayer o;
        swd_love_layer(lay, m - 1, wvno, omega, o);
        swd_love_apply(e1, e2, o);
    }
    return e1;
}

// ---- model preparation ---------------------------------------------------------------------------
// gtsolh, surfdisp96.f:367-388: five real*4 Newton steps on the half-space Rayleigh equation.
BH_DEV float swd_gtsolh(float a, float b)
{
    float c = 0.95f * b;
    for (int i = 0; i < 5; i++) {
        float gamma = b / a, kappa = c / b;
        float k2 = kappa * kappa;
        float gk = gamma * kappa, gk2 = gk * gk;
        float fac1 = sqrtf(1.0f - gk2), fac2 = sqrtf(1.0f - k2);
        float tk = 2.0f - k2;
        float fr = tk * tk - 4.0f * fac1 * fac2;
        float frp = -4.0f * (2.0f - k2) * kappa + 4.0f * fac2 * gamma * gamma * kappa / fac1 +
                    4.0f * fac1 * kappa / fac2;
        frp = frp / b;
        c = c - fr / frp;
    }
    return c;
}

// real*4 ** real*4 (surfdisp96.f:547): glibc's powf is correctly rounded but for rare near-ties, so
// the device evaluates in fp64 and rounds once; the host replay uses powf itself.
#if defined(BH_HOSTSIM)
BH_DEV float swd_powf(float x, float y) { return powf(x, y); }
#else
BH_DEV float swd_powf(float x, float y) { return (float)::pow((double)x, (double)y); }
#endif

// sphere, surfdisp96.f:486-553, both calls (iflag 0 then iflag 1 for this lane's wave type) folded
// into one pass: d,a,b are transformed, rho is scaled by btp**(-5) (Love) / btp**(-2.275) (Rayleigh).
template <class Lay>
#if defined(BH_HOSTSIM)
BH_DEV void swd_sphere(Lay &lay, int mmax, int ifunc)
#else
__device__ __noinline__ void swd_sphere(Lay &lay, int mmax, int ifunc)
#endif
{
    double ar = 6370.0, dr = 0.0, r0 = ar, r1, z0, z1, tmp;
    lay.set_d(mmax - 1, 1.0f);
    for (int i = 0; i < mmax; i++) {
        dr = dr + (double)lay.d(i);
        r1 = ar - dr;
        z0 = ar * log(ar / r0);
        z1 = ar * log(ar / r1);
        lay.set_d(i, (float)(z1 - z0));
        tmp = (ar + ar) / (r0 + r1);
        lay.set_a(i, (float)((double)lay.a(i) * tmp));
        lay.set_b(i, (float)((double)lay.b(i) * tmp));
        float btp = (float)tmp, rtp = lay.rho(i);
        if (ifunc == 1) {
            float x2 = btp * btp, x4 = x2 * x2, x5 = x4 * btp;
            lay.set_rho(i, rtp * (1.0f / x5));
        } else {
            lay.set_rho(i, rtp * swd_powf(btp, -2.275f));
        }
        r0 = r1;
    }
    lay.set_d(mmax - 1, 0.0f);
}

// Results are real*4 values promoted to fp64 (surfdisp96.f:298,303,306).  A lane finishes its periods
// one at a time; storing them in pairs (16 bytes) halves the scattered 8-byte stores, which the
// memory side tallies at 32-64 bytes each.
BH_DEV void swd_store_pair(double *p, float a, float b)
{
#if defined(BH_HOSTSIM)
    p[0] = (double)a; p[1] = (double)b;
#else
    typedef double d2_t __attribute__((ext_vector_type(2), aligned(8)));
    d2_t v;
    v.x = (double)a; v.y = (double)b;
    *(d2_t *)p = v;
#endif
}
BH_DEV void swd_put(double *out, int k, int kmax, float v, float &pend)      // k: 1-based period
{
    if (k & 1) {
        if (k == kmax) out[k - 1] = (double)v;
        else pend = v;
    } else {
        swd_store_pair(out + k - 2, pend, v);
    }
}

// ---- the search ----------------------------------------------------------------------------------
struct SwdState;
BH_DEV void swd_put_direct(SwdState &S, int k, int kmax, float v);
BH_DEV void swd_zero_direct(SwdState &S, int k, int kmax);
struct SwdTargetDev {
    int iwave, igr, mode, iflsph, nper, per_off, out_off, _pad;
};

enum { SWD_MAX_BRACKET_STEPS = 100000 };
enum { SWD_ST_A = 0, SWD_ST_B = 1, SWD_ST_TOP = 2, SWD_ST_MID = 3, SWD_ST_DONE = 4 };

// ---- search state -------------------------------------------------------------------------------
// Everything `surfdisp96` -> `getsol` -> `nevill` keep in local variables, SAVEd variables and on the
// call stack, flattened into one record so that the search can be suspended at every period-
// equation evaluation (swd_lane: one evaluation per loop trip; swd_team.h: several speculative
// evaluations per round).
enum { SWD_EV_NONE = 0, SWD_EV_FETCH, SWD_EV_BEGIN_PERIOD, SWD_EV_SOLVED, SWD_EV_NOROOT };

struct SwdState {
    // per-task constants
    int mmax, llw, err;
    float betmx;
    double cc, cfail;                 // cc = c1 = cm start value; cfail = betmx + dc
    double *out, *cws, *cbws;
    // period / mode bookkeeping
    int iq, k, ift, pass, ifirst;
    float t1a, t1b;
    float pend;                       // value of an odd period waiting to be stored with the next one
    double omega, cprev, ck;
    // getsol / nevill
    int st, ev, idir, nev, nctrl, m, nbrk;
    double c1, c2, c3, del1, del2, del3, clow, del1st, ceval;
};

// The (x, y) table of nevill's inverse interpolation (surfdisp96.f:572,642-658; up to 11 points),
// kept apart from the state because where it should live depends on the kernel:
//   NevRegs  in registers (throughput kernel: one search per lane; indexed through if-chains)
//   NevMem   in memory (LDS of a team / host arrays): 12 doubles each, entries 1..11 used.  A team runs
//            the control code redundantly on all lanes, so 44 fewer live registers and no if-chains
//            shorten every step of it.
struct NevRegs {
    double x1, x2, x3, x4, x5, x6, x7, x8, x9, x10, x11;
    double y1, y2, y3, y4, y5, y6, y7, y8, y9, y10, y11;
#if defined(BH_LANE_PROFILE)
    unsigned long long cycles;        // diagnostic build: shader-clock cycles the WAVE spends in swd_neville
                                      // (booked on the first lane that takes the step)
#endif
};
struct NevMem {
    double *x, *y;
};
BH_DEV void swd_nev_init(NevRegs &n)
{
    n.x1 = n.x2 = n.x3 = n.x4 = n.x5 = n.x6 = n.x7 = n.x8 = n.x9 = n.x10 = n.x11 = 0;
    n.y1 = n.y2 = n.y3 = n.y4 = n.y5 = n.y6 = n.y7 = n.y8 = n.y9 = n.y10 = n.y11 = 0;
}
BH_DEV void swd_nev_init(NevMem &n)
{
    for (int i = 0; i < 12; i++) { n.x[i] = 0; n.y[i] = 0; }
}

// One pass of Neville inverse interpolation (surfdisp96.f:642-658).  nev2: append (c3, del3) as point
// m+1, else restart the table from the bracket ends.  Returns false when the guard
// abs(denom) < 1e-10 abs(y(m+1)) fires (-> bisection); else *x1 receives the new estimate x(1).
BH_DEV bool swd_neville(NevRegs &n, int &m, bool nev2, double c1, double del1, double c2, double del2,
                        double c3, double del3, double *x1)
{
#if defined(BH_LANE_PROFILE)
    const unsigned long long lp_t0 = clock64();
#endif
    double ym1;                       // y(m+1)
    if (nev2) {                       // x(m+1)=c3, y(m+1)=del3
        ym1 = del3;
        // (tables longer than three points are rare: the two outer tests let a wave skip them)
        if (m <= 3) {
            if (m == 1) { n.x2 = c3; n.y2 = del3; } else if (m == 2) { n.x3 = c3; n.y3 = del3; }
            else { n.x4 = c3; n.y4 = del3; }
        } else if (m <= 6) {
            if (m == 4) { n.x5 = c3; n.y5 = del3; } else if (m == 5) { n.x6 = c3; n.y6 = del3; }
            else { n.x7 = c3; n.y7 = del3; }
        } else {
            if (m == 7) { n.x8 = c3; n.y8 = del3; } else if (m == 8) { n.x9 = c3; n.y9 = del3; }
            else if (m == 9) { n.x10 = c3; n.y10 = del3; } else { n.x11 = c3; n.y11 = del3; }
        }
    } else {
        n.x1 = c1; n.y1 = del1; n.x2 = c2; n.y2 = del2; m = 1;
        ym1 = del2;
    }
    // j = m .. 1 (surfdisp96.f:649-654)
    bool bad = false;
    const double guard = 1.0e-10 * fabs(ym1);
#define BH_NEV_STEP(J, XJ, YJ, XJ1)                                               \
    if (!bad && m >= J) {                                                         \
        double denom = ym1 - n.YJ;                                                \
        if (fabs(denom) < guard) bad = true;                                      \
        else n.XJ = (-n.YJ * n.XJ1 + ym1 * n.XJ) / denom;                         \
    }
    if (m >= 7) {
        BH_NEV_STEP(10, x10, y10, x11)
        BH_NEV_STEP(9, x9, y9, x10)
        BH_NEV_STEP(8, x8, y8, x9)
        BH_NEV_STEP(7, x7, y7, x8)
    }
    if (m >= 4) {
        BH_NEV_STEP(6, x6, y6, x7)
        BH_NEV_STEP(5, x5, y5, x6)
        BH_NEV_STEP(4, x4, y4, x5)
    }
    BH_NEV_STEP(3, x3, y3, x4)
    BH_NEV_STEP(2, x2, y2, x3)
    BH_NEV_STEP(1, x1, y1, x2)
#undef BH_NEV_STEP
    *x1 = n.x1;
#if defined(BH_LANE_PROFILE)
    {
        const unsigned long long act = __ballot(1);
        if ((int)(threadIdx.x & 63) == __ffsll((long long)act) - 1) n.cycles += clock64() - lp_t0;
    }
#endif
    return !bad;
}
BH_DEV bool swd_neville(NevMem &n, int &m, bool nev2, double c1, double del1, double c2, double del2,
                        double c3, double del3, double *x1)
{
    double ym1;
    if (nev2) {
        n.x[m + 1] = c3; n.y[m + 1] = del3;
        ym1 = del3;
    } else {
        n.x[1] = c1; n.y[1] = del1; n.x[2] = c2; n.y[2] = del2; m = 1;
        ym1 = del2;
    }
    const double guard = 1.0e-10 * fabs(ym1);
    double xj1 = n.x[m + 1];          // x(j+1) of the running pass
    for (int j = m; j >= 1; j--) {
        const double yj = n.y[j];
        const double denom = ym1 - yj;
        if (fabs(denom) < guard) return false;
        xj1 = (-yj * xj1 + ym1 * n.x[j]) / denom;
        n.x[j] = xj1;
    }
    *x1 = xj1;
    return true;
}

// The table a Neville step from the bracket ends leaves behind (swd_neville with nev2 = false): x(1) = the new
// estimate -- or c1 when the step failed its guard --, y(1) = del1, x(2) = c2, y(2) = del2.
BH_DEV void swd_nev_restart(NevMem &n, double x1, double del1, double c2, double del2)
{
    n.x[1] = x1; n.y[1] = del1; n.x[2] = c2; n.y[2] = del2;
}
BH_DEV void swd_nev_restart(NevRegs &n, double x1, double del1, double c2, double del2)
{
    n.x1 = x1; n.y1 = del1; n.x2 = c2; n.y2 = del2;
}

BH_DEV void swd_put_direct(SwdState &S, int k, int kmax, float v) { swd_put(S.out, k, kmax, v, S.pend); }
BH_DEV void swd_zero_direct(SwdState &S, int k, int kmax)
{
    if (!(k & 1)) S.out[k - 2] = (double)S.pend;      // an odd period's value was still waiting
    for (int i = k; i <= kmax; i++) S.out[i - 1] = 0.0;
}

BH_DEV void swd_state_init(SwdState &S)
{
    S.mmax = 1; S.llw = 1; S.err = 0; S.betmx = 0.f; S.cc = 0; S.cfail = 0;
    S.out = nullptr; S.cws = nullptr; S.cbws = nullptr;
    S.iq = 1; S.k = 1; S.ift = 999; S.pass = 0; S.ifirst = 0; S.t1a = 0; S.t1b = 0; S.pend = 0;
    S.omega = 0; S.cprev = 0; S.ck = 0;
    S.st = SWD_ST_DONE; S.ev = SWD_EV_FETCH; S.idir = 1; S.nev = 1; S.nctrl = 1; S.m = 1; S.nbrk = 0;
    S.c1 = S.c2 = S.c3 = S.del1 = S.del2 = S.del3 = S.clow = S.del1st = S.ceval = 0;
}

// next trial velocity of the bracketing scan (label 1000, surfdisp96.f:448-460); may turn the scan
// around at clow (then c1 is reset to clow, exactly like the reference)
BH_DEV double swd_bracket_next(double &c1, int &idir, double clow, double dc)
{
    double c2;
    for (;;) {
        c2 = (idir > 0) ? c1 + dc : c1 - dc;
        if (c2 <= clow) { idir = +1; c1 = clow; continue; }
        break;
    }
    return c2;
}

// ---- the two common events of a search (bodies of swd_driver's branches) ---------------------------
// A period begins (k <= kmax, k < ift): periods, start value and lower bound of the search,
// surfdisp96.f:231-271.
BH_DEV void swd_on_begin_period(SwdState &S, const SwdTargetDev &tg, const double *BH_RESTRICT per, int wss)
{
    const double TWOPI = 2.0 * 3.141592653589793;
    const double one = 1.0e-2;
    const double onea = 1.5;                        // dble(sone), sone = 1.5 (real*4)
    const double dc = (double)0.005f;               // dabs(dble(ddc)), ddc = 0.005 (real*4)
    const float h = 0.005f;
    double t1 = per[S.k - 1];
    if (tg.igr > 0) {
        S.t1a = (float)(t1 / (double)(1.f + h));
        S.t1b = (float)(t1 / (double)(1.f - h));
        t1 = (double)S.t1a;
    } else {
        S.t1a = (float)t1;
    }
    if (S.k == 1 && S.iq == 1) { S.c1 = S.cc; S.clow = S.cc; S.ifirst = 1; }
    else if (S.k == 1) { S.c1 = S.cws[0] + one * dc; S.clow = S.c1; S.ifirst = 1; }
    else if (S.iq > 1) {
        S.ifirst = 0;
        S.clow = S.cws[(S.k - 1) * wss] + one * dc;
        S.c1 = S.cprev;
        if (S.c1 < S.clow) S.c1 = S.clow;
    } else {
        S.ifirst = 0;
        S.c1 = S.cprev - onea * dc;
        S.clow = S.cc;                        // clow = cm
    }
    S.pass = 0;
    S.omega = TWOPI / t1;
    S.ceval = S.c1;
    S.st = SWD_ST_A;
    S.ev = SWD_EV_NONE;
}

// A root search has ended (SOLVED, or NOROOT on the second solve of a group-velocity pair): the
// second solve is set up, or the period's value is stored and the next period announced,
// surfdisp96.f:273-312.
template <class Src>
BH_DEV void swd_on_solved(SwdState &S, Src &src, const SwdTargetDev &tg, int wss)
{
    const double TWOPI = 2.0 * 3.141592653589793;
    const double one = 1.0e-2;
    const double onea = 1.5;
    const double dc = (double)0.005f;
    const int igr = tg.igr, kmax = tg.nper;
    const bool multimode = tg.mode > 1;
    if (S.pass == 0) {
        S.ck = S.c1;                          // c(k) = c1
        if (multimode) S.cws[(S.k - 1) * wss] = S.ck;
        if (igr > 0) {                        // second solve at t1b, surfdisp96.f:282-294
            double t1 = (double)S.t1b;
            S.ifirst = 0;
            S.clow = (multimode ? S.cbws[(S.k - 1) * wss] : 0.0) + one * dc;
            S.c1 = S.c1 - onea * dc;
            S.pass = 1;
            S.omega = TWOPI / t1;
            S.ceval = S.c1;
            S.st = SWD_ST_A;
            S.ev = SWD_EV_NONE;
            return;
        }
        S.c1 = 0.0;
    } else {
        if (S.ev == SWD_EV_NOROOT) S.c1 = S.ck;   // root not found at the larger period
        if (multimode) S.cbws[(S.k - 1) * wss] = S.c1;
    }
    float cc0 = (float)S.ck, cc1b = (float)S.c1;
    if (igr == 0) {
        src.put(S, S.k, kmax, cc0);
    } else {
        float gvel = (1 / S.t1a - 1 / S.t1b) / (1 / (S.t1a * cc0) - 1 / (S.t1b * cc1b));
        src.put(S, S.k, kmax, gvel);
    }
    S.cprev = S.ck;
    S.k++;
    S.ev = SWD_EV_BEGIN_PERIOD;
}

// ---- driver: task / period / pass / mode bookkeeping (surfdisp96.f:96-355) ------------------------
// Consumes the pending event(s) until the search needs a period-equation value (S.st != DONE, then
// S.omega / S.ceval say where) or the task source is drained (S.st == SWD_ST_DONE).
//   src      task source.  `int next(Lay &lay, double *&out, double *&cws, double *&cbws)` loads
//            the next model of this lane's target into `lay` and returns its layer count (>= 1), or
//            0 when the queue is drained; `void done(int err)` reports the reference's err flag of
//            the task just finished; `void sphere(Lay &, int mmax, int ifunc)` applies swd_sphere to
//            the model just loaded, exactly once (a team shares one copy of the model);
//            `void put(SwdState &, int k, int kmax, float v)` takes the value of period k (1-based),
//            `void fill_zero(SwdState &, int k, int kmax)` zeroes periods k..kmax (swd_put_direct /
//            swd_zero_direct below store straight to S.out; the throughput kernel stages a search's
//            values in LDS and writes the row once).
//   cws/cbws per-task c(k)/cb(k) arrays for mode > 1 (stride `wss` doubles), unused for mode 1
//   allow_fetch  false: return (with S.ev == SWD_EV_FETCH pending) instead of loading the next task --
//            used where the driver runs in the middle of a round (swd_team.h, wide teams)
template <class Lay, class Src>
BH_DEV void swd_driver(SwdState &S, Lay &lay, Src &src, const SwdTargetDev &tg,
                       const double *BH_RESTRICT per, int wss, bool allow_fetch = true)
{
    const double dc = (double)0.005f;               // dabs(dble(ddc)), ddc = 0.005 (real*4)
    const int ifunc = tg.iwave, kmax = tg.nper, nmode = tg.mode;
    const bool multimode = nmode > 1;
    while (S.ev != SWD_EV_NONE) {
        if (S.ev == SWD_EV_FETCH) {                   // surfdisp96.f:96-222 for the next model
            if (!allow_fetch) { S.st = SWD_ST_DONE; break; }
            S.mmax = src.next(lay, S.out, S.cws, S.cbws);
            if (S.mmax <= 0) { S.st = SWD_ST_DONE; S.ev = SWD_EV_NONE; break; }
            S.err = 0;
            S.llw = 1;
            if (lay.b(0) <= 0.0f) S.llw = 2;
            if (tg.iflsph == 1) src.sphere(lay, S.mmax, ifunc);   // once per model (shared by a team)
            int jmn = 0, jsol = 1;                    // extremal velocities, surfdisp96.f:139-156
            S.betmx = -1.e20f;
            float betmn = 1.e20f;
            for (int i = 0; i < S.mmax; i++) {
                float bi = lay.b(i), ai = lay.a(i);
                if (bi > 0.01f && bi < betmn) { betmn = bi; jmn = i; jsol = 1; }
                else if (bi <= 0.01f && ai < betmn) { betmn = ai; jmn = i; jsol = 0; }
                if (bi > S.betmx) S.betmx = bi;
            }
            float cc1 = (jsol == 0) ? betmn : swd_gtsolh(lay.a(jmn), lay.b(jmn));
            cc1 = .95f * cc1;
            cc1 = .90f * cc1;
            S.cc = (double)cc1;                       // cc = c1 = cm
            S.cfail = (double)S.betmx + dc;           // getsol: c1 >= betmx+dc -> no root
            if (multimode)
                for (int i = 0; i < kmax; i++) { S.cws[i * wss] = 0.0; S.cbws[i * wss] = 0.0; }
            S.iq = 1; S.k = 1; S.ift = 999; S.c1 = S.cc;
            if (kmax > 0 && nmode > 0) S.ev = SWD_EV_BEGIN_PERIOD;
            else { src.done(S.err); S.ev = SWD_EV_FETCH; }
        } else if (S.ev == SWD_EV_BEGIN_PERIOD) {
            if (S.k > kmax) {                         // 1600 loop done -> next mode
                S.iq++; S.k = 1;
                if (S.iq > nmode) { src.done(S.err); S.ev = SWD_EV_FETCH; }
                continue;
            }
            if (S.k >= S.ift) { S.ev = SWD_EV_NOROOT; S.pass = 0; continue; }
            swd_on_begin_period(S, tg, per, wss);
        } else if (S.ev == SWD_EV_SOLVED || (S.ev == SWD_EV_NOROOT && S.pass == 1)) {
            swd_on_solved(S, src, tg, wss);
        } else {                                      // NOROOT on the first solve: label 1700
            if (S.iq <= 1) S.err = 1;
            S.ift = S.k;
            src.fill_zero(S, S.k, kmax);                  // cg(k..kmax) = 0, surfdisp96.f:348-354
            S.iq++; S.k = 1;
            if (S.iq > nmode) { src.done(S.err); S.ev = SWD_EV_FETCH; }
            else S.ev = SWD_EV_BEGIN_PERIOD;
        }
    }
}

// The pending event(s) of a search, the common chain first: "root found -> value stored -> next period"
// is two straight-line blocks; only the rare events (task fetch, no root, end of the periods / of a
// mode) enter the driver's loop.  In a wave of the throughput kernel some lane ends a period in 86 %
// of the loop trips, and the loop -- an if-chain per trip, two trips per chain, the whole search state
// copied between loop headers -- was 10.6 % of the kernel's lane-cycles (tools/lane_phase_profile.py).
// Same operations in the same order as swd_driver alone.
template <class Lay, class Src>
BH_DEV void swd_events(SwdState &S, Lay &lay, Src &src, const SwdTargetDev &tg,
                       const double *BH_RESTRICT per, int wss, bool allow_fetch = true)
{
#if !defined(BH_NO_FAST_EVENTS)                   // (A/B switch: the driver's loop for every event)
    if (S.ev == SWD_EV_SOLVED || (S.ev == SWD_EV_NOROOT && S.pass == 1)) swd_on_solved(S, src, tg, wss);
    if (S.ev == SWD_EV_BEGIN_PERIOD && S.k <= tg.nper && S.k < S.ift) swd_on_begin_period(S, tg, per, wss);
    if (S.ev != SWD_EV_NONE)
#endif
        swd_driver(S, lay, src, tg, per, wss, allow_fetch);
}

// ---- control: getsol + nevill as a resumable machine, fed the period-equation value at S.ceval -----
template <class Nev>
BH_DEV void swd_control(SwdState &S, double del, Nev &nv)
{
    const double dc = (double)0.005f;
    const double pct = (double)0.01f;               // `0.01` literal in nevill is real*4
    bool bracket_step = false, finish = false;
    if (S.st == SWD_ST_A) {                           // getsol entry, surfdisp96.f:426-438
        S.del1 = del;
        if (S.ifirst == 1) S.del1st = S.del1;
        // plmn = dsign(1, del1st) * dsign(1, del1); idir = -1 iff plmn < 0
        S.idir = (S.ifirst == 1 || !bh_signs_differ(S.del1st, S.del1)) ? +1 : -1;
        S.nbrk = 0;
        bracket_step = true;
    } else if (S.st == SWD_ST_B) {                    // surfdisp96.f:461-470
        S.del2 = del;
        if (bh_signs_differ(S.del1, S.del2)) {        // bracketed -> nevill: first half (:583)
            S.c3 = 0.5 * (S.c1 + S.c2);
            S.nev = 1; S.nctrl = 1;
            S.ceval = S.c3; S.st = SWD_ST_TOP;
        } else {
            S.c1 = S.c2; S.del1 = S.del2;
            // the reference leaves the scan only through these two bounds; a NaN/Inf model would
            // spin forever there (and hang the GPU here), hence the hard step cap
            if (S.c1 < S.cc || S.c1 >= S.cfail || ++S.nbrk > SWD_MAX_BRACKET_STEPS) S.ev = SWD_EV_NOROOT;
            else bracket_step = true;
        }
    } else {
        bool mid = (S.st == SWD_ST_MID);
        S.del3 = del;
        if (!mid) {                                   // label 100, surfdisp96.f:587-598
            S.nctrl = S.nctrl + 1;
            if (S.nctrl >= 100) finish = true;
            else if (S.c3 < dmin(S.c1, S.c2) || S.c3 > dmax(S.c1, S.c2)) {
                S.nev = 0;
                S.c3 = 0.5 * (S.c1 + S.c2);
                S.ceval = S.c3; S.st = SWD_ST_MID;
            } else mid = true;
        }
        if (mid && !finish) {                         // surfdisp96.f:599-669
            double s13 = S.del1 - S.del3, s32 = S.del3 - S.del2;
            if (bh_signs_differ(S.del3, S.del1)) { S.c2 = S.c3; S.del2 = S.del3; }
            else { S.c1 = S.c3; S.del1 = S.del3; }
            if (fabs(S.c1 - S.c2) <= 1.e-6 * S.c1) finish = true;
            else {
                if (bh_signs_differ(s13, s32)) S.nev = 0;
                double ss1 = fabs(S.del1), s1 = pct * ss1, ss2 = fabs(S.del2), s2 = pct * ss2;
                bool do_half = (s1 > ss2 || s2 > ss1 || S.nev == 0);
                if (!do_half) {
                    double x1;
                    if (swd_neville(nv, S.m, S.nev == 2, S.c1, S.del1, S.c2, S.del2, S.c3, S.del3, &x1)) {
                        S.c3 = x1;
                        S.nev = 2;
                        S.m = S.m + 1;
                        if (S.m > 10) S.m = 10;
                    } else do_half = true;
                }
                if (do_half) {
                    S.c3 = 0.5 * (S.c1 + S.c2);
                    S.nev = 1;
                    S.m = 1;
                }
                S.ceval = S.c3; S.st = SWD_ST_TOP;
            }
        }
    }
    if (bracket_step) {                               // label 1000, surfdisp96.f:448-460
        S.c2 = swd_bracket_next(S.c1, S.idir, S.clow, dc);
        S.ceval = S.c2; S.st = SWD_ST_B;
    }
    if (finish) {                                     // label 1000 of nevill + getsol tail (:475-476)
        S.c1 = S.c3;
        S.ev = (S.c1 > (double)S.betmx) ? SWD_EV_NOROOT : SWD_EV_SOLVED;
    }
}

// Runs (model, target) tasks to completion, one after the other, on this lane: one period-equation
// evaluation per loop trip, at a single call site (throughput kernel; see the header comment).
// On the GPU `src` is a per-target atomic work queue: a lane whose search ends early pulls the next
// model instead of idling until the slowest lane of its wave is done (searches take 388..851
// evaluations on the bench models; static assignment wastes 15 % of the lane-cycles).
// *ncalls (optional) counts period-equation evaluations.
template <class Lay, class Src>
BH_DEV void swd_lane(Lay &lay, Src &src, const SwdTargetDev &tg, const double *BH_RESTRICT per,
                     int wss, long *ncalls)
{
    SwdState S;
    swd_state_init(S);
    NevRegs nv;
    swd_nev_init(nv);
    long nc = 0;
    for (;;) {
        swd_events(S, lay, src, tg, per, wss);
        if (S.st == SWD_ST_DONE) break;
        double wvno = S.omega / S.ceval;
        double del = (tg.iwave == 1) ? swd_dltar1(lay, S.mmax, S.llw, wvno, S.omega)
                                     : swd_dltar4(lay, S.mmax, S.llw, wvno, S.omega);
        nc++;
        swd_control(S, del, nv);
    }
    if (ncalls) *ncalls = nc;
}

}  // namespace bh
