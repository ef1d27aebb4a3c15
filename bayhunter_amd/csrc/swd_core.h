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
        w = sinp / ra;
        x = -ra * sinp;
    } else if (wvno == xka) {
        cosp = 1.0; w = dpth; x = 0.0;
    } else {
        pex = p;
        fac = 0.0;
        if (p < 16) fac = bh_exp(-2.0 * p);
        cosp = (1.0 + fac) * 0.5;
        sinp = (1.0 - fac) * 0.5;
        w = sinp / ra;
        x = ra * sinp;
    }
    if (wvno < xkb) {
        bh_sincos(q, &sinq, &cosq);
        y = sinq / rb;
        z = -rb * sinq;
    } else if (wvno == xkb) {
        cosq = 1.0; y = dpth; z = 0.0;
    } else {
        sex = q;
        fac = 0.0;
        if (q < 16) fac = bh_exp(-2.0 * q);
        cosq = (1.0 + fac) * 0.5;
        sinq = (1.0 - fac) * 0.5;
        y = sinq / rb;
        z = rb * sinq;
    }
    double exa = pex + sex;
    double a0 = 0.0;
    if (exa < 60.0) a0 = bh_exp(-exa);
    o.a0 = a0;
    o.cpcq = cosp * cosq; o.cpy = cosp * y; o.cpz = cosp * z;
    o.cqw = cosq * w;     o.cqx = cosq * x;
    o.xy = x * y; o.xz = x * z; o.wy = w * y; o.wz = w * z;
    *w_out = w;
    *cosp_out = cosp;
}

// One layer of the Dunkin recursion: ee = e * ca (surfdisp96.f:838-844) with ca from dnka
// (:1024-1068), then normc (:995-1020).  ca is never materialised as a 5x5 array: its 25 entries
// are 13 distinct values (ca15..ca55 mirror ca11..), kept in registers.
BH_DEV void swd_dunkin_layer(double e[5], double wvno2, double gam, double gammk, double rho,
                             const VarProd &v)
{
    const double one = 1.0, two = 2.0;
    double gamm1 = gam - one, twgm1 = gam + gamm1, gmgmk = gam * gammk, gmgm1 = gam * gamm1,
           gm1sq = gamm1 * gamm1, rho2 = rho * rho, a0pq = v.a0 - v.cpcq;
    double ca11 = v.cpcq - two * gmgm1 * a0pq - gmgmk * v.xz - wvno2 * gm1sq * v.wy;
    double ca12 = (wvno2 * v.cpy - v.cqx) / rho;
    double ca13 = -(twgm1 * a0pq + gammk * v.xz + wvno2 * gamm1 * v.wy) / rho;
    double ca14 = (v.cpz - wvno2 * v.cqw) / rho;
    double ca15 = -(two * wvno2 * a0pq + v.xz + wvno2 * wvno2 * v.wy) / rho2;
    double ca21 = (gmgmk * v.cpz - gm1sq * v.cqw) * rho;
    double ca22 = v.cpcq;
    double ca23 = gammk * v.cpz - gamm1 * v.cqw;
    double ca24 = -v.wz;
    double ca25 = ca14;
    double ca41 = (gm1sq * v.cpy - gmgmk * v.cqx) * rho;
    double ca42 = -v.xy;
    double ca43 = gamm1 * v.cpy - gammk * v.cqx;
    double ca44 = ca22;
    double ca45 = ca12;
    double ca51 = -(two * gmgmk * gm1sq * a0pq + gmgmk * gmgmk * v.xz + gm1sq * gm1sq * v.wy) * rho2;
    double ca52 = ca41;
    double ca53 = -(gammk * gamm1 * twgm1 * a0pq + gam * gammk * gammk * v.xz + gamm1 * gm1sq * v.wy) * rho;
    double ca54 = ca21;
    double ca55 = ca11;
    double t = -two * wvno2;
    double ca31 = t * ca53;
    double ca32 = t * ca43;
    double ca33 = v.a0 + two * (v.cpcq - ca11);
    double ca34 = t * ca23;
    double ca35 = t * ca13;
    // ee(i) = sum_j e(j)*ca(j,i), accumulated from 0.0 in j order (surfdisp96.f:838-844)
    double e1 = e[0], e2 = e[1], e3 = e[2], e4 = e[3], e5 = e[4];
    double ee1 = ((((0.0 + e1 * ca11) + e2 * ca21) + e3 * ca31) + e4 * ca41) + e5 * ca51;
    double ee2 = ((((0.0 + e1 * ca12) + e2 * ca22) + e3 * ca32) + e4 * ca42) + e5 * ca52;
    double ee3 = ((((0.0 + e1 * ca13) + e2 * ca23) + e3 * ca33) + e4 * ca43) + e5 * ca53;
    double ee4 = ((((0.0 + e1 * ca14) + e2 * ca24) + e3 * ca34) + e4 * ca44) + e5 * ca54;
    double ee5 = ((((0.0 + e1 * ca15) + e2 * ca25) + e3 * ca35) + e4 * ca45) + e5 * ca55;
    // normc: divide by the max-abs (the log of the scale is dead in the reference's caller)
    double t1 = 0.0;
    if (fabs(ee1) > t1) t1 = fabs(ee1);
    if (fabs(ee2) > t1) t1 = fabs(ee2);
    if (fabs(ee3) > t1) t1 = fabs(ee3);
    if (fabs(ee4) > t1) t1 = fabs(ee4);
    if (fabs(ee5) > t1) t1 = fabs(ee5);
    if (t1 < 1.e-40) t1 = 1.0;
    e[0] = ee1 / t1; e[1] = ee2 / t1; e[2] = ee3 / t1; e[3] = ee4 / t1; e[4] = ee5 / t1;
}

// Rayleigh / P-SV period equation, surfdisp96.f:773-871.  Lay: d(i),a(i),b(i),rho(i), 0-based.
template <class Lay>
BH_DEV double swd_dltar4(const Lay &lay, int mmax, int llw, double wvno, double omga)
{
    double e[5];
    double omega = omga;
    if (omega < 1.0e-4) omega = 1.0e-4;
    double wvno2 = wvno * wvno;
    double xka = omega / (double)lay.a(mmax - 1);
    double xkb = omega / (double)lay.b(mmax - 1);
    double wvnop = wvno + xka, wvnom = fabs(wvno - xka);
    double ra = sqrt(wvnop * wvnom);
    wvnop = wvno + xkb; wvnom = fabs(wvno - xkb);
    double rb = sqrt(wvnop * wvnom);
    double t = (double)lay.b(mmax - 1) / omega;
    double gammk = 2.0 * t * t, gam = gammk * wvno2, gamm1 = gam - 1.0;
    double rho1 = (double)lay.rho(mmax - 1);
    e[0] = rho1 * rho1 * (gamm1 * gamm1 - gam * gammk * ra * rb);
    e[1] = -rho1 * ra;
    e[2] = rho1 * (gamm1 - gammk * ra * rb);
    e[3] = rho1 * rb;
    e[4] = wvno2 - ra * rb;
    for (int m = mmax - 1; m >= llw; m--) {  // Fortran index m; 0-based layer m-1
        int i0 = m - 1;
        double am = (double)lay.a(i0), bm = (double)lay.b(i0);
        xka = omega / am;
        xkb = omega / bm;
        t = bm / omega;
        gammk = 2.0 * t * t;
        gam = gammk * wvno2;
        wvnop = wvno + xka; wvnom = fabs(wvno - xka);
        ra = sqrt(wvnop * wvnom);
        wvnop = wvno + xkb; wvnom = fabs(wvno - xkb);
        rb = sqrt(wvnop * wvnom);
        double dpth = (double)lay.d(i0);
        rho1 = (double)lay.rho(i0);
        double p = ra * dpth, q = rb * dpth, w, cosp;
        VarProd v;
        swd_var(p, q, ra, rb, wvno, xka, xkb, dpth, &w, &cosp, v);
        swd_dunkin_layer(e, wvno2, gam, gammk, rho1, v);
    }
    if (llw != 1) {  // water layer on top, surfdisp96.f:850-867
        xka = omega / (double)lay.a(0);
        wvnop = wvno + xka; wvnom = fabs(wvno - xka);
        ra = sqrt(wvnop * wvnom);
        double dpth = (double)lay.d(0);
        rho1 = (double)lay.rho(0);
        double p = ra * dpth, znul = 1.0e-05, w, cosp;
        VarProd v;
        swd_var(p, znul, ra, znul, wvno, xka, znul, dpth, &w, &cosp, v);
        double w0 = -rho1 * w;
        return cosp * e[0] + w0 * e[1];
    }
    return e[0];
}

// Love / SH period equation, surfdisp96.f:710-769.
template <class Lay>
BH_DEV double swd_dltar1(const Lay &lay, int mmax, int llw, double wvno, double omega)
{
    double beta1 = (double)lay.b(mmax - 1), rho1 = (double)lay.rho(mmax - 1);
    double xkb = omega / beta1;
    double wvnop = wvno + xkb, wvnom = fabs(wvno - xkb);
    double rb = sqrt(wvnop * wvnom);
    double e1 = rho1 * rb, e2 = 1.0 / (beta1 * beta1);
    for (int m = mmax - 1; m >= llw; m--) {
        int i0 = m - 1;
        beta1 = (double)lay.b(i0);
        rho1 = (double)lay.rho(i0);
        double dm = (double)lay.d(i0);
        double xmu = rho1 * beta1 * beta1;
        xkb = omega / beta1;
        wvnop = wvno + xkb; wvnom = fabs(wvno - xkb);
        rb = sqrt(wvnop * wvnom);
        double q = dm * rb, y, z, cosq, sinq, fac;
        if (wvno < xkb) {
            bh_sincos(q, &sinq, &cosq);
            y = sinq / rb;
            z = -rb * sinq;
        } else if (wvno == xkb) {
            cosq = 1.0; y = dm; z = 0.0;
        } else {
            fac = 0.0;
            if (q < 16) fac = bh_exp(-2.0 * q);
            cosq = (1.0 + fac) * 0.5;
            sinq = (1.0 - fac) * 0.5;
            y = sinq / rb;
            z = rb * sinq;
        }
        double e10 = e1 * cosq + e2 * xmu * z;
        double e20 = e1 * y / xmu + e2 * cosq;
        double xnor = fabs(e10), ynor = fabs(e20);
        if (ynor > xnor) xnor = ynor;
        if (xnor < 1.e-40) xnor = 1.0;
        e1 = e10 / xnor;
        e2 = e20 / xnor;
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

// sphere, surfdisp96.f:486-553, both calls (iflag 0 then iflag 1 for this lane's wave type) folded
// into one pass: d,a,b are transformed, rho is scaled by btp**(-5) (Love) / btp**(-2.275) (Rayleigh).
template <class Lay>
BH_DEV void swd_sphere(Lay &lay, int mmax, int ifunc)
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
            lay.set_rho(i, rtp * powf(btp, -2.275f));
        }
        r0 = r1;
    }
    lay.set_d(mmax - 1, 0.0f);
}

// ---- the search ----------------------------------------------------------------------------------
struct SwdTargetDev {
    int iwave, igr, mode, iflsph, nper, per_off, out_off, _pad;
};

enum { SWD_MAX_BRACKET_STEPS = 100000 };
enum { SWD_ST_A = 0, SWD_ST_B = 1, SWD_ST_TOP = 2, SWD_ST_MID = 3, SWD_ST_DONE = 4 };

// Runs (model, target) tasks to completion, one after the other, on this lane.
//
//   src      task source.  `int next(Lay &lay, double *&out, double *&cws, double *&cbws)` loads
//            the next model of this lane's target into `lay` and returns its layer count (>= 1), or
//            0 when the queue is drained; `void done(int err)` reports the reference's err flag of
//            the task just finished.  On the GPU this is a per-target atomic work queue: a lane
//            whose search ends early pulls the next model instead of idling until the slowest lane
//            of its wave is done (searches take 388..851 period-equation evaluations on the bench
//            models; static assignment wastes 15 % of the lane-cycles).
//   tg       the dispersion target (same for every task of this lane), per = its periods
//   cws/cbws per-task c(k)/cb(k) arrays for mode > 1 (stride `wss` doubles), unused for mode 1
// *ncalls (optional) counts period-equation evaluations.
template <class Lay, class Src>
BH_DEV void swd_lane(Lay &lay, Src &src, const SwdTargetDev &tg, const double *BH_RESTRICT per,
                     int wss, long *ncalls)
{
    const double TWOPI = 2.0 * 3.141592653589793;
    const double one = 1.0e-2;
    const double onea = 1.5;                        // dble(sone), sone = 1.5 (real*4)
    const double dc = (double)0.005f;               // dabs(dble(ddc)), ddc = 0.005 (real*4)
    const float h = 0.005f;
    const double pct = (double)0.01f;               // `0.01` literal in nevill is real*4
    const int ifunc = tg.iwave, igr = tg.igr, kmax = tg.nper, nmode = tg.mode;
    const bool multimode = nmode > 1;
    long nc = 0;

    // per-task constants
    int mmax = 1, llw = 1, err = 0;
    float betmx = 0.f;
    double cc = 0, cfail = 0;
    double *out = nullptr, *cws = nullptr, *cbws = nullptr;

    // search state
    int iq = 1, k = 1, ift = 999, pass = 0, st = SWD_ST_DONE, ifirst = 0, idir = 1;
    int nev = 1, nctrl = 1, m = 1, nbrk = 0;
    double t1 = 0, omega = 0, c1 = 0, c2 = 0, c3 = 0, del1 = 0, del2 = 0, del3 = 0, clow = 0,
           del1st = 0, cprev = 0, ck = 0, ceval = 0;
    float t1a = 0, t1b = 0;
    double x1 = 0, x2 = 0, x3 = 0, x4 = 0, x5 = 0, x6 = 0, x7 = 0, x8 = 0, x9 = 0, x10 = 0, x11 = 0;
    double y1 = 0, y2 = 0, y3 = 0, y4 = 0, y5 = 0, y6 = 0, y7 = 0, y8 = 0, y9 = 0, y10 = 0, y11 = 0;

    // control events raised by the search, consumed by the task/period/mode driver below
    enum { EV_NONE = 0, EV_FETCH, EV_BEGIN_PERIOD, EV_SOLVED, EV_NOROOT };
    int ev = EV_FETCH;

    for (;;) {
        // ---------------- driver: task / period / pass / mode bookkeeping -----------------------
        while (ev != EV_NONE) {
            if (ev == EV_FETCH) {                     // surfdisp96.f:96-222 for the next model
                mmax = src.next(lay, out, cws, cbws);
                if (mmax <= 0) { st = SWD_ST_DONE; ev = EV_NONE; break; }
                err = 0;
                llw = 1;
                if (lay.b(0) <= 0.0f) llw = 2;
                if (tg.iflsph == 1) swd_sphere(lay, mmax, ifunc);
                int jmn = 0, jsol = 1;                // extremal velocities, surfdisp96.f:139-156
                betmx = -1.e20f;
                float betmn = 1.e20f;
                for (int i = 0; i < mmax; i++) {
                    float bi = lay.b(i), ai = lay.a(i);
                    if (bi > 0.01f && bi < betmn) { betmn = bi; jmn = i; jsol = 1; }
                    else if (bi <= 0.01f && ai < betmn) { betmn = ai; jmn = i; jsol = 0; }
                    if (bi > betmx) betmx = bi;
                }
                float cc1 = (jsol == 0) ? betmn : swd_gtsolh(lay.a(jmn), lay.b(jmn));
                cc1 = .95f * cc1;
                cc1 = .90f * cc1;
                cc = (double)cc1;                     // cc = c1 = cm
                cfail = (double)betmx + dc;           // getsol: c1 >= betmx+dc -> no root
                if (multimode)
                    for (int i = 0; i < kmax; i++) { cws[i * wss] = 0.0; cbws[i * wss] = 0.0; }
                iq = 1; k = 1; ift = 999; c1 = cc;
                if (kmax > 0 && nmode > 0) ev = EV_BEGIN_PERIOD;
                else { src.done(err); ev = EV_FETCH; }
            } else if (ev == EV_BEGIN_PERIOD) {
                if (k > kmax) {                       // 1600 loop done -> next mode
                    iq++; k = 1;
                    if (iq > nmode) { src.done(err); ev = EV_FETCH; }
                    continue;
                }
                if (k >= ift) { ev = EV_NOROOT; pass = 0; continue; }
                t1 = per[k - 1];
                if (igr > 0) {
                    t1a = (float)(t1 / (double)(1.f + h));
                    t1b = (float)(t1 / (double)(1.f - h));
                    t1 = (double)t1a;
                } else {
                    t1a = (float)t1;
                }
                if (k == 1 && iq == 1) { c1 = cc; clow = cc; ifirst = 1; }
                else if (k == 1) { c1 = cws[0] + one * dc; clow = c1; ifirst = 1; }
                else if (iq > 1) {
                    ifirst = 0;
                    clow = cws[(k - 1) * wss] + one * dc;
                    c1 = cprev;
                    if (c1 < clow) c1 = clow;
                } else {
                    ifirst = 0;
                    c1 = cprev - onea * dc;
                    clow = cc;                        // clow = cm
                }
                pass = 0;
                omega = TWOPI / t1;
                ceval = c1;
                st = SWD_ST_A;
                ev = EV_NONE;
            } else if (ev == EV_SOLVED || (ev == EV_NOROOT && pass == 1)) {
                if (pass == 0) {
                    ck = c1;                          // c(k) = c1
                    if (multimode) cws[(k - 1) * wss] = ck;
                    if (igr > 0) {                    // second solve at t1b, surfdisp96.f:282-294
                        t1 = (double)t1b;
                        ifirst = 0;
                        clow = (multimode ? cbws[(k - 1) * wss] : 0.0) + one * dc;
                        c1 = c1 - onea * dc;
                        pass = 1;
                        omega = TWOPI / t1;
                        ceval = c1;
                        st = SWD_ST_A;
                        ev = EV_NONE;
                        continue;
                    }
                    c1 = 0.0;
                } else {
                    if (ev == EV_NOROOT) c1 = ck;     // root not found at the larger period
                    if (multimode) cbws[(k - 1) * wss] = c1;
                }
                float cc0 = (float)ck, cc1b = (float)c1;
                if (igr == 0) {
                    out[k - 1] = (double)cc0;
                } else {
                    float gvel = (1 / t1a - 1 / t1b) / (1 / (t1a * cc0) - 1 / (t1b * cc1b));
                    out[k - 1] = (double)gvel;
                }
                cprev = ck;
                k++;
                ev = EV_BEGIN_PERIOD;
            } else {                                  // EV_NOROOT on the first solve: label 1700
                if (iq <= 1) err = 1;
                ift = k;
                for (int i = k; i <= kmax; i++) out[i - 1] = 0.0;
                iq++; k = 1;
                if (iq > nmode) { src.done(err); ev = EV_FETCH; }
                else ev = EV_BEGIN_PERIOD;
            }
        }
        if (st == SWD_ST_DONE) break;

        // ---------------- the one period-equation evaluation per trip --------------------------
        double wvno = omega / ceval;
        double del = (ifunc == 1) ? swd_dltar1(lay, mmax, llw, wvno, omega)
                                  : swd_dltar4(lay, mmax, llw, wvno, omega);
        nc++;

        // ---------------- search control: getsol + nevill as a resumable machine ----------------
        bool bracket_step = false, finish = false;
        if (st == SWD_ST_A) {                         // getsol entry, surfdisp96.f:426-438
            del1 = del;
            if (ifirst == 1) del1st = del1;
            double plmn = dsign1(del1st) * dsign1(del1);
            idir = (ifirst == 1 || plmn >= 0.0) ? +1 : -1;
            nbrk = 0;
            bracket_step = true;
        } else if (st == SWD_ST_B) {                  // surfdisp96.f:461-470
            del2 = del;
            if (dsign1(del1) != dsign1(del2)) {       // bracketed -> nevill: first half (:583)
                c3 = 0.5 * (c1 + c2);
                nev = 1; nctrl = 1;
                ceval = c3; st = SWD_ST_TOP;
            } else {
                c1 = c2; del1 = del2;
                // the reference leaves the scan only through these two bounds; a NaN/Inf model would
                // spin forever there (and hang the GPU here), hence the hard step cap
                if (c1 < cc || c1 >= cfail || ++nbrk > SWD_MAX_BRACKET_STEPS) ev = EV_NOROOT;
                else bracket_step = true;
            }
        } else {
            bool mid = (st == SWD_ST_MID);
            del3 = del;
            if (!mid) {                               // label 100, surfdisp96.f:587-598
                nctrl = nctrl + 1;
                if (nctrl >= 100) finish = true;
                else if (c3 < dmin(c1, c2) || c3 > dmax(c1, c2)) {
                    nev = 0;
                    c3 = 0.5 * (c1 + c2);
                    ceval = c3; st = SWD_ST_MID;
                } else mid = true;
            }
            if (mid && !finish) {                     // surfdisp96.f:599-669
                double s13 = del1 - del3, s32 = del3 - del2;
                if (dsign1(del3) * dsign1(del1) < 0.0) { c2 = c3; del2 = del3; }
                else { c1 = c3; del1 = del3; }
                if (fabs(c1 - c2) <= 1.e-6 * c1) finish = true;
                else {
                    if (dsign1(s13) != dsign1(s32)) nev = 0;
                    double ss1 = fabs(del1), s1 = pct * ss1, ss2 = fabs(del2), s2 = pct * ss2;
                    bool do_half = (s1 > ss2 || s2 > ss1 || nev == 0);
                    if (!do_half) {
                        double ym1;                   // y(m+1)
                        if (nev == 2) {               // x(m+1)=c3, y(m+1)=del3
                            ym1 = del3;
                            if (m == 1) { x2 = c3; y2 = del3; } else if (m == 2) { x3 = c3; y3 = del3; }
                            else if (m == 3) { x4 = c3; y4 = del3; } else if (m == 4) { x5 = c3; y5 = del3; }
                            else if (m == 5) { x6 = c3; y6 = del3; } else if (m == 6) { x7 = c3; y7 = del3; }
                            else if (m == 7) { x8 = c3; y8 = del3; } else if (m == 8) { x9 = c3; y9 = del3; }
                            else if (m == 9) { x10 = c3; y10 = del3; } else { x11 = c3; y11 = del3; }
                        } else {
                            x1 = c1; y1 = del1; x2 = c2; y2 = del2; m = 1;
                            ym1 = del2;
                        }
                        // Neville inverse interpolation, j = m .. 1 (surfdisp96.f:649-654)
                        bool bad = false;
                        const double guard = 1.0e-10 * fabs(ym1);
#define BH_NEV_STEP(J, XJ, YJ, XJ1)                                               \
    if (!bad && m >= J) {                                                         \
        double denom = ym1 - YJ;                                                  \
        if (fabs(denom) < guard) bad = true;                                      \
        else XJ = (-YJ * XJ1 + ym1 * XJ) / denom;                                 \
    }
                        BH_NEV_STEP(10, x10, y10, x11)
                        BH_NEV_STEP(9, x9, y9, x10)
                        BH_NEV_STEP(8, x8, y8, x9)
                        BH_NEV_STEP(7, x7, y7, x8)
                        BH_NEV_STEP(6, x6, y6, x7)
                        BH_NEV_STEP(5, x5, y5, x6)
                        BH_NEV_STEP(4, x4, y4, x5)
                        BH_NEV_STEP(3, x3, y3, x4)
                        BH_NEV_STEP(2, x2, y2, x3)
                        BH_NEV_STEP(1, x1, y1, x2)
#undef BH_NEV_STEP
                        if (!bad) {
                            c3 = x1;
                            nev = 2;
                            m = m + 1;
                            if (m > 10) m = 10;
                        } else do_half = true;
                    }
                    if (do_half) {
                        c3 = 0.5 * (c1 + c2);
                        nev = 1;
                        m = 1;
                    }
                    ceval = c3; st = SWD_ST_TOP;
                }
            }
        }
        if (bracket_step) {                           // label 1000, surfdisp96.f:448-460
            for (;;) {
                c2 = (idir > 0) ? c1 + dc : c1 - dc;
                if (c2 <= clow) { idir = +1; c1 = clow; continue; }
                break;
            }
            ceval = c2; st = SWD_ST_B;
        }
        if (finish) {                                 // label 1000 of nevill + getsol tail (:475-476)
            c1 = c3;
            ev = (c1 > (double)betmx) ? EV_NOROOT : EV_SOLVED;
        }
    }
    (void)y11;
    if (ncalls) *ncalls = nc;
}

}  // namespace bh
