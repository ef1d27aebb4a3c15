/*
 * oracle_swd.c -- CPU restatement of the reference's SURF96 dispersion solver.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle_port.h).  Follows /root/reference/src/extensions/
 * surfdisp96.f routine by routine; every function names the lines it restates.  The mixed
 * real*4 / real*8 choreography of the Fortran (implicit typing!) is replayed exactly so that the
 * output is bit-identical to the flang/gfortran-compiled reference on x86-64 without FMA
 * (checked by tests/test_oracle.py::test_live_against_reference against oracle/_ref/libsurfdisp96_ref.so).
 *
 * Pinning: bit-identical to oracle/_ref on the seeded model sets of tests/golden/make_golden.py;
 * reproduces tutorial/observed/st3_{r,l}disp{ph,gr}.dat to the files' 4-decimal rounding.
 */
#include <math.h>
#include <string.h>
#include <omp.h>
#include "oracle_port.h"

typedef struct {
    int mmax, llw;
    float d[BHO_NL], a[BHO_NL], b[BHO_NL], rho[BHO_NL];
    float rtp[BHO_NL], dtp[BHO_NL], btp[BHO_NL];
    float dhalf;    /* SAVE dhalf   (surfdisp96.f:515) */
    double del1st;  /* SAVE del1st  (surfdisp96.f:415) */
    long ncalls;    /* period-equation evaluations */
} swd_ctx;

static const double TWOPI = 2.0 * 3.141592653589793; /* surfdisp96.f:136 */

static inline double dsign1(double x) { return copysign(1.0, x); }

/* ---- normc (surfdisp96.f:995-1020); the returned exponent is unused by the caller ---- */
static void normc(double ee[5])
{
    double t1 = 0.0;
    for (int i = 0; i < 5; i++)
        if (fabs(ee[i]) > t1) t1 = fabs(ee[i]);
    if (t1 < 1.e-40) t1 = 1.0;
    for (int i = 0; i < 5; i++) ee[i] = ee[i] / t1;
}

/* ---- var (surfdisp96.f:874-991) ---- */
typedef struct { double a0, cpcq, cpy, cpz, cqw, cqx, xy, xz, wy, wz; } varprod;

static void var(double p, double q, double ra, double rb, double wvno, double xka, double xkb,
                double dpth, double *w_out, double *cosp_out, varprod *o)
{
    double w = 0, x = 0, y = 0, z = 0, cosp = 0, cosq = 0, sinp, sinq, fac;
    double pex = 0.0, sex = 0.0, exa, a0;
    if (wvno < xka) {
        sinp = sin(p); w = sinp / ra; x = -ra * sinp; cosp = cos(p);
    } else if (wvno == xka) {
        cosp = 1.0; w = dpth; x = 0.0;
    } else {
        pex = p; fac = 0.0;
        if (p < 16) fac = exp(-2.0 * p);
        cosp = (1.0 + fac) * 0.5; sinp = (1.0 - fac) * 0.5;
        w = sinp / ra; x = ra * sinp;
    }
    if (wvno < xkb) {
        sinq = sin(q); y = sinq / rb; z = -rb * sinq; cosq = cos(q);
    } else if (wvno == xkb) {
        cosq = 1.0; y = dpth; z = 0.0;
    } else {
        sex = q; fac = 0.0;
        if (q < 16) fac = exp(-2.0 * q);
        cosq = (1.0 + fac) * 0.5; sinq = (1.0 - fac) * 0.5;
        y = sinq / rb; z = rb * sinq;
    }
    exa = pex + sex;
    a0 = 0.0;
    if (exa < 60.0) a0 = exp(-exa);
    o->a0 = a0;
    o->cpcq = cosp * cosq; o->cpy = cosp * y; o->cpz = cosp * z;
    o->cqw = cosq * w;     o->cqx = cosq * x;
    o->xy = x * y; o->xz = x * z; o->wy = w * y; o->wz = w * z;
    /* lines 984-989 rescale cosq,y,z locally: results are discarded by dltar4 */
    *w_out = w; *cosp_out = cosp;
}

/* ---- dnka (surfdisp96.f:1024-1068); ca is ca[row][col], 0-based ---- */
static void dnka(double ca[5][5], double wvno2, double gam, double gammk, double rho,
                 const varprod *v)
{
    const double one = 1.0, two = 2.0;
    double a0 = v->a0, cpcq = v->cpcq, cpy = v->cpy, cpz = v->cpz, cqw = v->cqw, cqx = v->cqx,
           xy = v->xy, xz = v->xz, wy = v->wy, wz = v->wz;
    double gamm1 = gam - one, twgm1 = gam + gamm1, gmgmk = gam * gammk, gmgm1 = gam * gamm1,
           gm1sq = gamm1 * gamm1, rho2 = rho * rho, a0pq = a0 - cpcq, t;
    ca[0][0] = cpcq - two * gmgm1 * a0pq - gmgmk * xz - wvno2 * gm1sq * wy;
    ca[0][1] = (wvno2 * cpy - cqx) / rho;
    ca[0][2] = -(twgm1 * a0pq + gammk * xz + wvno2 * gamm1 * wy) / rho;
    ca[0][3] = (cpz - wvno2 * cqw) / rho;
    ca[0][4] = -(two * wvno2 * a0pq + xz + wvno2 * wvno2 * wy) / rho2;
    ca[1][0] = (gmgmk * cpz - gm1sq * cqw) * rho;
    ca[1][1] = cpcq;
    ca[1][2] = gammk * cpz - gamm1 * cqw;
    ca[1][3] = -wz;
    ca[1][4] = ca[0][3];
    ca[3][0] = (gm1sq * cpy - gmgmk * cqx) * rho;
    ca[3][1] = -xy;
    ca[3][2] = gamm1 * cpy - gammk * cqx;
    ca[3][3] = ca[1][1];
    ca[3][4] = ca[0][1];
    ca[4][0] = -(two * gmgmk * gm1sq * a0pq + gmgmk * gmgmk * xz + gm1sq * gm1sq * wy) * rho2;
    ca[4][1] = ca[3][0];
    ca[4][2] = -(gammk * gamm1 * twgm1 * a0pq + gam * gammk * gammk * xz + gamm1 * gm1sq * wy) * rho;
    ca[4][3] = ca[1][0];
    ca[4][4] = ca[0][0];
    t = -two * wvno2;
    ca[2][0] = t * ca[4][2];
    ca[2][1] = t * ca[3][2];
    ca[2][2] = a0 + two * (cpcq - ca[0][0]);
    ca[2][3] = t * ca[1][2];
    ca[2][4] = t * ca[0][2];
}

/* ---- dltar4: Rayleigh / P-SV period equation (surfdisp96.f:773-871) ---- */
static double dltar4(const swd_ctx *c, double wvno, double omga)
{
    double e[5], ee[5], ca[5][5];
    double omega = omga;
    int mmax = c->mmax;
    if (omega < 1.0e-4) omega = 1.0e-4;
    double wvno2 = wvno * wvno;
    double xka = omega / (double)c->a[mmax - 1];
    double xkb = omega / (double)c->b[mmax - 1];
    double wvnop = wvno + xka, wvnom = fabs(wvno - xka);
    double ra = sqrt(wvnop * wvnom);
    wvnop = wvno + xkb; wvnom = fabs(wvno - xkb);
    double rb = sqrt(wvnop * wvnom);
    double t = (double)c->b[mmax - 1] / omega;
    double gammk = 2.0 * t * t, gam = gammk * wvno2, gamm1 = gam - 1.0;
    double rho1 = (double)c->rho[mmax - 1];
    e[0] = rho1 * rho1 * (gamm1 * gamm1 - gam * gammk * ra * rb);
    e[1] = -rho1 * ra;
    e[2] = rho1 * (gamm1 - gammk * ra * rb);
    e[3] = rho1 * rb;
    e[4] = wvno2 - ra * rb;
    for (int m = mmax - 1; m >= c->llw; m--) { /* Fortran m = mmax-1 .. llw (1-based) */
        int i0 = m - 1;
        xka = omega / (double)c->a[i0];
        xkb = omega / (double)c->b[i0];
        t = (double)c->b[i0] / omega;
        gammk = 2.0 * t * t;
        gam = gammk * wvno2;
        wvnop = wvno + xka; wvnom = fabs(wvno - xka);
        ra = sqrt(wvnop * wvnom);
        wvnop = wvno + xkb; wvnom = fabs(wvno - xkb);
        rb = sqrt(wvnop * wvnom);
        double dpth = (double)c->d[i0];
        rho1 = (double)c->rho[i0];
        double p = ra * dpth, q = rb * dpth, w, cosp;
        varprod v;
        var(p, q, ra, rb, wvno, xka, xkb, dpth, &w, &cosp, &v);
        dnka(ca, wvno2, gam, gammk, rho1, &v);
        for (int i = 0; i < 5; i++) {
            double cr = 0.0;
            for (int j = 0; j < 5; j++) cr = cr + e[j] * ca[j][i];
            ee[i] = cr;
        }
        normc(ee);
        for (int i = 0; i < 5; i++) e[i] = ee[i];
    }
    if (c->llw != 1) { /* water layer on top (surfdisp96.f:850-867) */
        xka = omega / (double)c->a[0];
        wvnop = wvno + xka; wvnom = fabs(wvno - xka);
        ra = sqrt(wvnop * wvnom);
        double dpth = (double)c->d[0];
        rho1 = (double)c->rho[0];
        double p = ra * dpth, znul = 1.0e-05, w, cosp;
        varprod v;
        var(p, znul, ra, znul, wvno, xka, znul, dpth, &w, &cosp, &v);
        double w0 = -rho1 * w;
        return cosp * e[0] + w0 * e[1];
    }
    return e[0];
}

/* ---- dltar1: Love / SH period equation (surfdisp96.f:710-769) ---- */
static double dltar1(const swd_ctx *c, double wvno, double omega)
{
    int mmax = c->mmax;
    double beta1 = (double)c->b[mmax - 1], rho1 = (double)c->rho[mmax - 1];
    double xkb = omega / beta1;
    double wvnop = wvno + xkb, wvnom = fabs(wvno - xkb);
    double rb = sqrt(wvnop * wvnom);
    double e1 = rho1 * rb, e2 = 1.0 / (beta1 * beta1);
    for (int m = mmax - 1; m >= c->llw; m--) {
        int i0 = m - 1;
        beta1 = (double)c->b[i0];
        rho1 = (double)c->rho[i0];
        double xmu = rho1 * beta1 * beta1;
        xkb = omega / beta1;
        wvnop = wvno + xkb; wvnom = fabs(wvno - xkb);
        rb = sqrt(wvnop * wvnom);
        double q = (double)c->d[i0] * rb, y, z, cosq, sinq, fac;
        if (wvno < xkb) {
            sinq = sin(q); y = sinq / rb; z = -rb * sinq; cosq = cos(q);
        } else if (wvno == xkb) {
            cosq = 1.0; y = (double)c->d[i0]; z = 0.0;
        } else {
            fac = 0.0;
            if (q < 16) fac = exp(-2.0 * q);
            cosq = (1.0 + fac) * 0.5; sinq = (1.0 - fac) * 0.5;
            y = sinq / rb; z = rb * sinq;
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

/* ---- dltar (surfdisp96.f:690-706) ---- */
static double dltar(swd_ctx *c, double wvno, double omega, int ifunc)
{
    c->ncalls++;
    return ifunc == 1 ? dltar1(c, wvno, omega) : dltar4(c, wvno, omega);
}

/* ---- half (surfdisp96.f:676-686) ---- */
static void half(swd_ctx *c, double c1, double c2, double *c3, double *del3, double omega, int ifunc)
{
    *c3 = 0.5 * (c1 + c2);
    double wvno = omega / *c3;
    *del3 = dltar(c, wvno, omega, ifunc);
}

/* ---- nevill (surfdisp96.f:557-674) ---- */
static double nevill(swd_ctx *c, double t, double c1, double c2, double del1, double del2, int ifunc)
{
    double x[21], y[21]; /* 1-based, dimension x(20),y(20) */
    double omega = TWOPI / t, c3, del3;
    const double pct = (double)0.01f; /* `0.01` is a real*4 literal (surfdisp96.f:625,627) */
    int nev, nctrl, m = 1;
    half(c, c1, c2, &c3, &del3, omega, ifunc);
    nev = 1;
    nctrl = 1;
    for (;;) {
        nctrl = nctrl + 1;
        if (nctrl >= 100) break;
        if (c3 < fmin(c1, c2) || c3 > fmax(c1, c2)) {
            nev = 0;
            half(c, c1, c2, &c3, &del3, omega, ifunc);
        }
        double s13 = del1 - del3, s32 = del3 - del2;
        if (dsign1(del3) * dsign1(del1) < 0.0) { c2 = c3; del2 = del3; }
        else                                   { c1 = c3; del1 = del3; }
        if (fabs(c1 - c2) <= 1.e-6 * c1) break;
        if (dsign1(s13) != dsign1(s32)) nev = 0;
        double ss1 = fabs(del1), s1 = pct * ss1, ss2 = fabs(del2), s2 = pct * ss2;
        if (s1 > ss2 || s2 > ss1 || nev == 0) {
            half(c, c1, c2, &c3, &del3, omega, ifunc);
            nev = 1;
            m = 1;
        } else {
            if (nev == 2) {
                x[m + 1] = c3; y[m + 1] = del3;
            } else {
                x[1] = c1; y[1] = del1; x[2] = c2; y[2] = del2; m = 1;
            }
            int bad = 0;
            for (int kk = 1; kk <= m; kk++) {
                int j = m - kk + 1;
                double denom = y[m + 1] - y[j];
                if (fabs(denom) < 1.0e-10 * fabs(y[m + 1])) { bad = 1; break; }
                x[j] = (-y[j] * x[j + 1] + y[m + 1] * x[j]) / denom;
            }
            if (!bad) {
                c3 = x[1];
                double wvno = omega / c3;
                del3 = dltar(c, wvno, omega, ifunc);
                nev = 2;
                m = m + 1;
                if (m > 10) m = 10;
            } else {
                half(c, c1, c2, &c3, &del3, omega, ifunc);
                nev = 1;
                m = 1;
            }
        }
    }
    return c3;
}

/* ---- getsol (surfdisp96.f:390-482); c1 is in/out like the Fortran dummy ---- */
static int getsol(swd_ctx *c, double t1, double *c1io, double clow, double dc, double cm,
                  float betmx, int ifunc, int ifirst)
{
    double c1 = *c1io, c2, del1, del2, omega, wvno, plmn;
    int idir;
    omega = TWOPI / t1;
    wvno = omega / c1;
    del1 = dltar(c, wvno, omega, ifunc);
    if (ifirst == 1) c->del1st = del1;
    plmn = dsign1(c->del1st) * dsign1(del1);
    if (ifirst == 1) idir = +1;
    else if (plmn >= 0.0) idir = +1;
    else idir = -1;
    for (;;) {
        if (idir > 0) c2 = c1 + dc; else c2 = c1 - dc;
        if (c2 <= clow) { idir = +1; c1 = clow; continue; }
        omega = TWOPI / t1;
        wvno = omega / c2;
        del2 = dltar(c, wvno, omega, ifunc);
        if (dsign1(del1) != dsign1(del2)) break;
        c1 = c2;
        del1 = del2;
        if (c1 < cm) { *c1io = c1; return -1; }
        if (c1 >= ((double)betmx + dc)) { *c1io = c1; return -1; }
    }
    c1 = nevill(c, t1, c1, c2, del1, del2, ifunc);
    *c1io = c1;
    if (c1 > (double)betmx) return -1;
    return 1;
}

/* ---- gtsolh (surfdisp96.f:367-388): all real*4 ---- */
static float gtsolh(float a, float b)
{
    float c = 0.95f * b;
    for (int i = 0; i < 5; i++) {
        float gamma = b / a, kappa = c / b;
        float k2 = kappa * kappa;
        float gk = gamma * kappa, gk2 = gk * gk;
        float fac1 = sqrtf(1.0f - gk2), fac2 = sqrtf(1.0f - k2);
        float tk = 2.0f - k2;
        float fr = tk * tk - 4.0f * fac1 * fac2;
        float frp = -4.0f * (2.0f - k2) * kappa + 4.0f * fac2 * gamma * gamma * kappa / fac1
                    + 4.0f * fac1 * kappa / fac2;
        frp = frp / b;
        c = c - fr / frp;
    }
    return c;
}

/* ---- sphere (surfdisp96.f:486-553) ---- */
static void sphere(swd_ctx *c, int ifunc, int iflag)
{
    double z0, z1, r0, r1, dr, ar, tmp;
    int mmax = c->mmax;
    ar = 6370.0; dr = 0.0; r0 = ar;
    c->d[mmax - 1] = 1.0f;
    if (iflag == 0) {
        for (int i = 0; i < mmax; i++) { c->dtp[i] = c->d[i]; c->rtp[i] = c->rho[i]; }
        for (int i = 0; i < mmax; i++) {
            dr = dr + (double)c->d[i];
            r1 = ar - dr;
            z0 = ar * log(ar / r0);
            z1 = ar * log(ar / r1);
            c->d[i] = (float)(z1 - z0);
            tmp = (ar + ar) / (r0 + r1);
            c->a[i] = (float)((double)c->a[i] * tmp);
            c->b[i] = (float)((double)c->b[i] * tmp);
            c->btp[i] = (float)tmp;
            r0 = r1;
        }
        c->dhalf = c->d[mmax - 1];
    } else {
        c->d[mmax - 1] = c->dhalf;
        for (int i = 0; i < mmax; i++) {
            if (ifunc == 1) { /* btp**(-5): integer power = 1/(((x*x)*(x*x))*x), as flang lowers it */
                float x = c->btp[i], x2 = x * x, x4 = x2 * x2, x5 = x4 * x;
                c->rho[i] = c->rtp[i] * (1.0f / x5);
            }
            else if (ifunc == 2) c->rho[i] = c->rtp[i] * powf(c->btp[i], -2.275f); /* btp**(-2.275) */
        }
    }
    c->d[mmax - 1] = 0.0f;
}

/* ---- surfdisp96 (surfdisp96.f:55-360) ---- */
int bho_surfdisp96(const float *thkm, const float *vpm, const float *vsm, const float *rhom,
                   int nlayer, int iflsph, int iwave, int mode, int igr, int kmax,
                   const double *t, double *cg, long *n_dltar)
{
    swd_ctx ctx;
    swd_ctx *c = &ctx;
    double cvel[BHO_NP], cb[BHO_NP];
    int err = 0, mmax = nlayer, nsph = iflsph;
    memset(c, 0, sizeof(*c));
    c->mmax = mmax;
    for (int i = 0; i < mmax; i++) {
        c->b[i] = vsm[i]; c->a[i] = vpm[i]; c->d[i] = thkm[i]; c->rho[i] = rhom[i];
    }
    int idispl = 0, idispr = 0;
    if (iwave == 1) { idispl = kmax; idispr = 0; }
    else if (iwave == 2) { idispl = 0; idispr = kmax; }
    int iverb[3] = {0, 0, 0};
    const float sone0 = 1.500f, ddc0 = 0.005f, h0 = 0.005f;
    c->llw = 1;
    if (c->b[0] <= 0.0f) c->llw = 2;
    const double one = 1.0e-2;
    if (nsph == 1) sphere(c, 0, 0);
    int jmn = 1, jsol = 1;
    float betmx = -1.e20f, betmn = 1.e20f;
    for (int i = 0; i < mmax; i++) {
        if (c->b[i] > 0.01f && c->b[i] < betmn) { betmn = c->b[i]; jmn = i + 1; jsol = 1; }
        else if (c->b[i] <= 0.01f && c->a[i] < betmn) { betmn = c->a[i]; jmn = i + 1; jsol = 0; }
        if (c->b[i] > betmx) betmx = c->b[i];
    }
    for (int ifunc = 1; ifunc <= 2; ifunc++) {
        if (ifunc == 1 && idispl <= 0) continue;
        if (ifunc == 2 && idispr <= 0) continue;
        if (nsph == 1) sphere(c, ifunc, 1);
        float ddc = ddc0, sone = sone0, h = h0, cc1;
        if (sone < 0.01f) sone = 2.0f;
        double onea = (double)sone;
        if (jsol == 0) cc1 = betmn;
        else cc1 = gtsolh(c->a[jmn - 1], c->b[jmn - 1]);
        cc1 = .95f * cc1;
        cc1 = .90f * cc1;
        double cc = (double)cc1, dc = fabs((double)ddc), c1 = cc, cm = cc, clow = 0.0, t1;
        for (int i = 0; i < kmax; i++) { cb[i] = 0.0; cvel[i] = 0.0; }
        int ift = 999;
        for (int iq = 1; iq <= mode; iq++) {
            int is = 1, ie = kmax, k, failed = 0;
            for (k = is; k <= ie; k++) {
                if (k >= ift) { failed = 1; break; }
                float t1a, t1b = 0.0f;
                int ifirst, iret;
                t1 = t[k - 1];
                if (igr > 0) {
                    t1a = (float)(t1 / (double)(1.f + h));
                    t1b = (float)(t1 / (double)(1.f - h));
                    t1 = (double)t1a;
                } else {
                    t1a = (float)t1;
                }
                if (k == is && iq == 1) { c1 = cc; clow = cc; ifirst = 1; }
                else if (k == is && iq > 1) { c1 = cvel[is - 1] + one * dc; clow = c1; ifirst = 1; }
                else if (k > is && iq > 1) {
                    ifirst = 0;
                    clow = cvel[k - 1] + one * dc;
                    c1 = cvel[k - 2];
                    if (c1 < clow) c1 = clow;
                } else { /* k > is, iq == 1 */
                    ifirst = 0;
                    c1 = cvel[k - 2] - onea * dc;
                    clow = cm;
                }
                iret = getsol(c, t1, &c1, clow, dc, cm, betmx, ifunc, ifirst);
                if (iret == -1) { failed = 1; break; }
                cvel[k - 1] = c1;
                if (igr > 0) {
                    t1 = (double)t1b;
                    ifirst = 0;
                    clow = cb[k - 1] + one * dc;
                    c1 = c1 - onea * dc;
                    iret = getsol(c, t1, &c1, clow, dc, cm, betmx, ifunc, ifirst);
                    if (iret == -1) c1 = cvel[k - 1];
                    cb[k - 1] = c1;
                } else {
                    c1 = 0.0;
                }
                float cc0 = (float)cvel[k - 1];
                float cc1b = (float)c1;
                if (igr == 0) {
                    cg[k - 1] = (double)cc0;
                } else {
                    float gvel = (1 / t1a - 1 / t1b) / (1 / (t1a * cc0) - 1 / (t1b * cc1b));
                    cg[k - 1] = (double)gvel;
                }
            }
            if (!failed) continue; /* go to 1800 */
            /* 1700 */
            if (iq <= 1) {
                if (iverb[ifunc] == 0) { iverb[ifunc] = 1; err = 1; }
            }
            /* 1750 */
            ift = k;
            for (int i = k; i <= ie; i++) cg[i - 1] = 0.0;
        }
    }
    if (n_dltar) *n_dltar = c->ncalls;
    return err;
}

void bho_surfdisp96_batch(int B, int Lmax, const int *nlay, const double *h, const double *vp,
                          const double *vs, const double *rho, int iflsph, int iwave, int mode,
                          int igr, int kmax, const double *t, double *out, int *err,
                          long *n_dltar_total, int nthreads)
{
    long total = 0;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 16) num_threads(nthreads) reduction(+ : total)
    for (int b = 0; b < B; b++) {
        float th[BHO_NL], a[BHO_NL], bb[BHO_NL], r[BHO_NL];
        int n = nlay[b];
        long nc = 0;
        for (int i = 0; i < n; i++) {
            th[i] = (float)h[(long)b * Lmax + i];
            a[i] = (float)vp[(long)b * Lmax + i];
            bb[i] = (float)vs[(long)b * Lmax + i];
            r[i] = (float)rho[(long)b * Lmax + i];
        }
        err[b] = bho_surfdisp96(th, a, bb, r, n, iflsph, iwave, mode, igr, kmax, t,
                                out + (long)b * kmax, &nc);
        total += nc;
    }
    if (n_dltar_total) *n_dltar_total = total;
}
