// rf_core.h -- receiver-function forward solver, restructured for one workgroup = M layered models.
//
// Reference path (rfmini): wrap.cpp:57-80 synrf_cwrap -> synrf.cpp:16-55 synrf (layer stack +
// earth flattening model.cpp:223-251) -> greens.cpp:685-756 calcresp -> :400-591 calcresp_core
// (interface coefficients :19-112, Mueller/Kennett top-down recursion :196-224) -> :343-398
// compute_rf (P/SV rotation, spectral division, Gauss filter) -> :136-158 iftr -> fork.cpp ccfork.
//
// Decomposition used here (all phases run by the threads of one workgroup on a block of LDS that
// holds M models; `phase_*` functions are written per "virtual thread" so that tests/hostsim can
// replay them on the CPU):
//   P1  one lane per (model, layer)        : depth = cumsum(h), earth flattening -> LDS
//   P2  one lane per (model, interface)    : frequency-independent R/T coefficient matrices -> LDS
//       (+ lane 0 of each model: displacement matrix, direct-wave delay t0, P/SV rotation)
//   P3  one lane per (model, frequency)    : phase matrices, top-down recursion, RF spectrum -> LDS
//       tasks are numbered model-major, 257 (= nsamp/2+1) per model, and dealt round-robin to the
//       workgroup's threads, so the awkward 2^k+1 task count wastes one partial round per M models
//   P4  all lanes                          : Hermitian fill, bit reversal, radix-2 butterflies in
//       LDS (twiddles from a table built on the host exactly like fork.cpp:50-51), scaled output
//
// Arithmetic follows the reference expression by expression; complex products/quotients use the
// same formulas as libgcc's __muldc3/__divdc3 (bh_common.h).  Off-diagonal zeros of the phase
// matrix e are dropped from products (adds of exact zeros).
#pragma once
#include "bh_common.h"
#include "bh_math.h"

namespace bh {

#ifndef BH_PI
#define BH_PI 3.14159265358979323846 /* M_PI */
#endif

// Host-prepared, launch-uniform parameters.  The derived constants are computed on the host with
// the reference's own expressions so that they are bit-identical to what rfmini uses.
struct RfLaunch {
    double slowness;   // p[s/deg] * 0.00899            wrap.cpp:55,76
    double p2;         // slowness^2                    greens.cpp:421
    double gauss;      // a
    double tshift;
    double nsv;        // <= 0: use vs of the top layer  rfmini_modrf.py:129-130
    double sigma;      // NaN: derive Poisson ratio from the top layer (rfmini_modrf.py:125-127)
    double dw;         // 2*pi*fsamp/nsamp               greens.cpp:360,507
    double qgauss;     // sqrt(pi)*fsamp/a               greens.cpp:361
    double sc;         // sqrt(1/n)                      fork.cpp:28
    double qn;         // 1/sqrt(n)                      greens.cpp:147
    double wref;       // 2*pi*fref, fref = 1 Hz         greens.cpp:447, synrf.cpp:25
    int nsamp, nfreq, log2n, waveno, nout, out_off, out_stride, Lmax;
    int nact;          // frequencies 0 .. nact-1 carry a Gauss-filter weight above 1e-24 (rf_host.h)
    int depth_input;   // 1: the `h` array holds depths z (single-model drop-in), 0: thicknesses
    int M;             // models per workgroup
};

// LDS block of one model, in doubles.  The FFT buffer [0, 2*nsamp) overlays the spectrum and -- once
// P3 is over -- the parameter/coefficient region.
// doubles between the coefficient blocks of consecutive interfaces: 32 used + 2, so that the lanes of
// phase 2 (one per interface) do not all store into the same banks (32 doubles = one bank row)
enum { RF_COEF = 34 };
struct RfLayout {
    int L, off_par, off_coef, off_sc, per_model;
};
BH_HD RfLayout rf_layout(int Lmax, int nsamp)
{
    RfLayout lo;
    int nfreq = nsamp / 2 + 1;
    lo.L = Lmax;
    // the spectrum occupies elements 0 .. n/2 of the (swizzled, rf_swz) FFT buffer: element n/2 may
    // land anywhere in its 16-element row, so the parameters start behind that row
    lo.off_par = 2 * (nfreq + 15);     // [9][L]: d, vp, vs, rho, 1/(pi qp), 1/(pi qs), 1/vp^2, 1/vs^2,
                                       //         interface coefficients real?
    lo.off_coef = lo.off_par + 9 * Lmax; // [L][RF_COEF]: rd, td, ru, tu of interface i (above layer i)
    lo.off_sc = lo.off_coef + RF_COEF * Lmax; // 16 scalars
    int need = lo.off_sc + 16;
    lo.per_model = need > 2 * nsamp ? need : 2 * nsamp;
    // A wave of phase 3 that straddles two models reads the same offsets of both blocks.  With a block
    // length that is a multiple of the bank row (2*nsamp doubles = 8 KiB) those pairs collide on every
    // read; 16 bytes of padding put the second model's accesses four banks further.
#if !defined(BH_RF_NO_PAD)
    lo.per_model += 2;
#endif
    return lo;
}
enum { RF_SC_H2 = 0, RF_SC_T0 = 8, RF_SC_M11 = 9, RF_SC_M12 = 10, RF_SC_M21 = 11, RF_SC_M22 = 12,
       RF_SC_DECOMP = 13, RF_SC_UNIFORM_Q = 14 };

BH_DEV void st_cd(double *p, cd v) { p[0] = v.re; p[1] = v.im; }
BH_DEV cd ld_cd(const double *p) { return mk(p[0], p[1]); }
BH_DEV void st_cm2(double *p, const cm2 &m)
{
    st_cd(p, m.c11); st_cd(p + 2, m.c12); st_cd(p + 4, m.c21); st_cd(p + 6, m.c22);
}
BH_DEV cm2 ld_cm2(const double *p)
{
    cm2 m;
    m.c11 = ld_cd(p); m.c12 = ld_cd(p + 2); m.c21 = ld_cd(p + 4); m.c22 = ld_cd(p + 6);
    return m;
}
BH_DEV cm2 ld_m2(const double *p, const cm2 *) { return ld_cm2(p); }
BH_DEV rm2 ld_m2(const double *p, const rm2 *)          // the real parts of a stored complex 2x2
{
    rm2 m;
    m.c11 = p[0]; m.c12 = p[2]; m.c21 = p[4]; m.c22 = p[6];
    return m;
}
BH_DEV cm2 to_cm2(const cm2 &m) { return m; }
BH_DEV cm2 to_cm2(const rm2 &m)
{
    cm2 r;
    r.c11 = mk(m.c11, 0.); r.c12 = mk(m.c12, 0.); r.c21 = mk(m.c21, 0.); r.c22 = mk(m.c22, 0.);
    return r;
}
BH_DEV cm2 cm2_zero()
{
    cm2 m;
    m.c11 = m.c12 = m.c21 = m.c22 = mk(0.0, 0.0);
    return m;
}

// ---- P1: depth + earth flattening of layer i (synrf.cpp:28-34, model.cpp:207-251) ----------------
BH_DEV void rf_phase1_layer(double *S, const RfLayout &lo, int nlay, int i, const double *h,
                            const double *vp, const double *vs, const double *rho,
                            const double *qp, const double *qs, int depth_input)
{
    const double R = 6371.0;
    double z, thick;
    if (depth_input) {
        z = h[i];
        thick = (i < nlay - 1) ? h[i + 1] - h[i] : -1.0;
    } else {
        double acc = 0.0;                       // z = concatenate(([0], cumsum(h)[:-1]))
        for (int k = 0; k < i; k++) acc += h[k];
        z = acc;
        thick = (i < nlay - 1) ? (acc + h[i]) - z : -1.0;   // z[i+1]-z[i]
    }
    double lvp = vp[i], lvs = vs[i], lrh = rho[i];
    double zb = z + thick, r = R - z, q = R / r;
    z = R * log(q);
    lvp *= q;
    lvs *= q;
    lrh /= q;
    bool lower_halfspace = !(thick > 0.) && !(lvp < 1. && lrh < 0.1);
    if (!lower_halfspace) {
        r = R - zb;
        q = R / r;
        zb = R * log(q);
        thick = zb - z;
    }
    double *par = S + lo.off_par;
    par[0 * lo.L + i] = thick;
    par[1 * lo.L + i] = lvp;
    par[2 * lo.L + i] = lvs;
    par[3 * lo.L + i] = lrh;
    // Q enters every frequency only through 1/(pi Q) and 1/(2 Q) = (pi/2) / (pi Q) (Mueller 1985
    // eq. 132, greens.cpp:539-540): keep the reciprocal, and 1/v^2 for the shared-Q form of phase 3
    par[4 * lo.L + i] = frcp(BH_PI * (qp ? qp[i] : 500.0));      // rfmini_modrf.py:119-120
    par[5 * lo.L + i] = frcp(BH_PI * (qs ? qs[i] : 225.0));
    par[6 * lo.L + i] = frcp(lvp * lvp);
    par[7 * lo.L + i] = frcp(lvs * lvs);
}

// ---- P2: interface coefficients ---------------------------------------------------------------------
// coeffm, P-SV part (greens.cpp:19-76): interface between medium 1 (above) and 2 (below).
BH_DEV void rf_coeffm(double u, double vp1, double vs1, double rho1, double vp2, double vs2,
                      double rho2, cm2 &rd, cm2 &td, cm2 &ru, cm2 &tu)
{
    double mue1 = rho1 * vs1 * vs1, mue2 = rho2 * vs2 * vs2, c = 2. * (mue1 - mue2), u2 = u * u,
           cu2 = c * u2, t1, t2, t3;
    cd rpp, rps, rsp, rss, tpp, tps, tsp, tss, d1, d2, t4, t5, t7;
    cd a1 = conj(csqrt_(mk(1. / (vp1 * vp1) - u2, 0.)));
    cd a2 = conj(csqrt_(mk(1. / (vp2 * vp2) - u2, 0.)));
    cd b1 = conj(csqrt_(mk(1. / (vs1 * vs1) - u2, 0.)));
    cd b2 = conj(csqrt_(mk(1. / (vs2 * vs2) - u2, 0.)));

    t1 = cu2 - rho1 + rho2;
    t2 = cu2 - rho1;
    t3 = cu2 + rho2;
    t4 = a1 * t3 - a2 * t2;

    d1 = (t1 * t1 * u2 + (a2 * (t2 * t2)) * b2) + (a2 * (rho1 * rho2)) * b1;
    d2 = (((a1 * (c * c * u2)) * a2) * b1) * b2 + (a1 * (t3 * t3)) * b1 + (a1 * (rho1 * rho2)) * b2;
    t5 = rdiv(1., d1 + d2);
    t7 = t5 * (2. * rho1);

    rpp = (d2 - d1) * t5;
    rps = ((a1 * (-2. * u)) * t5) * (t1 * t3 + (a2 * (c * t2)) * b2);
    tpp = (a1 * t7) * (b1 * t3 - b2 * t2);
    tps = (((-a1) * t7) * u) * (t1 + (a2 * c) * b1);
    rss = ((d2 - d1) - (a1 * b2 - a2 * b1) * (2. * rho1 * rho2)) * t5;
    rsp = ((b1 * (2. * u)) * t5) * (t1 * t3 + (a2 * (c * t2)) * b2);
    tss = (b1 * t7) * t4;
    tsp = ((b1 * t7) * u) * (t1 + (a1 * c) * b2);
    rd.c11 = rpp; rd.c12 = rsp; rd.c21 = rps; rd.c22 = rss;
    td.c11 = tpp; td.c12 = tsp; td.c21 = tps; td.c22 = tss;

    d1 = (t1 * t1 * u2 + (a1 * (t3 * t3)) * b1) + (a1 * (rho1 * rho2)) * b2;
    d2 = (((a1 * (c * c * u2)) * a2) * b1) * b2 + (a2 * (t2 * t2)) * b2 + (a2 * (rho1 * rho2)) * b1;
    t5 = rdiv(1., d1 + d2);
    t7 = t5 * (2. * rho2);

    rpp = (d2 - d1) * t5;
    rps = ((a2 * (2. * u)) * t5) * (t1 * t2 + (a1 * (c * t3)) * b1);
    tpp = (a2 * t7) * (b1 * t3 - b2 * t2);
    tps = (((-a2) * t7) * u) * (t1 + (a1 * c) * b2);
    rss = ((d2 - d1) - (a2 * b1 - a1 * b2) * (2. * rho1 * rho2)) * t5;
    rsp = ((b2 * (-2. * u)) * t5) * (t1 * t2 + (a1 * (c * t3)) * b1);
    tss = (b2 * t7) * t4;
    tsp = ((b2 * t7) * u) * (t1 + (a2 * c) * b1);
    ru.c11 = rpp; ru.c12 = rsp; ru.c21 = rps; ru.c22 = rss;
    tu.c11 = tpp; tu.c12 = tsp; tu.c21 = tps; tu.c22 = tss;
}

// coeffs: free surface (greens.cpp:87-112)
BH_DEV void rf_coeffs(double u, double vp, double vs, cm2 &ru)
{
    double u2 = u * u;
    cd a = csqrt_(mk(1. / (vp * vp) - u2, 0.));
    cd b = csqrt_(mk(1. / (vs * vs) - u2, 0.));
    cd t1 = mk(2. * vs * vs, 0.);
    cd t2 = t1 * u2 - 1.;
    cd d1 = t2 * t2;
    cd d2 = (((t1 * t1) * u2) * a) * b;
    cd d = d1 + d2;
    cd t3 = (((2. * t1) * u) * t2) / d;
    cd rpp = (d2 - d1) / d;
    ru.c11 = rpp; ru.c12 = (-b) * t3; ru.c21 = a * t3; ru.c22 = rpp;
}

// displacement_matrix (greens.cpp:307-322), returned already multiplied by 2 (greens.cpp:572)
BH_DEV cm2 rf_displacement2(double p, double vp, double vs)
{
    double vp2 = vp * vp, vs2 = vs * vs, p2 = p * p, x = 1. - 2. * vs2 * p2;
    cd a1 = conj(csqrt_(mk(1. / vp2 - p2, 0.)));
    cd b1 = conj(csqrt_(mk(1. / vs2 - p2, 0.)));
    cd q = rdiv(1., x * x + (a1 * (4. * vs2 * vs2 * p2)) * b1);
    cm2 m;
    m.c11 = ((((q * a1) * b1) * 2.) * vs2) * p;
    m.c12 = (q * b1) * (1. - 2. * vs2 * p2);
    m.c21 = (q * a1) * (1. - 2. * vs2 * p2);
    m.c22 = (((((-q) * a1) * b1) * 2.) * vs2) * p;
    m.c11 = m.c11 * 2.0; m.c12 = m.c12 * 2.0; m.c21 = m.c21 * 2.0; m.c22 = m.c22 * 2.0;
    return m;
}

BH_DEV void rf_phase2_interface(double *S, const RfLayout &lo, const RfLaunch &P, int nlay, int i,
                                double vp0_in, double vs0_in)
{
    const double *par = S + lo.off_par;
    double *coef = S + lo.off_coef + RF_COEF * i;
    cm2 rd = cm2_zero(), td = cm2_zero(), ru = cm2_zero(), tu = cm2_zero();
    double u = P.slowness;
    if (i == 0) {
        rf_coeffs(u, par[1 * lo.L], par[2 * lo.L], ru);
        // per-model scalars
        double *sc = S + lo.off_sc;
        st_cm2(sc + RF_SC_H2, rf_displacement2(u, par[1 * lo.L], par[2 * lo.L]));
        double t0 = 0.;                         // greens.cpp:510-526 (includes half-space d = -1)
        const double *v = par + (P.waveno == 0 ? 1 : 2) * lo.L;
        for (int k = 0; k < nlay; k++) t0 += par[k] * sqrt(1. / (v[k] * v[k]) - P.p2);
        sc[RF_SC_T0] = t0;
        // one Q for all layers (what BayHunter passes): the complex velocity factor is then the same
        // for every layer of a frequency
        bool uniform = true;
        for (int k = 1; k < nlay - 1; k++)
            uniform = uniform && par[4 * lo.L + k] == par[4 * lo.L] && par[5 * lo.L + k] == par[5 * lo.L];
        sc[RF_SC_UNIFORM_Q] = uniform ? 1.0 : 0.0;
        // rotation velocities: wrap.cpp:13,73-74 with rfmini_modrf.py:125-130
        double sigma = P.sigma;
        if (!(sigma == sigma)) {
            double vpvs = vp0_in / vs0_in;
            sigma = (2 - vpvs * vpvs) / (2 - 2 * (vpvs * vpvs));
        }
        double nsv = P.nsv > 0 ? P.nsv : vs0_in;
        double vpt = nsv * sqrt((1. - (sigma)) / (.5 - (sigma))), vst = nsv;
        double pp = P.slowness;
        int dec = (vst > 0.01 && fabs(pp) > 0.0001) ? 1 : 0;   // greens.cpp:365
        double a = sqrt(1. / (vpt * vpt) - pp * pp), b = sqrt(1. / (vst * vst) - pp * pp);
        sc[RF_SC_M11] = -(2 * vst * vst * pp * pp - 1.) / (vpt * a);   // decomp, greens.cpp:328-333
        sc[RF_SC_M12] = 2. * pp * vst * vst / vpt;
        sc[RF_SC_M21] = -2. * pp * vst;
        sc[RF_SC_M22] = (1. - 2. * vst * vst * pp * pp) / (vst * b);
        sc[RF_SC_DECOMP] = (double)dec;
    } else {
        rf_coeffm(u, par[1 * lo.L + i - 1], par[2 * lo.L + i - 1], par[3 * lo.L + i - 1],
                  par[1 * lo.L + i], par[2 * lo.L + i], par[3 * lo.L + i], rd, td, ru, tu);
    }
    st_cm2(coef, rd); st_cm2(coef + 8, td); st_cm2(coef + 16, ru); st_cm2(coef + 24, tu);
    const cm2 *four[4] = {&rd, &td, &ru, &tu};
    bool real = true;
    for (int k = 0; k < 4; k++)
        real = real && four[k]->c11.im == 0. && four[k]->c12.im == 0. && four[k]->c21.im == 0. &&
               four[k]->c22.im == 0.;
    S[lo.off_par + 8 * lo.L + i] = real ? 1.0 : 0.0;
}

// ---- P3: one frequency of one model (greens.cpp:528-585 + compute_rf :377-395) ----------------------
// What a frequency needs that does not depend on the model: ln(w/wref) of the anelastic velocities
// (greens.cpp:530,539-540) and the Gauss filter x time shift factor of compute_rf (:389-392).  Both come
// from a table the host fills with the reference's own expressions (rf_host.h, rf_fill_freq_table) --
// glibc's log and std::exp(Complex), bit for bit what rfmini uses -- instead of a device log, exp and
// sincos per (model, frequency).  Three doubles per frequency: lgw, re(cq), im(cq).
struct RfFreq {
    double lgw;
    cd cq;
};
enum { RF_FTAB = 3 };
BH_DEV RfFreq rf_freq_load(const double *BH_RESTRICT ftab, int j)
{
    RfFreq F;
    F.lgw = ftab[RF_FTAB * j];
    F.cq = mk(ftab[RF_FTAB * j + 1], ftab[RF_FTAB * j + 2]);
    return F;
}

#if !defined(BH_HOSTSIM)
#pragma clang fp contract(fast)
#endif
template <class M>
BH_DEV cd rf_phase3_body(const double *S, const RfLayout &lo, const RfLaunch &P, int nlay, int j,
                         const RfFreq &F, cd *zr_r, cd *zr_z)
{
    const double *par = S + lo.off_par;
    const double *coef = S + lo.off_coef;
    const double *sc = S + lo.off_sc;
    const double w = P.dw * j;
    const double lgw = F.lgw;
    cm2 nb = cm2_zero(), q = cm2_zero(), g = cm2_zero();
    // complex velocity v (1 + ln(w/wref)/(pi Q) + i/(2 Q)), Mueller (1985) eq. 132: 1/v_c^2 =
    // (1/v^2) * 1/f^2 with f the bracket; with one Q for all layers f is a per-frequency constant
    const bool uniform = sc[RF_SC_UNIFORM_Q] != 0.0;
    cd gp, gs;
    {
        const double ap = par[4 * lo.L], as = par[5 * lo.L];
        cd fp = mk(1. + lgw * ap, ap * (0.5 * BH_PI)), fs = mk(1. + lgw * as, as * (0.5 * BH_PI));
        gp = crecip(fp * fp);
        gs = crecip(fs * fs);
    }
    for (int i = 0; i < nlay - 1; i++) {        // reference layer index i+1 = 1 .. nlay-1
        const double wd = w * par[i];               // exp(-i w d * slowness): (-i wd)(a + ib) = wd b - i wd a
        if (!uniform && i > 0) {                // this layer has its own Q
            const double ap = par[4 * lo.L + i], as = par[5 * lo.L + i];
            cd fp = mk(1. + lgw * ap, ap * (0.5 * BH_PI)), fs = mk(1. + lgw * as, as * (0.5 * BH_PI));
            gp = crecip(fp * fp);
            gs = crecip(fs * fs);
        }
        cd plc = csqrt_fast(gp * par[6 * lo.L + i] - P.p2);         // Q finite -> im != 0
        cd slc = csqrt_fast(gs * par[7 * lo.L + i] - P.p2);
#if defined(BH_RF_OLD_CEXP)                        // (A/B switch: the general complex product and the clamped exp)
        cd e11 = cexp_(mk(0., -wd) * plc), e22 = cexp_(mk(0., -wd) * slc);
#else
        cd e11 = cexp_bounded(mk(wd * plc.im, -(wd * plc.re))), e22 = cexp_bounded(mk(wd * slc.im, -(wd * slc.re)));
#endif
        const double *ci = coef + RF_COEF * i, *cn = coef + RF_COEF * (i + 1);
        cm2 nt;
        if (i == 0) nt = to_cm2(ld_m2(ci + 16, (const M *)nullptr));                   // nt = ru[1]
        else nt = ld_m2(ci + 16, (const M *)nullptr) + (ld_m2(ci + 8, (const M *)nullptr) * nb) * q;     // ru + td*nb*q
        {   // nb = e*nt*e, greens.cpp:829-845
            cd e12 = e11 * e22, e11s = e11 * e11, e22s = e22 * e22;
            nb.c11 = nt.c11 * e11s; nb.c12 = nt.c12 * e12; nb.c21 = nt.c21 * e12; nb.c22 = nt.c22 * e22s;
        }
        {   // q = inv(I - rd[i+1]*nb) * tu[i+1]
            cm2 x = ld_m2(cn, (const M *)nullptr) * nb;
            x.c11 = mk(1., 0.) - x.c11; x.c12 = mk(0., 0.) - x.c12;
            x.c21 = mk(0., 0.) - x.c21; x.c22 = mk(1., 0.) - x.c22;
            cd qi = crecip(x.c11 * x.c22 - x.c12 * x.c21);
            cm2 inv;
            inv.c11 = qi * x.c22; inv.c12 = (-qi) * x.c12; inv.c21 = (-qi) * x.c21; inv.c22 = qi * x.c11;
            q = inv * ld_m2(cn + 24, (const M *)nullptr);
        }
        if (i == 0) {                                             // g = e*q
            g.c11 = e11 * q.c11; g.c12 = e11 * q.c12; g.c21 = e22 * q.c21; g.c22 = e22 * q.c22;
        } else {                                                  // g = (g*e)*q
            cm2 ge;
            ge.c11 = g.c11 * e11; ge.c12 = g.c12 * e22; ge.c21 = g.c21 * e11; ge.c22 = g.c22 * e22;
            g = ge * q;
        }
    }
    cm2 t = ld_cm2(sc + RF_SC_H2) * g;                            // t = 2*h*g[nlay-1]
    cd cr, cz;
    if (P.waveno == 0) { cr = t.c11; cz = t.c21; } else { cr = t.c12; cz = t.c22; }
    // exp(i w t0), the direct-wave delay (greens.cpp:583-585): a unit factor common to cr and cz.  The
    // vertical / radial spectra need it; the receiver function is their ratio, in which it cancels -- all
    // that must survive is the reference's NaN when t0 is (post-critical incidence: the root of a negative
    // number in the delay sum, greens.cpp:510-526).
    if (zr_r) {
        cd qq = cexp_(mk(0., w * sc[RF_SC_T0]));
        cr = cr * qq;
        cz = cz * qq;
    } else {
        const double t0nan = sc[RF_SC_T0] * 0.0;                  // 0, or NaN for a NaN / infinite delay
        cr = cr + mk(t0nan, t0nan);
    }
    if (sc[RF_SC_DECOMP] != 0.0) {                                // decomp, greens.cpp:335-340
        cd cx = cz * sc[RF_SC_M11] + cr * sc[RF_SC_M12];
        cd cy = cz * sc[RF_SC_M21] + cr * sc[RF_SC_M22];
        cz = cx;
        cr = cy;
    }
    const cd arr_r = cr, arr_z = cz;                              // the cr[] / cz[] arrays of compute_rf
    if (P.waveno == 1) { cd tmp = cz; cz = cr; cr = tmp; }        // greens.cpp:369-373
    double denom = cz.re * cz.re + cz.im * cz.im;                 // real(cz*conj(cz)); no water level
    cd crf = (cr * conj(cz)) / denom;
    const cd cq = F.cq;                                           // q exp(-(w/a)^2/4 - i w tshift), host table
    if (zr_r) { *zr_r = arr_r * cq; *zr_z = arr_z * cq; }       // greens.cpp:393-394 (for iftr2)
    return crf * cq;
}

BH_DEV cd rf_phase3_task(const double *S, const RfLayout &lo, const RfLaunch &P, int nlay, int j,
                         const RfFreq &F, cd *zr_r = nullptr, cd *zr_z = nullptr)
{
    // all interface matrices of this model real (no post-critical wave anywhere): half the
    // multiplications in the products with rd, td, ru, tu
    const double *flag = S + lo.off_par + 8 * lo.L;
    bool real = true;
    for (int i = 0; i < nlay; i++) real = real && flag[i] != 0.0;
    return real ? rf_phase3_body<rm2>(S, lo, P, nlay, j, F, zr_r, zr_z)
                : rf_phase3_body<cm2>(S, lo, P, nlay, j, F, zr_r, zr_z);
}

#if !defined(BH_HOSTSIM)
#pragma clang fp contract(off)
#endif

// ---- P4: inverse FFT pieces (greens.cpp:136-158, fork.cpp:10-60) ---------------------------------------
// Where complex element e of the FFT buffer lives.  The bit-reversal permutation makes the 64 lanes
// of a wave touch elements 8 apart (128 bytes): all in the same few LDS banks, a 32-way conflict
// (measured: a third of the kernel's bank-conflict cycles for 5 % of its LDS instructions).  XOR-ing
// bits 4..7 of the index into bits 0..3 permutes elements only inside their 16-element (256-byte)
// row, so unit-stride runs (butterflies, spectrum, output) stay conflict-free, while elements 8
// apart now spread over a row: 2-way.
BH_HD int rf_swz(int e) { return e ^ ((e >> 4) & 15); }
BH_DEV cd rf_xld(const double *X, int e) { return ld_cd(X + 2 * rf_swz(e)); }
BH_DEV void rf_xst(double *X, int e, cd v) { st_cd(X + 2 * rf_swz(e), v); }

BH_DEV unsigned rf_bitrev(unsigned i, int log2n)
{
    unsigned r = 0;
    for (int b = 0; b < log2n; b++) { r = (r << 1) | (i & 1u); i >>= 1; }
    return r;
}
// Hermitian extension: cx[i] = conj(cx[n-i]) for i in (n/2, n)
BH_DEV void rf_fft_hermitian(double *X, int n, int i)
{
    rf_xst(X, i, conj(rf_xld(X, n - i)));
}
// bit-reversal permutation fused with the 1/sqrt(n) scaling of fork.cpp:30-45: handle pair (i, rev i)
BH_DEV void rf_fft_bitrev_scale(double *X, int n, int log2n, double sc, int i)
{
    int j = (int)rf_bitrev((unsigned)i, log2n);
    if (i > j) return;
    cd xi = rf_xld(X, i), xj = rf_xld(X, j);
    rf_xst(X, j, xi * sc);
    rf_xst(X, i, xj * sc);
    (void)n;
}
// iftr2 (greens.cpp:161-194): entry i of cx = cx1 + i*cx2 built from the two half spectra
BH_DEV cd rf_fft_pair_entry(const double *spec_r, const double *spec_z, int n, int i)
{
    cd c1 = (i <= n / 2) ? ld_cd(spec_r + 2 * i) : conj(ld_cd(spec_r + 2 * (n - i)));
    cd c2 = (i <= n / 2) ? ld_cd(spec_z + 2 * i) : conj(ld_cd(spec_z + 2 * (n - i)));
    return c1 + mk(0., 1.) * c2;
}
// butterfly number bf (0 .. n/2-1) of the stage with half-span l; tw[l+m] = exp(i*pi*m/l)
BH_DEV void rf_fft_butterfly(double *X, const double *tw, int l, int bf)
{
    int m = bf & (l - 1), i = ((bf - m) << 1) + m;
    cd w = ld_cd(tw + 2 * (l + m));
    cd a = rf_xld(X, i), b = rf_xld(X, i + l);
    cd tmp = w * b;
    rf_xst(X, i + l, a - tmp);
    rf_xst(X, i, a + tmp);
}


}  // namespace bh
