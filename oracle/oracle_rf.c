/*
 * oracle_rf.c -- CPU restatement of the reference's rfmini receiver-function solver.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle_port.h).  Follows /root/reference/src/extensions/rfmini/
 * {wrap.cpp, synrf.cpp, model.cpp, greens.cpp, fork.cpp, cmat2.h}; every function names the lines
 * it restates.  C99 `double _Complex` arithmetic lowers to the same libgcc (__muldc3/__divdc3) and
 * glibc (csqrt, cexp) routines as the reference's std::complex<double>, so the output is
 * bit-identical to the g++-compiled reference (checked against oracle/_ref/librfmini_ref.so).
 *
 * Only the non-partial-derivative branch is restated (BayHunter never asks for drdp,
 * synrf.cpp:51).  The SH block of greens.cpp:553-560 reads uninitialised memory and its result
 * is discarded (greens.cpp:716-717) -- deliberately not reproduced.
 *
 * Pinning: bit-identical to oracle/_ref on the seeded model sets of tests/golden/make_golden.py;
 * reproduces tutorial/observed/st3_{prf,srf}.dat to 1e-4 (those files stem from an older build).
 */
#include <complex.h>
#include <math.h>
#include <stdlib.h>
#include <omp.h>
#include "oracle_port.h"

typedef double _Complex cplx;
typedef struct { cplx c11, c12, c21, c22; } cmat2;

#define CX(re, im) CMPLX((re), (im))

/* ---- cmat2.h ---- */
static inline cmat2 cm_mul(cmat2 x, cmat2 y) /* cmat2.h:176-183 */
{
    cmat2 r;
    r.c11 = x.c11 * y.c11 + x.c12 * y.c21;
    r.c12 = x.c11 * y.c12 + x.c12 * y.c22;
    r.c21 = x.c21 * y.c11 + x.c22 * y.c21;
    r.c22 = x.c21 * y.c12 + x.c22 * y.c22;
    return r;
}
static inline cmat2 cm_add(cmat2 x, cmat2 y) /* cmat2.h:118-123 */
{
    cmat2 r = { x.c11 + y.c11, x.c12 + y.c12, x.c21 + y.c21, x.c22 + y.c22 };
    return r;
}
static inline cmat2 cm_sub(cmat2 x, cmat2 y) /* cmat2.h:131-136 */
{
    cmat2 r = { x.c11 - y.c11, x.c12 - y.c12, x.c21 - y.c21, x.c22 - y.c22 };
    return r;
}
static inline cplx rdivc(double x, cplx z) { return CX(x, 0.0) / z; } /* T / complex<T> */
static inline cplx cscale(cplx z, double r) { return CX(creal(z) * r, cimag(z) * r); }
static inline cmat2 cm_inv(cmat2 x) /* cmat2.h:144-153 */
{
    cplx q = rdivc(1., x.c11 * x.c22 - x.c12 * x.c21);
    cmat2 r = { q * x.c22, (-q) * x.c12, (-q) * x.c21, q * x.c11 };
    return r;
}
static inline cmat2 cm_rscale(double r, cmat2 x) /* cmat2.h:199-204 */
{
    cmat2 o = { cscale(x.c11, r), cscale(x.c12, r), cscale(x.c21, r), cscale(x.c22, r) };
    return o;
}
static inline cmat2 cm_diag(cplx f) { cmat2 r = { f, 0., 0., f }; return r; } /* cmat2.h:57-62 */

/* e*x*e for diagonal e (greens.cpp:829-845) */
static inline cmat2 exe(cmat2 e, cmat2 x)
{
    cplx e11 = e.c11, e22 = e.c22, e12;
    e12 = e11 * e22;
    e11 = e11 * e11;
    e22 = e22 * e22;
    cmat2 r = { x.c11 * e11, x.c12 * e12, x.c21 * e12, x.c22 * e22 };
    return r;
}

typedef struct { double z, h, vp, vs, rh, qp, qs; } layer_t;

/* FlatLayer::isLowerHalfspace (model.cpp:207-216) */
static int is_lower_halfspace(const layer_t *l)
{
    if (l->h > 0.) return 0;
    if (l->vp < 1. && l->rh < 0.1) return 0;
    return 1;
}

/* FlatLayer::flatten (model.cpp:221-251) */
static void flatten(layer_t *l)
{
    const double R = 6371.0;
    double zb = l->z + l->h, r = R - l->z, q = R / r;
    l->z = R * log(q);
    l->vp *= q;
    l->vs *= q;
    l->rh /= q;
    if (!is_lower_halfspace(l)) {
        r = R - zb;
        q = R / r;
        zb = R * log(q);
        l->h = zb - l->z;
    }
}

/* coeffm, P-SV part (greens.cpp:19-76); SH part (:78-84) feeds nothing */
static void coeffm(double u, double vp1, double vs1, double rho1, double vp2, double vs2,
                   double rho2, cmat2 *rd, cmat2 *td, cmat2 *ru, cmat2 *tu)
{
    double mue1 = rho1 * vs1 * vs1, mue2 = rho2 * vs2 * vs2, c = 2. * (mue1 - mue2), u2 = u * u,
           cu2 = c * u2, t1, t2, t3;
    cplx rpp, rps, rsp, rss, tpp, tps, tsp, tss, d1, d2, t4, t5, t7;
    cplx a1 = conj(csqrt(CX(1. / (vp1 * vp1) - u2, 0.)));
    cplx a2 = conj(csqrt(CX(1. / (vp2 * vp2) - u2, 0.)));
    cplx b1 = conj(csqrt(CX(1. / (vs1 * vs1) - u2, 0.)));
    cplx b2 = conj(csqrt(CX(1. / (vs2 * vs2) - u2, 0.)));

    t1 = cu2 - rho1 + rho2;
    t2 = cu2 - rho1;
    t3 = cu2 + rho2;
    t4 = cscale(a1, t3) - cscale(a2, t2);

    d1 = cscale(a2, t2 * t2) * b2;
    d1 = t1 * t1 * u2 + d1;                  /* double + complex */
    d1 = d1 + cscale(a2, rho1 * rho2) * b1;
    /* d1  = t1*t1*u2 + t2*t2*a2*b2 + rho1*rho2*a2*b1   (greens.cpp:42) */
    d2 = cscale(a1, c * c * u2) * a2 * b1 * b2 + cscale(a1, t3 * t3) * b1
         + cscale(a1, rho1 * rho2) * b2;     /* greens.cpp:43 */
    t5 = rdivc(1., d1 + d2);
    t7 = cscale(t5, 2. * rho1);

    rpp = (d2 - d1) * t5;
    rps = cscale(a1, -2. * u) * t5 * (t1 * t3 + cscale(a2, c * t2) * b2);
    tpp = a1 * t7 * (cscale(b1, t3) - cscale(b2, t2));
    tps = cscale((-a1) * t7, u) * (t1 + cscale(a2, c) * b1);
    rss = (d2 - d1 - cscale(a1 * b2 - a2 * b1, 2. * rho1 * rho2)) * t5;
    rsp = cscale(b1, 2. * u) * t5 * (t1 * t3 + cscale(a2, c * t2) * b2);
    tss = b1 * t7 * t4;
    tsp = cscale(b1 * t7, u) * (t1 + cscale(a1, c) * b2);
    rd->c11 = rpp; rd->c12 = rsp; rd->c21 = rps; rd->c22 = rss;
    td->c11 = tpp; td->c12 = tsp; td->c21 = tps; td->c22 = tss;

    d1 = t1 * t1 * u2 + cscale(a1, t3 * t3) * b1;
    d1 = d1 + cscale(a1, rho1 * rho2) * b2;  /* greens.cpp:61 */
    d2 = cscale(a1, c * c * u2) * a2 * b1 * b2 + cscale(a2, t2 * t2) * b2
         + cscale(a2, rho1 * rho2) * b1;     /* greens.cpp:62 */
    t5 = rdivc(1., d1 + d2);
    t7 = cscale(t5, 2. * rho2);

    rpp = (d2 - d1) * t5;
    rps = cscale(a2, 2. * u) * t5 * (t1 * t2 + cscale(a1, c * t3) * b1);
    tpp = a2 * t7 * (cscale(b1, t3) - cscale(b2, t2));
    tps = cscale((-a2) * t7, u) * (t1 + cscale(a1, c) * b2);
    rss = (d2 - d1 - cscale(a2 * b1 - a1 * b2, 2. * rho1 * rho2)) * t5;
    rsp = cscale(b2, -2. * u) * t5 * (t1 * t2 + cscale(a1, c * t3) * b1);
    tss = b2 * t7 * t4;
    tsp = cscale(b2 * t7, u) * (t1 + cscale(a2, c) * b1);
    ru->c11 = rpp; ru->c12 = rsp; ru->c21 = rps; ru->c22 = rss;
    tu->c11 = tpp; tu->c12 = tsp; tu->c21 = tps; tu->c22 = tss;
}

/* coeffs: free surface (greens.cpp:87-112) */
static void coeffs(double u, double vp, double vs, cmat2 *ru)
{
    double u2 = u * u;
    cplx rpp, rps, rsp, rss, a, b, t1, t2, t3, d, d1, d2;
    a = csqrt(CX(1. / (vp * vp) - u2, 0.));
    b = csqrt(CX(1. / (vs * vs) - u2, 0.));
    t1 = 2. * vs * vs;
    t2 = t1 * u2 - 1.;
    d1 = t2 * t2;
    d2 = t1 * t1 * u2 * a * b;
    d = d1 + d2;
    t3 = 2. * t1 * u * t2 / d;
    rpp = (d2 - d1) / d;
    rsp = -b * t3;
    rps = a * t3;
    rss = rpp;
    ru->c11 = rpp; ru->c12 = rsp; ru->c21 = rps; ru->c22 = rss;
}

/* displacement_matrix (greens.cpp:307-322) */
static void displacement_matrix(double p, double vp, double vs, cmat2 *m)
{
    double vp2 = vp * vp, vs2 = vs * vs, p2 = p * p, x = 1. - 2. * vs2 * p2;
    cplx a1 = conj(csqrt(CX(1. / vp2 - p2, 0.)));
    cplx b1 = conj(csqrt(CX(1. / vs2 - p2, 0.)));
    cplx q = rdivc(1., x * x + cscale(a1, 4. * vs2 * vs2 * p2) * b1);
    m->c11 = cscale(cscale(q * a1 * b1, 2.), vs2);
    m->c11 = cscale(m->c11, p);
    m->c12 = cscale(q * b1, 1. - 2. * vs2 * p2);
    m->c21 = cscale(q * a1, 1. - 2. * vs2 * p2);
    m->c22 = cscale(cscale(cscale((-q) * a1 * b1, 2.), vs2), p);
}

/* ccfork (fork.cpp:10-60) */
static void ccfork(int n, cplx *x, int signi)
{
    cplx w, tmp;
    double sc;
    int i, istep, j = 0, l, m;
    sc = sqrt(1. / (double)n);
    for (i = 0; i < n; i++) {
        if (i <= j) {
            tmp = cscale(x[j], sc);
            x[j] = cscale(x[i], sc);
            x[i] = tmp;
        }
        m = n >> 1;
        do {
            if (j < m) break;
            j -= m;
            m >>= 1;
        } while (m >= 1);
        j += m;
    }
    l = 1;
    do {
        istep = 2 * l;
        for (m = 0; m < l; m++) {
            w = cexp(CX(0.0, M_PI * (double)(signi * m) / (double)l));
            for (i = m; i < n; i += istep) {
                tmp = w * x[i + l];
                x[i + l] = x[i] - tmp;
                x[i] += tmp;
            }
        }
        l = istep;
    } while (l < n);
}

/* iftr (greens.cpp:136-158) */
static void iftr(int nsamp, const cplx *cf, double *f, cplx *cx)
{
    double q = 1. / sqrt((double)nsamp);
    for (int i = 0; i < nsamp / 2 + 1; i++) cx[i] = cf[i];
    for (int i = nsamp / 2 + 1; i < nsamp; i++) cx[i] = conj(cx[nsamp - i]);
    ccfork(nsamp, cx, 1);
    for (int i = 0; i < nsamp; i++) f[i] = q * creal(cx[i]);
}

/* iftr2 (greens.cpp:161-194) */
static void iftr2(int nsamp, const cplx *cf1, const cplx *cf2, double *f1, double *f2, cplx *cx,
                  cplx *cx1, cplx *cx2)
{
    double q = 1. / sqrt((double)nsamp);
    cplx imi = CX(0., 1.);
    for (int i = 0; i < nsamp / 2 + 1; i++) { cx1[i] = cf1[i]; cx2[i] = cf2[i]; }
    for (int i = nsamp / 2 + 1; i < nsamp; i++) {
        cx1[i] = conj(cx1[nsamp - i]);
        cx2[i] = conj(cx2[nsamp - i]);
    }
    for (int i = 0; i < nsamp; i++) cx[i] = cx1[i] + imi * cx2[i];
    ccfork(nsamp, cx, 1);
    for (int i = 0; i < nsamp; i++) { f1[i] = q * creal(cx[i]); f2[i] = q * cimag(cx[i]); }
}

/* decomp (greens.cpp:324-341) */
static void decomp(int n, cplx *cz, cplx *cr, double p, double vp, double vs)
{
    double a = sqrt(1. / (vp * vp) - p * p), b = sqrt(1. / (vs * vs) - p * p),
           m11 = -(2 * vs * vs * p * p - 1.) / (vp * a), m12 = 2. * p * vs * vs / vp,
           m21 = -2. * p * vs, m22 = (1. - 2. * vs * vs * p * p) / (vs * b);
    for (int i = 0; i < n; i++) {
        cplx cx = cscale(cz[i], m11) + cscale(cr[i], m12);
        cplx cy = cscale(cz[i], m21) + cscale(cr[i], m22);
        cz[i] = cx;
        cr[i] = cy;
    }
}

/* compute_rf (greens.cpp:343-398); the water level is not applied (:384) */
static void compute_rf(int wave_type, cplx *cr, cplx *cz, int nsamp, double fsamp, double tshift,
                       double a, double p, double vp0, double vs0, cplx *crf)
{
    double w, wa, dw = 2.0 * M_PI * fsamp / nsamp, denom, q = sqrt(M_PI) * fsamp / a;
    int nfreq = nsamp / 2 + 1;
    cplx cq;
    if (vs0 > 0.01 && fabs(p) > 0.0001) decomp(nfreq, cz, cr, p, vp0, vs0);
    if (wave_type == 1) { cplx *tmp = cz; cz = cr; cr = tmp; }
    for (int j = 0; j < nfreq; j++) {
        w = dw * j;
        denom = creal(cz[j] * conj(cz[j]));
        crf[j] = cr[j] * conj(cz[j]);
        crf[j] = CX(creal(crf[j]) / denom, cimag(crf[j]) / denom);
        wa = w / a;
        wa = (wa > 50.0) ? 50.0 : wa;
        cq = cscale(cexp(CX(-0.25 * (wa * wa), -w * tshift)), q);
        crf[j] = crf[j] * cq;
        cr[j] = cr[j] * cq;
        cz[j] = cz[j] * cq;
    }
}

/* calcresp_core, non-PD branch (greens.cpp:400-591) + calcresp (:685-756) */
static void calcresp(int nlay, layer_t *lay /*1-based*/, int wave_type, double slowness,
                     double fref, int nsamp, double fsamp, double tshift, double a, double vp_top,
                     double vs_top, double *zz, double *rr, double *rf)
{
    int nfreq = nsamp / 2 + 1;
    double p = slowness, p2 = p * p, dw, wref;
    cplx ii = CX(0., 1.);
    cmat2 *ru = calloc(nlay + 2, sizeof(cmat2)), *rd = calloc(nlay + 2, sizeof(cmat2)),
          *tu = calloc(nlay + 2, sizeof(cmat2)), *td = calloc(nlay + 2, sizeof(cmat2)),
          *nb = calloc(nlay + 2, sizeof(cmat2)), *nt = calloc(nlay + 2, sizeof(cmat2)),
          *g = calloc(nlay + 2, sizeof(cmat2)), *e = calloc(nlay + 2, sizeof(cmat2));
    cplx *cz = calloc(nfreq, sizeof(cplx)), *cr = calloc(nfreq, sizeof(cplx)),
         *crf = calloc(nfreq, sizeof(cplx)), *cx = calloc(3 * (size_t)nsamp, sizeof(cplx));
    cmat2 t, h;

    wref = 2. * M_PI * fref;
    for (int i = 1; i <= nlay; i++) { /* greens.cpp:462-468 via coeff :114-132 */
        if (i == 1) {
            coeffs(p, lay[1].vp, lay[1].vs, &ru[1]);
            rd[1] = td[1] = tu[1] = cm_diag(0.);
        } else {
            coeffm(p, lay[i - 1].vp, lay[i - 1].vs, lay[i - 1].rh, lay[i].vp, lay[i].vs, lay[i].rh,
                   &rd[i], &td[i], &ru[i], &tu[i]);
        }
    }
    displacement_matrix(p, lay[1].vp, lay[1].vs, &h);
    dw = 2.0 * M_PI * fsamp / nsamp;

    double t0 = 0.; /* greens.cpp:510-526; includes the half-space with d = -1 */
    for (int i = 1; i <= nlay; i++) {
        double v = (wave_type == 0) ? lay[i].vp : lay[i].vs, d = lay[i].h;
        t0 += d * sqrt(1. / (v * v) - p2);
    }

    for (int j = 0; j < nfreq; j++) {
        double w = dw * j, lgw = j ? log(w / wref) : 0;
        for (int i = 1; i <= nlay; i++) { /* phase matrix, greens.cpp:533-549 */
            double d = lay[i].h, vp = lay[i].vp, vs = lay[i].vs, qp = lay[i].qp, qs = lay[i].qs;
            cplx miwd = CX(0., -w * d);
            cplx vpc = cscale((1. + lgw / (M_PI * qp)) + CX(creal(ii) / (2. * qp), cimag(ii) / (2. * qp)), vp);
            cplx vsc = cscale((1. + lgw / (M_PI * qs)) + CX(creal(ii) / (2. * qs), cimag(ii) / (2. * qs)), vs);
            cplx plc = csqrt(rdivc(1., vpc * vpc) - p2);
            cplx slc = csqrt(rdivc(1., vsc * vsc) - p2);
            e[i].c11 = cexp(miwd * plc); e[i].c12 = 0; e[i].c21 = 0; e[i].c22 = cexp(miwd * slc);
        }
        /* top_down (greens.cpp:196-224), normal case */
        cmat2 q = cm_diag(0.);
        const cmat2 Id = cm_diag(1.);
        for (int i = 1; i < nlay; i++) {
            if (i == 1) nt[i] = ru[1];
            else nt[i] = cm_add(ru[i], cm_mul(cm_mul(td[i], nb[i - 1]), q));
            nb[i] = exe(e[i], nt[i]);
            q = cm_mul(cm_inv(cm_sub(Id, cm_mul(rd[i + 1], nb[i]))), tu[i + 1]);
            if (i == 1) g[i] = cm_mul(e[1], q);
            else g[i] = cm_mul(cm_mul(g[i - 1], e[i]), q);
        }
        t = cm_mul(cm_rscale(2, h), g[nlay - 1]); /* greens.cpp:572 */
        if (wave_type == 0) { cr[j] = t.c11; cz[j] = t.c21; } /* :576-578 */
        else                { cr[j] = t.c12; cz[j] = t.c22; } /* :579-581 */
        cplx qq = cexp(CX(0., w * t0));
        cr[j] *= qq;
        cz[j] *= qq;
    }

    compute_rf(wave_type, cr, cz, nsamp, fsamp, tshift, a, slowness, vp_top, vs_top, crf);
    iftr(nsamp, crf, rf, cx);
    if (rr != NULL && zz != NULL) iftr2(nsamp, cr, cz, rr, zz, cx, cx + nsamp, cx + 2 * nsamp);

    free(ru); free(rd); free(tu); free(td); free(nb); free(nt); free(g); free(e);
    free(cz); free(cr); free(crf); free(cx);
}

/* synrf_cwrap (wrap.cpp:57-80) -> synrf (synrf.cpp:16-55) */
int bho_synrf(int nsamp, double fsamp, double tshift, double p, double a, double nsv,
              double sigma, int waveno, int nlay, const double *z, const double *vp,
              const double *vs, const double *rh, const double *qp, const double *qs,
              double *fz, double *fr, double *rf)
{
    double vptop = nsv * sqrt((1. - (sigma)) / (.5 - (sigma))), vstop = nsv;
    double slowness = p * 0.00899; /* DEGREES_PER_KM, wrap.cpp:55,76 */
    layer_t *lay = calloc(nlay + 2, sizeof(layer_t));
    for (int i = 0; i < nlay - 1; i++) {
        layer_t l = { z[i], z[i + 1] - z[i], vp[i], vs[i], rh[i], qp[i], qs[i] };
        lay[i + 1] = l;
    }
    layer_t hs = { z[nlay - 1], -1, vp[nlay - 1], vs[nlay - 1], rh[nlay - 1], qp[nlay - 1], qs[nlay - 1] };
    lay[nlay] = hs;
    for (int i = 1; i <= nlay; i++) flatten(&lay[i]);
    calcresp(nlay, lay, waveno, slowness, 1., nsamp, fsamp, tshift, a, vptop, vstop, fz, fr, rf);
    free(lay);
    return 1;
}

void bho_rf_batch(int B, int Lmax, const int *nlay, const double *h, const double *vp,
                  const double *vs, const double *rho, double p, double gauss, int nsamp,
                  double fsamp, double tshift, double nsv_override, int waveno, int nout,
                  double *rf, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads)
    for (int b = 0; b < B; b++) {
        int n = nlay[b];
        double z[BHO_NL], qp[BHO_NL], qs[BHO_NL];
        const double *hb = h + (long)b * Lmax, *vpb = vp + (long)b * Lmax,
                     *vsb = vs + (long)b * Lmax, *rb = rho + (long)b * Lmax;
        double *tmp = malloc(sizeof(double) * nsamp);
        /* rfmini_modrf.py:119-130 */
        double acc = 0.0;
        for (int i = 0; i < n; i++) { z[i] = acc; acc += hb[i]; qp[i] = 500.; qs[i] = 225.; }
        double vpvs = vpb[0] / vsb[0];
        double poisson = (2 - vpvs * vpvs) / (2 - 2 * (vpvs * vpvs));
        double nsv = nsv_override > 0 ? nsv_override : vsb[0];
        bho_synrf(nsamp, fsamp, tshift, p, gauss, nsv, poisson, waveno, n, z, vpb, vsb, rb, qp, qs,
                  NULL, NULL, tmp);
        for (int i = 0; i < nout; i++) rf[(long)b * nout + i] = tmp[i];
        free(tmp);
    }
}
