// bh_math.h -- lean fp64 sincos / exp for the solver cores.
//
// The reference calls glibc's sin, cos, exp (through dsin/dcos/dexp and std::complex).  No device
// libm reproduces those bit for bit, so the only requirement here is accuracy (< 1 ulp, measured
// against glibc in tests/test_hostsim.py::test_math_accuracy) -- and a short instruction stream:
// the period equation spends half of its instructions in these three functions, and a wave whose
// lanes straddle the oscillatory/evanescent branch of surfdisp96.f:929-968 pays for both sides.
//
//   bh_sincos : one-step Cody-Waite reduction by pi/2 (fdlibm's 33+53-bit split, exact with FMA)
//               valid for |x| < 1e4 (max error 1.04 ulp, 0.79 below 300), then the classic __kernel_sin/__kernel_cos minimax
//               polynomials on [-pi/4, pi/4] with the reduction tail folded in.  Larger arguments
//               (never produced by physical models: p = k_z * d) take an exact-product reduction
//               with fdlibm's three 33-bit pieces of pi/2 up to 1e12 and are NaN beyond (a phase
//               of 1e12 rad is no seismological model; the search then reports "no root").  The
//               device library's sincos is deliberately not used: inlined into the recursions its
//               Payne-Hanek path cost registers in every kernel (spills in rf_kernel and the team
//               kernels) although it never ran.
//   bh_exp    : k = rint(x/ln2), r = x - k*ln2 (hi/lo, exact with FMA), degree-13 polynomial,
//               scaled by 2^k with ldexp; saturates like exp() outside [-745, 709].
#pragma once
#include "bh_common.h"

namespace bh {

// bh_fma_k(a, b, K) = fma(a, b, K) for a compile-time constant K: a Horner step.  The compiler keeps
// polynomial coefficients in VGPR pairs and, the kernels being at their register limit, copies each
// into the accumulator before a two-address v_fmac_f64 (a v_mov_b64 per step: 50 of the ~450 vector
// instructions of a layer step of swd_kernel).  The three-address form with the coefficient as a scalar
// operand needs neither the copy nor the VGPRs.  An "s" operand is read from one lane, so the scalar form
// is only taken when K is a compile-time constant after inlining (__builtin_constant_p, resolved late
// through llvm.is.constant); anything else gets the plain builtin.  (-DBH_NO_FMA_K: always the plain
// builtin, for A/B measurements and the bit-parity matrix.)
#if defined(BH_HOSTSIM)
BH_DEV double bh_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
BH_DEV double bh_fma_k(double a, double b, double k) { return __builtin_fma(a, b, k); }
BH_DEV double bh_rint(double x) { return __builtin_rint(x); }
BH_DEV double bh_ldexp(double x, int k) { return __builtin_ldexp(x, k); }
#else
BH_DEV double bh_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
#if defined(BH_NO_FMA_K)
BH_DEV double bh_fma_k(double a, double b, double k) { return __builtin_fma(a, b, k); }
#else
BH_DEV double bh_fma_k(double a, double b, double k)
{
    if (!__builtin_constant_p(k)) return __builtin_fma(a, b, k);
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(k));
    return r;
}
#endif
BH_DEV double bh_rint(double x) { return __builtin_rint(x); }
BH_DEV double bh_ldexp(double x, int k) { return __builtin_amdgcn_ldexp(x, k); }
#endif

// x if two == 0, -x if two == 2: the sign bit flipped by an integer xor (three instructions instead of
// compare + select + xor; -x is exactly a flip of the sign bit, for zeros and NaNs too)
#if defined(BH_HOSTSIM)
BH_DEV double bh_negate_if2(double x, int two) { return two ? -x : x; }
#else
BH_DEV double bh_negate_if2(double x, int two)
{
    return __hiloint2double(__double2hiint(x) ^ (two << 30), __double2loint(x));
}
#endif

#if defined(BH_HOSTSIM) && defined(BH_HOSTSIM_GLIBC_MATH)
// tests/hostsim "exact" build: glibc's functions, so that the replay is bit-identical to the oracle
BH_DEV void bh_sincos(double x, double *sn, double *cs) { *sn = std::sin(x); *cs = std::cos(x); }
BH_DEV double bh_exp(double x) { return std::exp(x); }
BH_DEV double bh_exp_bounded(double x) { return std::exp(x); }
#else
BH_DEV void bh_sincos(double x, double *sn, double *cs)
{
    const double INVPIO2 = 6.36619772367581382433e-01;  // 2/pi
    const double PIO2_1 = 1.57079632673412561417e+00;   // first 33 bits of pi/2
    const double PIO2_1T = 6.07710050650619224932e-11;  // pi/2 - PIO2_1
    double y0, y1;
    int n;
    if (fabs(x) < 1.0e4) {     // beyond 1e4 the two-constant reduction exceeds 1 ulp
        double fn = bh_rint(x * INVPIO2);
        double r = bh_fma(-fn, PIO2_1, x);  // exact: fn < 2^20, PIO2_1 has 33 significant bits
        double w = fn * PIO2_1T;
        y0 = r - w;
        y1 = (r - y0) - w;                  // tail: x - fn*pi/2 = y0 + y1
        n = (int)fn;
    } else if (fabs(x) < 1.0e12) {
        // fn < 2^40 splits into fh (a multiple of 2^20, <= 20 significant bits) + fl (|fl| <= 2^19):
        // every product with a 33-bit piece of pi/2 is exact, x - fh*PIO2_1 is exact (Sterbenz), and
        // the remaining terms are summed in double-double (two-sum).  152 bits of pi/2 in total.
        const double PIO2_2 = 6.07710050630396597660e-11, PIO2_3 = 2.02226624871116645580e-21,
                     PIO2_3T = 8.47842766036889956997e-32;
        double fn = bh_rint(x * INVPIO2);
        double fh = bh_rint(fn * 9.5367431640625e-07) * 1048576.0, fl = fn - fh;
        double hi = x - fh * PIO2_1, lo = 0.0;
        const double terms[5] = {fl * PIO2_1, fh * PIO2_2, fl * PIO2_2, fh * PIO2_3, fl * PIO2_3};
        for (int k = 0; k < 5; k++) {
            double t = -terms[k];
            double s = hi + t, bb = s - hi;
            lo += (hi - (s - bb)) + (t - bb);
            hi = s;
        }
        lo -= fn * PIO2_3T;
        y0 = hi + lo;
        y1 = (hi - y0) + lo;
        n = (int)(fn - 4.0 * bh_rint(fn * 0.25));   // quadrant; fn itself does not fit an int
    } else {                                        // 1e12 and beyond, Inf, NaN
        *sn = *cs = __builtin_nan("");
        return;
    }
    double z = y0 * y0;
    // __kernel_sin(y0, y1, 1)
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    double v = z * y0;
    double rs = bh_fma_k(z, bh_fma_k(z, bh_fma_k(z, bh_fma(z, S6, S5), S4), S3), S2);
    double s = y0 - ((z * (0.5 * y1 - v * rs) - y1) - v * S1);
    // __kernel_cos(y0, y1)
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double ww = z * z;
    double rc = z * bh_fma_k(z, bh_fma(z, C3, C2), C1) + (ww * ww) * bh_fma_k(z, bh_fma(z, C6, C5), C4);
    double hz = 0.5 * z;
    double w1 = 1.0 - hz;
    double c = w1 + (((1.0 - w1) - hz) + (z * rc - y0 * y1));
    // quadrant
    double so = (n & 1) ? c : s;
    double co = (n & 1) ? s : c;
    *sn = bh_negate_if2(so, n & 2);
    *cs = bh_negate_if2(co, (n + 1) & 2);
}

BH_DEV double bh_exp(double x)
{
    const double LOG2E = 1.44269504088896338700e+00;
    const double LN2_HI = 6.93147180369123816490e-01;  // 32 significant bits
    const double LN2_LO = 1.90821492927058770002e-10;
    double xc = x;
    if (xc > 710.0) xc = 710.0;      // -> +inf through ldexp
    if (xc < -746.0) xc = -746.0;    // -> 0
    double k = bh_rint(xc * LOG2E);
    double r = bh_fma(-k, LN2_HI, xc);
    r = bh_fma(-k, LN2_LO, r);
    // exp(r) on |r| <= ln2/2: Taylor to degree 13 (truncation 4e-18)
    double p = 1.6059043836821613e-10;                  // 1/13!
    p = bh_fma_k(p, r, 2.08767569878681e-09);             // 1/12!
    p = bh_fma_k(p, r, 2.505210838544172e-08);            // 1/11!
    p = bh_fma_k(p, r, 2.755731922398589e-07);            // 1/10!
    p = bh_fma_k(p, r, 2.7557319223985893e-06);           // 1/9!
    p = bh_fma_k(p, r, 2.48015873015873e-05);             // 1/8!
    p = bh_fma_k(p, r, 1.984126984126984e-04);            // 1/7!
    p = bh_fma_k(p, r, 1.3888888888888889e-03);           // 1/6!
    p = bh_fma_k(p, r, 8.333333333333333e-03);            // 1/5!
    p = bh_fma_k(p, r, 4.1666666666666664e-02);           // 1/4!
    p = bh_fma_k(p, r, 1.6666666666666666e-01);           // 1/3!
    p = bh_fma(p, r, 0.5);
    p = bh_fma(p, r, 1.0);
    p = bh_fma(p, r, 1.0);
    double res = bh_ldexp(p, (int)k);
    return (x != x) ? x : res;
}

// exp for arguments known to lie in [-700, 700] (the period equation only asks for exp(-2p), p < 16,
// and exp(-exa), exa < 60): same reduction and polynomial as bh_exp without the range clamps.
BH_DEV double bh_exp_bounded(double x)
{
    const double LOG2E = 1.44269504088896338700e+00;
    const double LN2_HI = 6.93147180369123816490e-01;
    const double LN2_LO = 1.90821492927058770002e-10;
    double k = bh_rint(x * LOG2E);
    double r = bh_fma(-k, LN2_HI, x);
    r = bh_fma(-k, LN2_LO, r);
    double p = 1.6059043836821613e-10;
    p = bh_fma_k(p, r, 2.08767569878681e-09);
    p = bh_fma_k(p, r, 2.505210838544172e-08);
    p = bh_fma_k(p, r, 2.755731922398589e-07);
    p = bh_fma_k(p, r, 2.7557319223985893e-06);
    p = bh_fma_k(p, r, 2.48015873015873e-05);
    p = bh_fma_k(p, r, 1.984126984126984e-04);
    p = bh_fma_k(p, r, 1.3888888888888889e-03);
    p = bh_fma_k(p, r, 8.333333333333333e-03);
    p = bh_fma_k(p, r, 4.1666666666666664e-02);
    p = bh_fma_k(p, r, 1.6666666666666666e-01);
    p = bh_fma(p, r, 0.5);
    p = bh_fma(p, r, 1.0);
    p = bh_fma(p, r, 1.0);
    return bh_ldexp(p, (int)k);
}
#endif

// exp(z) = e^re (cos im + i sin im), the formulation of glibc's cexp for finite arguments
BH_DEV cd cexp_(cd z)
{
    double s, c, e = bh_exp(z.re);
    bh_sincos(z.im, &s, &c);
    return mk(e * c, e * s);
}


// The same without bh_exp's range clamps: the exponent of a phase factor of the receiver-function recursion
// is w d Im(slowness) -- attenuation, or the decay of an evanescent wave: the reduction handles any magnitude
// a layered model can produce (ldexp saturates to 0 / inf, NaN propagates through the polynomial).
BH_DEV cd cexp_bounded(cd z)
{
    double s, c, e = bh_exp_bounded(z.re);
    bh_sincos(z.im, &s, &c);
    return mk(e * c, e * s);
}

}  // namespace bh
