// bh_common.h -- shared definitions for the HIP kernels of libbayhunter_amd.
//
// The per-lane solver cores (swd_core.h, rf_core.h) are plain C++ functions marked BH_DEV.  hipcc
// compiles them for gfx950; tests/hostsim/ compiles the very same headers with g++ (BH_HOSTSIM) to
// check the control-flow transformation against the oracle bit for bit on the CPU.  The host build
// is test infrastructure only: nothing in the shipped library or Python package uses it.
#pragma once

#if defined(BH_HOSTSIM)
#include <cmath>
#define BH_DEV static inline
#define BH_HD static inline
#define BH_RESTRICT __restrict__
#else
#include <hip/hip_runtime.h>
#define BH_DEV __device__ __forceinline__
#define BH_HD __host__ __device__ __forceinline__
#define BH_RESTRICT __restrict__
#endif

namespace bh {

#if defined(BH_HOSTSIM)
using std::copysign;
using std::cos;
using std::exp;
using std::fabs;
using std::log;
using std::sin;
using std::sqrt;
#else
#endif

// ---- IEEE division, shared-divisor form ------------------------------------------------------------
// hipcc expands `a / b` (fp64) to v_div_scale x2, v_rcp, two Newton steps on the reciprocal, q0 = a*r,
// rem = fma(-b, q0, a), v_div_fmas, v_div_fixup.  The scale / fix-up instructions only act on
// denormal, overflowing or non-finite operands; for the normal-range operands of the period
// equation the quotient is exactly q1 = fma(rem, r, q0).  Recip keeps the refined reciprocal so that
// the quotients by one divisor (three by rho, five by the normalisation factor of normc) share it:
// 5 + 3 instructions per quotient instead of 11.  Bit-identical to `/` there
// (bh_selftest / tests/test_gpu_parity.py::test_division_selftest); a zero numerator yields +0 where
// IEEE gives -0 for -0/b, non-finite operands may yield NaN where IEEE gives Inf/0.
// The host replay uses the plain operator.
#if defined(BH_HOSTSIM)
struct Recip { double b; };
BH_DEV Recip recip_of(double b) { Recip R; R.b = b; return R; }
BH_DEV double qdiv(double a, const Recip &R) { return a / R.b; }
#else
struct Recip { double b, r; };
BH_DEV Recip recip_of(double b)
{
    Recip R;
    R.b = b;
    double r = __builtin_amdgcn_rcp(b);
    r = __builtin_fma(r, __builtin_fma(-b, r, 1.0), r);
    r = __builtin_fma(r, __builtin_fma(-b, r, 1.0), r);
    R.r = r;
    return R;
}
BH_DEV double qdiv(double a, const Recip &R)
{
    const double q0 = a * R.r;
    return __builtin_fma(__builtin_fma(-R.b, q0, a), R.r, q0);
}
#endif
BH_DEV double xdiv(double a, double b) { return qdiv(a, recip_of(b)); }

// IEEE square root for arguments that are zero or in the normal range (no denormals, no Inf): the
// compiler's own sequence (v_rsq, one coupled Newton step, two residual corrections) without its
// 2^256 range scaling.  Bit-identical to sqrt() there (bh_selftest_division).
#if defined(BH_HOSTSIM)
BH_DEV double xsqrt(double x) { return sqrt(x); }
#else
BH_DEV double xsqrt(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    g = __builtin_fma(__builtin_fma(-g, g, x), h, g);
    g = __builtin_fma(__builtin_fma(-g, g, x), h, g);
    return (x == 0.0) ? x : g;
}
#endif

BH_DEV double dsign1(double x) { return copysign(1.0, x); }
// dsign(1, a) != dsign(1, b), equivalently dsign(1, a) * dsign(1, b) < 0 (the reference's sign tests,
// surfdisp96.f:431,461,600,616): the sign bits differ.  One xor instead of two copysigns and a product;
// the same truth value for every pair of operands, zeros and NaNs included.
#if defined(BH_HOSTSIM)
BH_DEV bool bh_signs_differ(double a, double b) { return std::signbit(a) != std::signbit(b); }
#else
BH_DEV bool bh_signs_differ(double a, double b) { return (__double2hiint(a) ^ __double2hiint(b)) < 0; }
#endif
BH_DEV double bh_fmax(double a, double b) { return __builtin_fmax(a, b); }
BH_DEV double dmin(double a, double b) { return a < b ? a : b; }
BH_DEV double dmax(double a, double b) { return a > b ? a : b; }

// ---- complex fp64 --------------------------------------------------------------------------
// Complex arithmetic is only used by the receiver-function path, which is tolerance-checked
// (|diff| <= 1e-10, observed 1e-13): it may use FMA contraction.  The surface-wave path (real
// arithmetic, exact replay) is compiled with contraction off.
#if !defined(BH_HOSTSIM)
#pragma clang fp contract(fast)
#endif
struct cd {
    double re, im;
};
BH_DEV cd mk(double re, double im) { cd r; r.re = re; r.im = im; return r; }
BH_DEV cd operator+(cd a, cd b) { return mk(a.re + b.re, a.im + b.im); }
BH_DEV cd operator-(cd a, cd b) { return mk(a.re - b.re, a.im - b.im); }
BH_DEV cd operator-(cd a) { return mk(-a.re, -a.im); }
BH_DEV cd operator*(cd a, cd b) { return mk(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
BH_DEV cd operator*(cd a, double r) { return mk(a.re * r, a.im * r); }
BH_DEV cd operator*(double r, cd a) { return mk(a.re * r, a.im * r); }
BH_DEV cd operator/(cd a, double r) { return mk(a.re / r, a.im / r); }
BH_DEV cd operator+(double r, cd a) { return mk(a.re + r, a.im); }
BH_DEV cd operator+(cd a, double r) { return mk(a.re + r, a.im); }
BH_DEV cd operator-(cd a, double r) { return mk(a.re - r, a.im); }
BH_DEV cd conj(cd a) { return mk(a.re, -a.im); }

// complex / complex: Smith's algorithm as in libgcc's __divdc3 (the reference's std::complex
// division lowers to it); the NaN/Inf recovery tail of __divdc3 is not needed here.
BH_DEV cd operator/(cd x, cd y)
{
    double a = x.re, b = x.im, c = y.re, d = y.im, ratio, denom;
    if (fabs(c) < fabs(d)) {
        ratio = c / d;
        denom = (c * ratio) + d;
        return mk(((a * ratio) + b) / denom, ((b * ratio) - a) / denom);
    }
    ratio = d / c;
    denom = (d * ratio) + c;
    return mk(((b * ratio) + a) / denom, (b - (a * ratio)) / denom);
}
BH_DEV cd rdiv(double x, cd y) { return mk(x, 0.0) / y; }
// Fast reciprocal / square root for the tolerance-checked receiver-function recursion: hardware
// seed (v_rcp_f64 / v_rsq_f64) + Newton steps, ~1 ulp, half the instructions of the IEEE sequences.
// Arguments there are well scaled (1e-6..1e6) and non-zero.
#if defined(BH_HOSTSIM)
BH_DEV double frcp(double x) { return 1.0 / x; }
BH_DEV double fsqrt(double x) { return sqrt(x); }
#else
BH_DEV double frcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    return r;
}
BH_DEV double fsqrt(double x)
{
    double y = __builtin_amdgcn_rsq(x);          // ~ 1/sqrt(x)
    double g = x * y, h = 0.5 * y;
    double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    return (x > 0.0) ? g : ((x == 0.0) ? 0.0 : __builtin_nan(""));
}
#endif

// 1/z = conj(z)/|z|^2: one real reciprocal instead of Smith's three divisions
BH_DEV cd crecip(cd z)
{
    double r = frcp(z.re * z.re + z.im * z.im);
    return mk(z.re * r, -(z.im * r));
}

// sqrt(x) and 1/(2 sqrt(x)) from one v_rsq_f64 seed (the coupled Newton iteration of fsqrt carries both)
#if defined(BH_HOSTSIM)
BH_DEV void fsqrt_hinv(double x, double *g, double *h) { *g = sqrt(x); *h = 0.5 / *g; }
#else
BH_DEV void fsqrt_hinv(double x, double *gout, double *hout)
{
    double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    r = __builtin_fma(-h, g, 0.5);                // one more step on h: ~1 ulp like frcp
    h = __builtin_fma(h, r, h);
    *gout = g;
    *hout = h;
}
#endif

// principal square root for generic complex arguments of the recursion (im != 0): with m = the larger of
// the two roots sqrt((|z| +- re)/2), the other one is im / (2 m)
BH_DEV cd csqrt_fast(cd z)
{
    double re = z.re, im = z.im;
    double d = fsqrt(re * re + im * im), m, hinv;
    fsqrt_hinv(0.5 * (d + fabs(re)), &m, &hinv);
    const double o = im * hinv;                   // im / (2 m)
    return (re > 0) ? mk(m, o) : mk(fabs(o), copysign(m, im));
}

// principal square root, glibc csqrt's formulation for finite non-zero arguments
BH_DEV cd csqrt_(cd z)
{
    double re = z.re, im = z.im;
    if (im == 0.0) {
        if (re < 0.0) return mk(0.0, copysign(sqrt(-re), im));
        return mk(fabs(sqrt(re)), copysign(0.0, im));
    }
    if (re == 0.0) {
        double r = sqrt(0.5 * fabs(im));
        return mk(r, copysign(r, im));
    }
    double d = sqrt(re * re + im * im), r, s;
    if (re > 0) {
        r = sqrt(0.5 * (d + re));
        s = 0.5 * (im / r);
    } else {
        s = sqrt(0.5 * (d - re));
        r = fabs(0.5 * (im / s));
    }
    return mk(r, copysign(s, im));
}


// ---- complex 2x2 (rfmini cmat2.h) ------------------------------------------------------------
struct cm2 {
    cd c11, c12, c21, c22;
};
BH_DEV cm2 operator*(const cm2 &x, const cm2 &y)
{
    cm2 r;
    r.c11 = x.c11 * y.c11 + x.c12 * y.c21;
    r.c12 = x.c11 * y.c12 + x.c12 * y.c22;
    r.c21 = x.c21 * y.c11 + x.c22 * y.c21;
    r.c22 = x.c21 * y.c12 + x.c22 * y.c22;
    return r;
}
BH_DEV cm2 operator+(const cm2 &x, const cm2 &y)
{
    cm2 r;
    r.c11 = x.c11 + y.c11; r.c12 = x.c12 + y.c12; r.c21 = x.c21 + y.c21; r.c22 = x.c22 + y.c22;
    return r;
}

// real 2x2 times / plus complex 2x2: the interface coefficient matrices of a model are real when no
// wave is post-critical anywhere in the stack (the teleseismic case), and a complex product with a
// zero imaginary part is the real one
struct rm2 {
    double c11, c12, c21, c22;
};
BH_DEV cm2 operator*(const rm2 &x, const cm2 &y)
{
    cm2 r;
    r.c11 = y.c11 * x.c11 + y.c21 * x.c12;
    r.c12 = y.c12 * x.c11 + y.c22 * x.c12;
    r.c21 = y.c11 * x.c21 + y.c21 * x.c22;
    r.c22 = y.c12 * x.c21 + y.c22 * x.c22;
    return r;
}
BH_DEV cm2 operator*(const cm2 &x, const rm2 &y)
{
    cm2 r;
    r.c11 = x.c11 * y.c11 + x.c12 * y.c21;
    r.c12 = x.c11 * y.c12 + x.c12 * y.c22;
    r.c21 = x.c21 * y.c11 + x.c22 * y.c21;
    r.c22 = x.c21 * y.c12 + x.c22 * y.c22;
    return r;
}
BH_DEV cm2 operator+(const rm2 &x, const cm2 &y)
{
    cm2 r;
    r.c11 = mk(x.c11 + y.c11.re, y.c11.im); r.c12 = mk(x.c12 + y.c12.re, y.c12.im);
    r.c21 = mk(x.c21 + y.c21.re, y.c21.im); r.c22 = mk(x.c22 + y.c22.re, y.c22.im);
    return r;
}

#if !defined(BH_HOSTSIM)
#pragma clang fp contract(off)
#endif

}  // namespace bh
