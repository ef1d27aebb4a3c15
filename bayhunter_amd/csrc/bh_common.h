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

BH_DEV double dsign1(double x) { return copysign(1.0, x); }
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
// 1/z = conj(z)/|z|^2: one real division instead of Smith's three; for the well-scaled values of
// the per-frequency recursion (|z| within 1e-3..1e3)
BH_DEV cd crecip(cd z)
{
    double r = 1.0 / (z.re * z.re + z.im * z.im);
    return mk(z.re * r, -(z.im * r));
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

#if !defined(BH_HOSTSIM)
#pragma clang fp contract(off)
#endif

}  // namespace bh
