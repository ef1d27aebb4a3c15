/* ulp_noise_libm.c -- TEST INFRASTRUCTURE (tests/scenarios/lvz_worst_cases.py, tests/test_oracle.py).
 *
 * An LD_PRELOAD shim that answers sin / cos / sincos / exp with glibc's own result moved by -1, 0 or
 * +1 ulp -- a fixed function of (argument, BH_ULP_NOISE_SEED), so within a process each routine stays
 * a function.  The reference's native code (oracle/_ref) and the C restatement call exactly these libm
 * symbols (nm -D: exp, sincos): run under this shim they show how far the REFERENCE's own results move
 * when its transcendental functions are accurate to 1 ulp instead of glibc's 0.5x -- which is precisely
 * what separates the device math (bh_math.h, < 1 ulp) from glibc.  Nothing is replaced in the
 * reference: same binary, same control flow, perturbed last bits of three libm functions.
 *
 *   gcc -O2 -fPIC -shared -o libulpnoise.so ulp_noise_libm.c -ldl -lm
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static double (*real_sin)(double), (*real_cos)(double), (*real_exp)(double);
static uint64_t seed;
static int ready;

static void init(void)
{
    const char *e = getenv("BH_ULP_NOISE_SEED");
    real_sin = (double (*)(double))dlsym(RTLD_NEXT, "sin");
    real_cos = (double (*)(double))dlsym(RTLD_NEXT, "cos");
    real_exp = (double (*)(double))dlsym(RTLD_NEXT, "exp");
    seed = e ? strtoull(e, 0, 10) : 0;
    ready = 1;
}

static double nudge(double y, double x, uint64_t salt)
{
    uint64_t h;
    int d;
    if (!isfinite(y) || y == 0.0) return y;
    memcpy(&h, &x, 8);
    h ^= seed * 0x9E3779B97F4A7C15ull + salt;
    h ^= h >> 33; h *= 0xff51afd7ed558ccdull; h ^= h >> 33; h *= 0xc4ceb9fe1a85ec53ull; h ^= h >> 33;
    d = (int)(h % 3) - 1;
    return d == 0 ? y : nextafter(y, d > 0 ? INFINITY : -INFINITY);
}

double sin(double x) { if (!ready) init(); return nudge(real_sin(x), x, 1); }
double cos(double x) { if (!ready) init(); return nudge(real_cos(x), x, 2); }
double exp(double x) { if (!ready) init(); return nudge(real_exp(x), x, 3); }
void sincos(double x, double *s, double *c)
{
    if (!ready) init();
    *s = nudge(real_sin(x), x, 1);
    *c = nudge(real_cos(x), x, 2);
}
