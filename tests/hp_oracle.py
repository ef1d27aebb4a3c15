"""TEST ONLY.  The receiver-function oracle evaluated in extended precision (x87 long double, 64-bit mantissa): the same
C restatement (oracle/oracle_rf.c) with every `double` turned into `long double` and the libm calls into their `l`
forms, compiled into the temporary directory.  It tells how far the ORACLE itself is from the exact result of the
reference's algorithm on a given model -- the yardstick for the rare ill-conditioned model of the random campaigns
(tests/rf_extreme.py), where fp64 evaluations of the same formulas differ from each other by ~1e-10."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_lib = None


def _build():
    src = open(os.path.join(ROOT, 'oracle', 'oracle_rf.c')).read()
    src = src.replace('#include "oracle_port.h"', '#define BHO_NL 100\n#define BHO_NP 60\n')
    src = re.sub(r'\bdouble\b', 'long double', src)
    for f in ('cexp', 'csqrt', 'cimag', 'creal', 'conj', 'fabs', 'log', 'sqrt', 'exp', 'sin', 'cos', 'floor', 'cabs'):
        src = re.sub(r'\b%s\(' % f, f + 'l(', src)
    src = re.sub(r'\bM_PI\b', 'M_PIl', src)
    src = re.sub(r'\bCMPLX\(', 'CMPLXL(', src)
    src = re.sub(r'\bbho_', 'bhol_', src)
    # floating constants stay double literals (0.00899, 1., 500.): exactly the reference's constants, widened
    d = tempfile.mkdtemp(prefix='bh_hp_oracle_')
    c, so = os.path.join(d, 'oracle_rf_ld.c'), os.path.join(d, 'liboracle_rf_ld.so')
    open(c, 'w').write('#define _GNU_SOURCE\n' + src)
    subprocess.run(['gcc', '-O2', '-std=gnu11', '-fPIC', '-shared', '-fopenmp', '-ffp-contract=off', '-o', so, c, '-lm'], check=True)
    return C.CDLL(so)


def rf_model_ld(h, vp, vs, rho, p, gauss, nsamp, fsamp, tshift, nsv, waveno, nout):
    """One model through the long-double oracle; returns the first nout samples as np.longdouble."""
    global _lib
    if _lib is None:
        _lib = _build()
    ld = np.longdouble
    a = [np.ascontiguousarray(x, dtype=ld)[None, :] for x in (h, vp, vs, rho)]
    n = np.array([a[0].shape[1]], dtype=np.int32)
    out = np.zeros((1, nout), dtype=ld)
    ptr = lambda x: x.ctypes.data_as(C.c_void_p)
    f = _lib.bhol_rf_batch
    f.restype = None
    f.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longdouble,
                  C.c_longdouble, C.c_int, C.c_longdouble, C.c_longdouble, C.c_longdouble, C.c_int, C.c_int, C.c_void_p, C.c_int]
    f(1, a[0].shape[1], ptr(n), ptr(a[0]), ptr(a[1]), ptr(a[2]), ptr(a[3]), ld(p), ld(gauss), int(nsamp), ld(fsamp),
      ld(tshift), ld(-1.0 if nsv is None else nsv), int(waveno), int(nout), ptr(out), 1)
    return out[0]
