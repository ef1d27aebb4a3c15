import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope='session')
def oracle():
    """The CPU oracle (plain-C restatement), compiled on demand."""
    from oracle import pyoracle
    pyoracle.port_lib()
    return pyoracle


@pytest.fixture(scope='session')
def lib():
    """The in-tree HIP library (built on demand; hipcc cross-compiles without a GPU)."""
    import bayhunter_amd
    bayhunter_amd.build()
    return bayhunter_amd.load()


def _build_hostsim(name, extra):
    d = os.path.join(ROOT, 'tests', 'hostsim')
    so = os.path.join(d, name)
    srcs = [os.path.join(d, 'hostsim.cpp')] + [
        os.path.join(ROOT, 'bayhunter_amd', 'csrc', f)
        for f in ('bh_common.h', 'bh_math.h', 'swd_core.h', 'swd_team.h', 'rf_core.h', 'rf_host.h')]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.run(['g++', '-O2', '-std=c++17', '-fPIC', '-shared', '-ffp-contract=off'] + extra +
                       ['-o', so, srcs[0]], check=True)
    return C.CDLL(so)


def _wrap_hostsim(hs):
    fp, dp = C.POINTER(C.c_float), C.POINTER(C.c_double)
    hs.hs_surfdisp96.restype = C.c_int
    hs.hs_surfdisp96.argtypes = [fp, fp, fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_int, dp, dp, C.POINTER(C.c_long)]
    hs.hs_surfdisp96_team.restype = C.c_int
    hs.hs_surfdisp96_team.argtypes = [fp, fp, fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int, dp, dp, C.c_int, C.POINTER(C.c_long),
                                      C.POINTER(C.c_long), C.POINTER(C.c_long)]
    hs.hs_surfdisp96_teamw.restype = C.c_int
    hs.hs_surfdisp96_teamw.argtypes = hs.hs_surfdisp96_team.argtypes
    hs.hs_rf.restype = C.c_int
    hs.hs_rf.argtypes = [C.c_int, dp, dp, dp, dp, dp, dp, C.c_double, C.c_double, C.c_int,
                         C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, dp]
    hs.hs_scan_cell.argtypes = [C.c_int, dp, C.POINTER(C.c_int), dp, dp]
    hs.hs_sincos.argtypes = [C.c_int, dp, dp, dp]
    hs.hs_exp.argtypes = [C.c_int, dp, dp]

    class HS(object):
        @staticmethod
        def swd(h, vp, vs, rho, per, iw, ig, mode=1, fl=0):
            f = [np.ascontiguousarray(np.asarray(x, dtype=np.float64).astype(np.float32))
                 for x in (h, vp, vs, rho)]
            t = np.ascontiguousarray(per, dtype=np.float64)
            cg = np.zeros(len(t))
            nc = C.c_long(0)
            e = hs.hs_surfdisp96(*[x.ctypes.data_as(fp) for x in f], len(h), fl, iw, mode, ig,
                                 len(t), t.ctypes.data_as(dp), cg.ctypes.data_as(dp), C.byref(nc))
            return cg, e, nc.value

        @staticmethod
        def swd_team(h, vp, vs, rho, per, iw, ig, mode=1, fl=0, nlanes=64, wide=False):
            f = [np.ascontiguousarray(np.asarray(x, dtype=np.float64).astype(np.float32))
                 for x in (h, vp, vs, rho)]
            t = np.ascontiguousarray(per, dtype=np.float64)
            cg = np.zeros(len(t))
            nc, ns, nr = C.c_long(0), C.c_long(0), C.c_long(0)
            fn = hs.hs_surfdisp96_teamw if wide else hs.hs_surfdisp96_team
            e = fn(*[x.ctypes.data_as(fp) for x in f], len(h), fl, iw, mode, ig,
                                      len(t), t.ctypes.data_as(dp), cg.ctypes.data_as(dp), nlanes,
                                      C.byref(nc), C.byref(ns), C.byref(nr))
            return cg, e, nc.value, ns.value, nr.value

        @staticmethod
        def rf(h, vp, vs, rho, p=6.4, gauss=1.0, nsamp=512, fsamp=5.0, tshift=5.0, nsv=None,
               waveno=0, nout=201, qp=None, qs=None):
            a = [np.ascontiguousarray(x, dtype=np.float64) for x in (h, vp, vs, rho)]
            q = [None if x is None else np.ascontiguousarray(x, dtype=np.float64) for x in (qp, qs)]
            out = np.zeros(nout)
            hs.hs_rf(len(a[0]), *[x.ctypes.data_as(dp) for x in a],
                     *[None if x is None else x.ctypes.data_as(dp) for x in q], p, gauss, nsamp,
                     fsamp, tshift, -1.0 if nsv is None else nsv, waveno, nout,
                     out.ctypes.data_as(dp))
            return out

        @staticmethod
        def scan_cell(base, cell):
            base = np.ascontiguousarray(base, dtype=np.float64)
            cell = np.ascontiguousarray(cell, dtype=np.int32)
            b, cn = np.zeros_like(base), np.zeros_like(base)
            hs.hs_scan_cell(base.size, base.ctypes.data_as(dp), cell.ctypes.data_as(C.POINTER(C.c_int)),
                            b.ctypes.data_as(dp), cn.ctypes.data_as(dp))
            return b, cn

        @staticmethod
        def sincos(x):
            x = np.ascontiguousarray(x, dtype=np.float64)
            s_, c_ = np.zeros_like(x), np.zeros_like(x)
            hs.hs_sincos(x.size, x.ctypes.data_as(dp), s_.ctypes.data_as(dp), c_.ctypes.data_as(dp))
            return s_, c_

        @staticmethod
        def exp(x):
            x = np.ascontiguousarray(x, dtype=np.float64)
            y = np.zeros_like(x)
            hs.hs_exp(x.size, x.ctypes.data_as(dp), y.ctypes.data_as(dp))
            return y
    return HS


@pytest.fixture(scope='session')
def hostsim():
    """g++ build of the device solver cores with glibc math: the replay must be bit-identical to
    the oracle (tests only)."""
    return _wrap_hostsim(_build_hostsim('libhostsim.so', ['-DBH_HOSTSIM_GLIBC_MATH']))


@pytest.fixture(scope='session')
def hostsim_devmath():
    """Same cores with the device math of bh_math.h (software FMA through libm when the host
    CPU build has none): what the GPU computes, up to ocml's sqrt/log (tests only)."""
    fma = ['-mfma'] if ' fma ' in open('/proc/cpuinfo').read() else []
    return _wrap_hostsim(_build_hostsim('libhostsim_devmath.so', fma))


@pytest.fixture(scope='session')
def golden():
    return {name: np.load(os.path.join(GOLDEN, name + '.npz'))
            for name in ('swd_rf_random', 'swd_variants', 'rf_variants', 'tutorial_full', 'swd_water')}


@pytest.fixture(scope='session')
def golden_chains():
    return np.load(os.path.join(GOLDEN, 'chains_golden.npz'))


REFS = [('rdispph', 2, 0), ('rdispgr', 2, 1), ('ldispph', 1, 0), ('ldispgr', 1, 1)]
SETS = ['L%d_%s' % (L, s) for L in (2, 5, 10, 15, 31) for s in ('sorted', 'lvz')]
