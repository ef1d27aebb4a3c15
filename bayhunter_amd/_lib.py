"""ctypes binding of libbayhunter_amd.so (include/bayhunter_amd.h).

There is no CPU fallback: if the HIP library is missing, or no gfx950 device is usable, every
compute entry point raises.  `build()` compiles the library in-tree with hipcc (cross-compiles
without a GPU).
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libbayhunter_amd.so")
SOURCES = ["kernels.hip", "like_kernel.hip", "capi.hip", "evalplan.hip", "chains.cpp"]
HEADERS = ["bh_common.h", "bh_math.h", "swd_core.h", "swd_team.h", "rf_core.h", "rf_host.h", "kernels.h",
           "swd_form_table.h"]
# -disable-machine-licm (device code only): the kernels are register-bound, and constants hoisted out
# of the persistent loops (polynomial coefficients, masks) end up in VGPR pairs or spilled SGPRs and are
# copied back at every use; rematerialised next to their use they are scalar moves.  swd_kernel 254 ->
# 189 VGPRs, 451 -> 413 vector instructions per layer step, 55.2 -> 52.4 ms.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
               "-Xarch_device", "-mllvm=-disable-machine-licm"]

BH_OK, BH_ERR_ARG, BH_ERR_HIP, BH_ERR_NO_DEVICE, BH_ERR_WORKSPACE = 0, 1, 2, 3, 4
MAX_LAYERS, MAX_PERIODS, MAX_TARGETS = 100, 60, 16


class SwdTarget(C.Structure):
    """struct bh_swd_target"""
    _fields_ = [("iwave", C.c_int), ("igr", C.c_int), ("mode", C.c_int), ("iflsph", C.c_int),
                ("nper", C.c_int), ("per_off", C.c_int), ("out_off", C.c_int), ("_pad", C.c_int)]


class RfParams(C.Structure):
    """struct bh_rf_params"""
    _fields_ = [("p", C.c_double), ("gauss", C.c_double), ("fsamp", C.c_double),
                ("tshift", C.c_double), ("nsv", C.c_double), ("nsamp", C.c_int),
                ("waveno", C.c_int), ("nout", C.c_int), ("out_off", C.c_int)]


class LikeTarget(C.Structure):
    """struct bh_like_target"""
    _fields_ = [("n", C.c_int), ("off", C.c_int), ("cov", C.c_int), ("aux_off", C.c_int),
                ("logdet_extra", C.c_double)]


class EvalInterp(C.Structure):
    """struct bh_eval_interp"""
    _fields_ = [("target", C.c_int), ("dst_off", C.c_int), ("n_dst", C.c_int), ("_pad", C.c_int),
                ("obsx", C.c_void_p)]


class ModelPriors(C.Structure):
    """struct bh_model_priors"""
    _fields_ = [("layers_min", C.c_int), ("layers_max", C.c_int), ("vs_min", C.c_double),
                ("vs_max", C.c_double), ("z_min", C.c_double), ("z_max", C.c_double),
                ("thickmin", C.c_double), ("lowvelperc", C.c_double), ("highvelperc", C.c_double),
                ("mantle_vs", C.c_double), ("mantle_vpvs", C.c_double)]


COV_NOCORR, COV_NOCORR_SCALED, COV_EXP, COV_GAUSS = 0, 1, 2, 3


class BayHunterAmdError(RuntimeError):
    pass


def source_hash():
    """Hash of everything the library is built from (sources, headers, the C ABI header, the
    compiler flags): compiled into the library (bh_version) and stamped on profiles, so that a
    number can be tied to the code that produced it."""
    import hashlib
    h = hashlib.sha256()
    extra = os.environ.get("BH_EXTRA_HIPCC_FLAGS", "")
    h.update((" ".join(HIPCC_FLAGS) + "|" + extra).encode())
    for f in [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.join(_HERE, "..", "include", "bayhunter_amd.h")]:
        with open(f, "rb") as fh:
            h.update(os.path.basename(f).encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def built_hash():
    """Source hash recorded when libbayhunter_amd.so was last built here ('' if unknown)."""
    try:
        with open(LIB_PATH + ".srchash") as fh:
            return fh.read().strip()
    except (IOError, OSError):
        return ""


def needs_build():
    """By content, not by time stamp: a library pushed from another tree is rebuilt when its
    recorded source hash is not this tree's."""
    return not os.path.exists(LIB_PATH) or built_hash() != source_hash()


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> bayhunter_amd/csrc/libbayhunter_amd.so (in-tree)."""
    if not force and not needs_build():
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = os.environ.get("BH_EXTRA_HIPCC_FLAGS", "").split()      # compiler experiments
    sh = source_hash()
    cmd = [hipcc] + HIPCC_FLAGS + extra + ['-DBH_SRC_HASH="%s"' % sh] + SOURCES + ["-o", LIB_PATH]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    subprocess.run(cmd, cwd=CSRC, check=True)
    with open(LIB_PATH + ".srchash", "w") as fh:
        fh.write(sh + "\n")
    return LIB_PATH


def loaded_hash():
    """The source hash compiled into the loaded library (bh_version: '... src <hash>')."""
    v = load().bh_version().decode()
    return v.rsplit("src ", 1)[1] if "src " in v else ""


class ChainConfig(C.Structure):
    """struct bh_chain_config (priors and initparams of the reference, src/defaults/defaults.ini)."""
    _fields_ = [("ntargets", C.c_int), ("layers_min", C.c_int), ("layers_max", C.c_int),
                ("vs_min", C.c_double), ("vs_max", C.c_double), ("z_min", C.c_double), ("z_max", C.c_double),
                ("vpvs_fixed", C.c_int), ("vpvs_min", C.c_double), ("vpvs_max", C.c_double),
                ("has_mantle", C.c_int), ("mantle_vs", C.c_double), ("mantle_vpvs", C.c_double),
                ("has_mohoest", C.c_int), ("moho_mean", C.c_double), ("moho_std", C.c_double),
                ("thickmin", C.c_double), ("has_lvz", C.c_int), ("has_hvz", C.c_int),
                ("lvz", C.c_double), ("hvz", C.c_double), ("propdist", C.c_double * 5),
                ("acceptance", C.c_double * 2), ("iter_burnin", C.c_long), ("iter_main", C.c_long),
                ("noise_fixed", C.c_int * (2 * MAX_TARGETS)), ("noise_lo", C.c_double * (2 * MAX_TARGETS)),
                ("noise_hi", C.c_double * (2 * MAX_TARGETS))]


class ChainStorage(C.Structure):
    """struct bh_chain_storage: caller-owned float32 sample arrays (src/mcmcOptimizer.py:77-125)."""
    _fields_ = [("nmodels", C.c_long), ("models", C.c_void_p), ("misfits", C.c_void_p),
                ("likes", C.c_void_p), ("noise", C.c_void_p), ("vpvs", C.c_void_p), ("iter", C.c_void_p)]


_vp = C.c_void_p
_SIGS = {
    "bh_chains_create": (C.c_int, [C.POINTER(ChainConfig), C.c_int, _vp, C.POINTER(ChainStorage),
                                   C.POINTER(_vp)]),
    "bh_chains_destroy": (None, [_vp]),
    "bh_chains_set_threads": (C.c_int, [_vp, C.c_int]),
    "bh_chains_propose": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, C.POINTER(C.c_int)]),
    "bh_chains_accept": (C.c_int, [_vp, _vp, _vp]),
    "bh_chains_moves": (C.c_int, [_vp, _vp]),
    "bh_chains_accepted": (C.c_int, [_vp, _vp]),
    "bh_chains_done": (C.c_int, [_vp]),
    "bh_chains_iteration": (C.c_long, [_vp]),
    "bh_chains_set_lookahead": (C.c_int, [_vp, C.c_int]),
    "bh_chains_lookahead": (C.c_int, [_vp]),
    "bh_chains_rows": (C.c_long, [_vp]),
    "bh_chains_advance": (C.c_int, [_vp, C.POINTER(C.c_long), C.POINTER(C.c_long), C.POINTER(C.c_long)]),
    "bh_chains_iterations": (C.c_int, [_vp, _vp]),
    "bh_chains_counters": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "bh_chains_current": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_int), _vp, _vp, C.POINTER(C.c_double),
                                    C.POINTER(C.c_double), _vp]),
    "bh_chains_get_rng": (C.c_int, [_vp, C.c_int, _vp, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                    C.POINTER(C.c_double)]),
    "bh_chains_set_rng": (C.c_int, [_vp, C.c_int, _vp, C.c_int, C.c_int, C.c_double]),
    "bh_chains_draw": (C.c_int, [_vp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, _vp]),
    "bh_eval_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(SwdTarget), _vp, C.c_int, C.c_int,
                                 C.POINTER(RfParams), C.c_int, C.POINTER(LikeTarget), C.c_int, _vp, _vp, C.c_size_t,
                                 C.c_int, C.POINTER(EvalInterp), C.c_int, C.POINTER(_vp)]),
    "bh_eval_destroy": (None, [_vp]),
    "bh_eval_buffers": (C.c_int, [_vp] + [C.POINTER(_vp)] * 5),
    "bh_eval_submit": (C.c_int, [_vp, C.c_int]),
    "bh_eval_wait": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "bh_eval_set_concurrency": (C.c_int, [_vp, C.c_int]),
    "bh_swd_hint": (C.c_int, [C.c_double, C.c_int]),
    "bh_version": (C.c_char_p, []),
    "bh_last_error": (C.c_char_p, []),
    "bh_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "bh_set_device": (C.c_int, [C.c_int]),
    "bh_swd_set_kernel": (C.c_int, [C.c_int]),
    "bh_swd_last_form": (C.c_int, []),
    "bh_swd_last_forms": (C.c_int, [C.POINTER(C.c_int), C.c_int]),
    "bh_swd_set_forms": (C.c_int, [C.POINTER(C.c_int), C.c_int]),
    "bh_swd_plan_forms": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(SwdTarget), C.c_int, C.POINTER(C.c_int)]),
    "bh_rf_active_frequencies": (C.c_int, [C.POINTER(RfParams)]),
    "surfdisp96_": (None, [_vp] * 13),
    "synrf_cwrap": (C.c_int, [C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                              C.c_double, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "bh_swd_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.POINTER(SwdTarget)]),
    "bh_swd_batch": (C.c_int, [C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, C.c_int,
                               C.POINTER(SwdTarget), _vp, _vp, C.c_int, _vp, _vp, C.c_size_t, _vp]),
    "bh_swd_batch_ordered": (C.c_int, [C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, C.c_int,
                                       C.POINTER(SwdTarget), _vp, _vp, C.c_int, _vp, _vp, _vp, C.c_size_t, _vp]),
    "bh_swd_order_keys": (C.c_int, [C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, C.c_double, C.c_int, _vp, _vp]),
    "bh_rf_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.POINTER(RfParams)]),
    "bh_rf_batch": (C.c_int, [C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                              C.POINTER(RfParams), _vp, C.c_int, _vp, C.c_size_t, _vp]),
    "bh_voronoi_to_layers": (C.c_int, [C.c_int, C.c_int, _vp, _vp, _vp, _vp, C.POINTER(ModelPriors),
                                       _vp, _vp, _vp]),
    "bh_likelihood_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.POINTER(LikeTarget)]),
    "bh_likelihood_batch": (C.c_int, [C.c_int, C.c_int, C.POINTER(LikeTarget), _vp, C.c_int, _vp,
                                      C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "bh_likelihood_stage": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(LikeTarget), _vp, C.c_int, _vp,
                                      C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "bh_surfdisp96": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                C.c_int, _vp, _vp, C.POINTER(C.c_int)]),
    "bh_synrf": (C.c_int, [C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                           C.c_double, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "bh_selftest_division": (C.c_int, [C.c_long, C.c_uint, C.c_int, C.POINTER(C.c_long)]),
    "bh_malloc": (C.c_int, [C.POINTER(_vp), C.c_size_t]),
    "bh_free": (C.c_int, [_vp]),
    "bh_memcpy_h2d": (C.c_int, [_vp, _vp, C.c_size_t, _vp]),
    "bh_memcpy_d2h": (C.c_int, [_vp, _vp, C.c_size_t, _vp]),
    "bh_stream_synchronize": (C.c_int, [_vp]),
    "bh_stream_retire": (C.c_int, [_vp]),
    "bh_stream_create": (C.c_int, [C.POINTER(_vp)]),
    "bh_stream_destroy": (C.c_int, [_vp]),
    "bh_rf_cached_tables": (C.c_int, []),
}
EXPORTS = sorted(_SIGS)

_lib = None


def load():
    """dlopen the in-tree library; raise (never fall back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BayHunterAmdError(
                "%s not found: build it with bayhunter_amd.build() / python -c "
                "'import __graft_entry__ as g; g.build()'. There is no CPU fallback." % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc):
    if rc != BH_OK:
        msg = load().bh_last_error().decode()
        raise BayHunterAmdError("libbayhunter_amd error %d: %s" % (rc, msg))


def set_swd_kernel(mode):
    """'auto' | 'lane' | 'team' | 'team32' | 'team16' | 'team8' | 'team128' | 'team256' | 'team512'
    (include/bayhunter_amd.h, bh_swd_set_kernel)."""
    check(load().bh_swd_set_kernel({'auto': 0, 'lane': 1, 'team': 2, 'team32': 3, 'team16': 4, 'team8': 5,
                                    'team128': 6, 'team256': 7, 'team512': 8}[mode]))


def set_swd_forms(forms=None):
    """Kernel form per target for this thread's next calls with len(forms) targets ('lane', 'team8', ... or
    lanes per search); None: the library chooses again (bh_swd_set_forms)."""
    if not forms:
        check(load().bh_swd_set_forms(None, 0))
        return
    names = {'lane': 0, 'team': 64, 'team8': 8, 'team16': 16, 'team32': 32, 'team128': 128, 'team256': 256, 'team512': 512}
    arr = (C.c_int * len(forms))(*[names.get(f, f) for f in forms])
    check(load().bh_swd_set_forms(arr, len(forms)))


def device_count():
    n = C.c_int(0)
    load().bh_device_count(C.byref(n))
    return n.value
