"""Voronoi-nuclei parametrisation -> layered models, and the prior checks on a proposal.

Stand-in for the two pieces of the reference that sit immediately in front of the forward path:
`Model.get_vp_vs_h` / `Model.get_vp` (src/Models.py:26-52) and `SingleChain._validmodel`
(src/SingleChain.py:330-392).  `Model` is the single-model NumPy form (same static-method names
and results); `layers_from_voronoi` does the same for a batch on the GPU (bh_voronoi_to_layers)
and returns the packed model block that ForwardEngine consumes without another copy.
"""
import numpy as np

from . import _lib


class Model(object):
    """A model vector is [vs_1..vs_n, z_1..z_n] (nuclei velocities and depths, NaN padded)."""

    @staticmethod
    def split_modelparams(model):
        model = model[~np.isnan(model)]
        n = model.size // 2
        return n, model[:n], model[-n:]

    @staticmethod
    def get_vp(vs, vpvs=1.73, mantle=(4.3, 1.8)):
        """vp = vs*vpvs in the crust; from the first layer with vs >= mantle[0] downwards
        vp = vs*mantle[1]."""
        vp = vs * vpvs
        deep = np.nonzero(vs >= mantle[0])[0]
        if deep.size:
            vp[deep[0]:] = vs[deep[0]:] * mantle[1]
        return vp

    @staticmethod
    def get_vp_vs_h(model, vpvs=1.73, mantle=None):
        """Interfaces lie midway between neighbouring nuclei; the half-space gets h = 0."""
        n, vs, z_vnoi = Model.split_modelparams(model)
        interfaces = (z_vnoi[:n - 1] + z_vnoi[1:n]) / 2.
        h = np.concatenate((interfaces - np.concatenate(([0], interfaces[:-1])), [0]))
        vp = Model.get_vp(vs, vpvs, mantle) if mantle is not None else vs * vpvs
        return vp, vs, h


def valid_model(model, vpvs, priors, thickmin=0., lowvelperc=None, highvelperc=None, mantle=None):
    """SingleChain._validmodel for one model vector: layer count, minimum thickness, vs and
    interface-depth priors, optional low/high velocity-zone limits."""
    vp, vs, h = Model.get_vp_vs_h(model, vpvs, mantle)
    nlayers = h.size - 1
    if not (priors['layers'][0] <= nlayers <= priors['layers'][1]):
        return False
    if np.any(h[:-1] < thickmin):
        return False
    if np.any(vs < priors['vs'][0]) or np.any(vs > priors['vs'][1]):
        return False
    z = np.cumsum(h)
    if np.any(z < priors['z'][0]) or np.any(z > priors['z'][1]):
        return False
    if lowvelperc is not None and not np.all(vs[1:] - (vs[:-1] * (1 - lowvelperc)) > 0):
        return False
    if highvelperc is not None and not np.all((vs[:-1] * (1 + highvelperc)) - vs[1:] > 0):
        return False
    return True


def make_priors(priors, thickmin=0., lowvelperc=None, highvelperc=None, mantle=None):
    nan = float('nan')
    return _lib.ModelPriors(int(priors['layers'][0]), int(priors['layers'][1]),
                            float(priors['vs'][0]), float(priors['vs'][1]),
                            float(priors['z'][0]), float(priors['z'][1]), float(thickmin),
                            nan if lowvelperc is None else float(lowvelperc),
                            nan if highvelperc is None else float(highvelperc),
                            nan if mantle is None else float(mantle[0]),
                            nan if mantle is None else float(mantle[1]))


def layers_from_voronoi(VSN, ZV, nlay, vpvs, priors, thickmin=0., lowvelperc=None,
                        highvelperc=None, mantle=None, device=None, stream=None):
    """Batch on the GPU.  VSN, ZV: [B, Lmax] nuclei (rows sorted by depth, padding ignored),
    nlay[B], vpvs[B].  Returns (models, valid): engine.DeviceModels (pass it straight to
    ForwardEngine.run / JointTarget.evaluate_batch) and int32 prior-check flags."""
    import ctypes as C
    import torch
    lib = _lib.load()
    if not torch.cuda.is_available():
        raise _lib.BayHunterAmdError("no HIP device visible to torch; there is no CPU fallback")
    dev = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)

    def as_dev(x, dt):
        t = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))
        return t.to(device=dev, dtype=dt).contiguous()
    VSN, ZV = as_dev(VSN, torch.float64), as_dev(ZV, torch.float64)
    nlay, vpvs = as_dev(nlay, torch.int32), as_dev(vpvs, torch.float64)
    B, Lmax = VSN.shape
    packed = torch.empty((B, 4, Lmax), dtype=torch.float64, device=dev)
    valid = torch.empty(B, dtype=torch.int32, device=dev)
    pri = make_priors(priors, thickmin, lowvelperc, highvelperc, mantle)
    st = torch.cuda.current_stream(dev) if stream is None else stream
    with torch.cuda.device(dev):
        _lib.check(lib.bh_voronoi_to_layers(B, Lmax, nlay.data_ptr(), VSN.data_ptr(), ZV.data_ptr(),
                                            vpvs.data_ptr(), C.byref(pri), packed.data_ptr(),
                                            valid.data_ptr(), C.c_void_p(st.cuda_stream)))
    from .engine import DeviceModels
    return DeviceModels(packed, nlay), valid
