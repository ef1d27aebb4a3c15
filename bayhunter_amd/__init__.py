"""bayhunter_amd -- MI355X-native forward engine for BayHunter's per-proposal likelihood hot path
(surface-wave dispersion + receiver functions), behind the reference's forward-plugin interface.
"""
from ._lib import BayHunterAmdError, build, device_count, load  # noqa: F401
from .plugins import RFminiModRF, SurfDisp  # noqa: F401

__all__ = ['SurfDisp', 'RFminiModRF', 'build', 'load', 'device_count', 'BayHunterAmdError']
