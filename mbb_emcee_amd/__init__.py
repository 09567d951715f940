"""mbb_emcee_amd -- the mbb_emcee per-walker likelihood hot path on MI355X (gfx950).

Same API surface as the reference for this path: ``modified_blackbody``,
``response`` / ``response_set``, ``likelihood`` (the emcee lnprob callable) and
``mbb_fitter``.  All SED / likelihood arithmetic runs in hand-written HIP
kernels behind a C-ABI (include/mbb_hip.h); there is no CPU fallback.
"""
from .response import response, response_set, special_types
from .modified_blackbody import modified_blackbody, alpha_merge_eqn
from .utility import isiterable
from .likelihood import likelihood
from .ensemble import EnsembleSampler
from .device_sampler import DeviceEnsembleSampler
from .mbb_fit import mbb_fitter

__version__ = "0.1.0"
__all__ = ["response", "response_set", "modified_blackbody", "alpha_merge_eqn", "isiterable", "likelihood",
           "EnsembleSampler", "DeviceEnsembleSampler", "mbb_fitter"]
