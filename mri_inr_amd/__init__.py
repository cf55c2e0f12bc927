"""mri_inr_amd -- MI355X-native modulated-SIREN inference path (drop-in for one path of mri-inr).

Public surface mirrors the reference's: ``ModulatedSiren`` (src/networks/modulated_siren.py) and
``load_configuration`` (src/configuration/configuration.py).  Compute lives in libmsiren.so
(hand-written gfx950 HIP kernels behind the C ABI of include/msiren.h).
"""

from .configuration import load_configuration, model_kwargs  # noqa: F401
from .model import ModulatedSiren  # noqa: F401

__all__ = ["ModulatedSiren", "load_configuration", "model_kwargs"]
__version__ = "0.1.0"
