"""transkun_amd: MI355X-native (gfx950) Neural Semi-CRF interval layer, drop-in for
Transkun's `transkun.CRF.NeuralSemiCRFInterval` and `ScaledInnerProductIntervalScorer`.

The compute path is hand-written HIP behind a C-ABI shared library (include/semicrf_hip.h,
transkun_amd/csrc).  There is no CPU fallback: using the layer without the built library or
without a GPU raises.
"""
from . import CRF  # noqa: F401
from .CRF import NeuralSemiCRFInterval  # noqa: F401

__all__ = ["CRF", "NeuralSemiCRFInterval"]
