"""advntr_amd -- MI355X-native profile-HMM Viterbi scoring for adVNTR-style VNTR genotyping.

One hot path, hand-written in HIP for gfx950 (advntr_amd/csrc), behind a C ABI (include/advntr_hip.h)
and a host-side mirror of the pomegranate / hmm_utils surface the reference drives it through.
Importing the package needs no GPU; scoring does, and fails loudly without the built extension.
"""
__version__ = "0.1.0"

from . import settings  # noqa: F401
from .pomegranate import DiscreteDistribution, HiddenMarkovModel, State  # noqa: F401
