"""neural-tape-modeling_amd -- MI355X-native engine for the tape-nonlinearity forward path of
01tot10/neural-tape-modeling (GRU-HS[64], DiffDelGRU-HS[64], TCN contrast point).

The directory name contains a hyphen, so import it through the root-level shim:  `import ntm_amd`.
Layout:  csrc/ (HIP kernels + C ABI -> libntm.so), model.py (reference object protocol),
utilities.py (name parsers), weights.py (exported checkpoints), harness.py (test-model.py loss loop),
distributed.py (stream sharding + the one RCCL all-reduce).
"""
from . import _lib  # noqa: F401
from ._lib import NtmError, build  # noqa: F401
from .model import (RNN, DCPreESR, DiffDelRNN, ESRLoss, MRSTFTLoss, TimeVaryingDelayLine, ValLossSupervised,  # noqa: F401
                    esr_dcpre_sums, esr_per_segment, esr_sums, mel_sums, spec_sums, stft_sums)
from .tape import Tape, TapeMagnetization  # noqa: F401
from .tcn import TCN  # noqa: F401
from .utilities import nextpow2, parse_hidden_size, parse_loss, parse_model  # noqa: F401
from . import distributed, feeder, harness, weights  # noqa: F401

__all__ = ["RNN", "DiffDelRNN", "TimeVaryingDelayLine", "TCN", "Tape", "TapeMagnetization", "ESRLoss", "DCPreESR", "MRSTFTLoss", "ValLossSupervised", "stft_sums", "spec_sums", "mel_sums", "esr_dcpre_sums", "esr_sums", "esr_per_segment",
           "parse_hidden_size", "parse_model", "parse_loss", "nextpow2", "weights", "build", "NtmError"]
