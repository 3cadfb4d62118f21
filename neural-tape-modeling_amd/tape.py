"""N4 ("next" row): the magnetisation stage of the reference's white-box tape simulator -- `Tape.H_mag`
(code/tape.py:516-551) with `Tape._f` (:587-635): Jiles-Atherton hysteresis integrated with RK4 at the
oversampled rate, fp64, stateful across calls.  Only this stage is built (the per-sample Python loop that
dominates the reference's target generation); resampling, bias, playback filters are out of scope."""
import ctypes

import torch

from . import _lib
from ._lib import ptr


class TapeMagnetization:
    """Attribute names follow the reference's Tape class (TAPE_*, Ts_OS, M_prev / H_prev / Hprime_prev)."""

    def __init__(self, batch_size=1, fs=int(48e3), oversampling=16, device="cuda"):
        self.batch_size, self.fs, self.oversampling = batch_size, fs, oversampling
        self.Ts_OS = 1 / (fs * oversampling)
        # TAPE, Holters & Zoelzer (code/tape.py:251-256)
        self.TAPE_Ms, self.TAPE_A, self.TAPE_ALPHA, self.TAPE_K, self.TAPE_C = 1.6e6, 1.1e3, 1.6e-3, 4.0e2, 1.7e-1
        self.device = torch.device(device)
        self._state = torch.zeros(batch_size, 3, dtype=torch.float64, device=self.device)   # code/tape.py:303-309

    M_prev = property(lambda self: self._state[:, 0])
    H_prev = property(lambda self: self._state[:, 1])
    Hprime_prev = property(lambda self: self._state[:, 2])

    @torch.no_grad()
    def H_mag(self, H):
        """H (B,N) float64 on a HIP device -> M (B,N); carries M_prev / H_prev / Hprime_prev."""
        if not H.is_cuda:
            raise RuntimeError("TapeMagnetization.H_mag: this engine runs on a HIP device only (no CPU fallback)")
        if H.dim() != 2 or H.shape[0] != self.batch_size:
            raise RuntimeError(f"H_mag: expected ({self.batch_size}, N), got {tuple(H.shape)}")
        H = H.to(torch.float64).contiguous()
        M = torch.empty_like(H)
        par = (ctypes.c_double * 5)(self.TAPE_Ms, self.TAPE_A, self.TAPE_ALPHA, self.TAPE_K, self.TAPE_C)
        rc = _lib.lib().ntm_tape_hmag(ptr(H), ptr(M), H.shape[0], H.shape[1], ptr(self._state), self.Ts_OS, par,
                                      _lib.current_stream())
        _lib.check(rc, "ntm_tape_hmag")
        return M
