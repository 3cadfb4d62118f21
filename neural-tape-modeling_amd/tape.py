"""N4 ("next" row): the magnetisation stage of the reference's white-box tape simulator -- `Tape.H_mag`
(code/tape.py:516-551) with `Tape._f` (:587-635): Jiles-Atherton hysteresis integrated with RK4 at the
oversampled rate, fp64, stateful across calls (the per-sample Python loop that dominates the reference's target
generation) -- and the stages around it that need no resampler: H_pre, bias, H_rec in front (:466-514), H_play without
the loss filter and H_post behind (:565-579).  The two torchaudio resamplers (oversample / downsample) and the
lfilter-based playback loss are not built (nothing to pin them against here): the caller supplies the 16x
oversampled signal."""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import ptr


class TapeMagnetization:
    """Attribute names follow the reference's Tape class (TAPE_*, Ts_OS, M_prev / H_prev / Hprime_prev)."""

    def __init__(self, batch_size=1, fs=int(48e3), oversampling=16, device="cuda", signal_amplitude=1e-3,
                 bias_amplitude=5e-3, bias_enable=True):
        self.batch_size, self.fs, self.oversampling = batch_size, fs, oversampling
        self.Ts_OS = 1 / (fs * oversampling)
        # TAPE, Holters & Zoelzer (code/tape.py:251-256)
        self.TAPE_Ms, self.TAPE_A, self.TAPE_ALPHA, self.TAPE_K, self.TAPE_C = 1.6e6, 1.1e3, 1.6e-3, 4.0e2, 1.7e-1
        self.device = torch.device(device)
        self._state = torch.zeros(batch_size, 3, dtype=torch.float64, device=self.device)   # code/tape.py:303-309
        # pre-amplifier, bias, record and playback heads, post-amplifier (code/tape.py:164-301)
        self.signal_amplitude, self.bias_amplitude, self.bias_enable = signal_amplitude, bias_amplitude, bias_enable
        self.BIAS_FREQ, self.bias_phase, self.T_STARTUP, self.FLAG_STARTUP = 48e3, 0.0, 1e-2, True
        self.REC_N, self.REC_E, self.REC_G = 100, 0.1, 6e-6
        self.TAPE_V, self.PLAY_N, self.PLAY_E, self.PLAY_G = 7.5 * 2.54e-2, 1000, 1.0, 6e-6
        self.PLAY_MU0, self.PLAY_W, self.PLAY_D = 1.257e-6, 0.125 * 2.54e-2, 20e-6
        unity_term = 0.75 if np.isclose(signal_amplitude, 1e-4) else 0.525 if np.isclose(signal_amplitude, 1e-3) else 0.935
        self.POST_GAIN = unity_term * (1 / signal_amplitude) * (1 / self.PLAY_D)

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

    # ---- the stages around H_mag (fp64, names as in the reference) --------------------------------------------
    def H_pre(self, V_in):
        """Pre-amplifier, code/tape.py:466-469."""
        return self.signal_amplitude * V_in

    def bias_signal(self, n):
        """The bias waveform of the next `n` oversampled samples (host, numpy, exactly as code/tape.py:480-499: the
        time axis is a linspace that INCLUDES its end point and the phase enters twice); advances bias_phase and
        clears the start-up flag like the reference."""
        t_bias = np.linspace(self.bias_phase, self.bias_phase + n * self.Ts_OS, n)
        b = np.sin(2 * np.pi * self.BIAS_FREQ * t_bias + self.bias_phase)
        amp = self.bias_amplitude * np.ones(t_bias.shape)
        if self.FLAG_STARTUP:
            settle = int(self.T_STARTUP / (2 * self.Ts_OS))
            amp[:settle] = np.zeros(settle)
            amp[settle:settle + settle] = np.linspace(0, self.bias_amplitude, int(settle))
            self.FLAG_STARTUP = False
        b *= amp
        self.bias_phase += len(t_bias) * self.Ts_OS
        return b

    @torch.no_grad()
    def record_field(self, I_in_OS):
        """bias + H_rec in one pass (code/tape.py:476-514): I (B,N) fp64 on the device -> H (B,N)."""
        if not I_in_OS.is_cuda:
            raise RuntimeError("TapeMagnetization.record_field: this engine runs on a HIP device only (no CPU fallback)")
        I = I_in_OS.to(torch.float64).contiguous()
        B, N = I.shape
        bias = torch.from_numpy(self.bias_signal(N)).to(I.device) if self.bias_enable else None
        H = torch.empty_like(I)
        rc = _lib.lib().ntm_tape_record_field(ptr(I), ptr(bias) if bias is not None else None, ptr(H), B, N,
                                              float(self.REC_N * self.REC_E), float(self.REC_G), _lib.current_stream())
        _lib.check(rc, "ntm_tape_record_field")
        return H

    def H_play(self, M):
        """Playback head without the loss filter, code/tape.py:565-574."""
        g = self.PLAY_N * self.PLAY_W * self.PLAY_E * self.TAPE_V * self.PLAY_MU0 * self.PLAY_G
        return g * M

    def H_post(self, x):
        """Post-amplifier, code/tape.py:576-579."""
        return self.POST_GAIN * x
