"""N4 ("next" row): the magnetisation stage of the reference's white-box tape simulator -- `Tape.H_mag`
(code/tape.py:516-551) with `Tape._f` (:587-635): Jiles-Atherton hysteresis integrated with RK4 at the
oversampled rate, fp64, stateful across calls (the per-sample Python loop that dominates the reference's target
generation) -- and the stages around it that need no resampler: H_pre, bias, H_rec in front (:466-514), H_play without
the loss filter and H_post behind (:565-579) -- all pinned to the reference (goldens g9, g14).
`Tape` adds what is left of `Tape.__call__` (:389-464): the two torchaudio sinc resamplers (oversample / downsample,
:330-332, 471-474, 553-558) and the lfilter-based playback loss (:333-374, 565-574).  torchaudio is un-vendored and absent
from the build container, so those two follow its PUBLISHED algorithms (Resample: sinc interpolation, Hann window,
lowpass_filter_width 6, rolloff 0.99; lfilter with a = [1, 0, ...] and its default clamp) -- parity unpinned."""
import math
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


def sinc_resample_kernel(orig_freq, new_freq, lowpass_filter_width=6, rolloff=0.99):
    """torchaudio.functional's `_get_sinc_resample_kernel(..., resampling_method="sinc_interp_hann", dtype=float64)`
    restated (what `T.Resample(fs_orig, fs_new, dtype=torch.float64)` of code/tape.py:330-332 precomputes):
    -> (kernel float64 [new, 2 width + orig], width, orig, new) with orig / new reduced by their gcd."""
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base_freq = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base_freq)
    idx = np.arange(-width, width + orig, dtype=np.float64)[None, :] / orig
    t = np.arange(0, -new, -1, dtype=np.float64)[:, None] / new + idx
    t = np.clip(t * base_freq, -lowpass_filter_width, lowpass_filter_width)
    window = np.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    with np.errstate(invalid="ignore", divide="ignore"):
        kernels = np.where(t == 0, 1.0, np.sin(t) / t)
    return kernels * window * (base_freq / orig), width, orig, new


class Tape(TapeMagnetization):
    """The reference's `Tape` (code/tape.py:160-464) end to end on the device: V_in (B, N) at fs ->
    H_pre -> oversample -> bias + H_rec -> H_mag (Jiles-Atherton, RK4) -> downsample -> delay (dummy) -> H_play
    [-> playback loss] -> H_post -> V_out (B, N).  Stateful like the reference: magnetisation state, bias phase, and the
    previous call's oversampled magnetisation as left context of the downsampler (:553-558).  Same constructor
    arguments where they apply; `startup_enable` runs the reference's 10 ms of silence through the chain (:376-386)."""

    def __init__(self, batch_size=1, fs=int(48e3), oversampling=16, signal_amplitude=1e-3, bias_amplitude=5e-3,
                 bias_enable=True, playback_loss_enable=False, FIR_order=2**7, startup_enable=True, delay_enable=False,
                 device="cuda"):
        super().__init__(batch_size, fs, oversampling, device, signal_amplitude, bias_amplitude, bias_enable)
        self.Ts, self.fs_OS = 1 / fs, fs * oversampling
        self.delay_enable, self.playback_loss_enable, self.N_FIR, self.TAPE_DELTA = delay_enable, playback_loss_enable, FIR_order, 35e-6
        k_os, self._w_os, _, self._up = sinc_resample_kernel(fs, self.fs_OS)
        k_ds, self._w_ds, self._down, _ = sinc_resample_kernel(self.fs_OS, fs)
        self._k_os = torch.from_numpy(np.ascontiguousarray(k_os)).to(self.device)
        self._k_ds = torch.from_numpy(np.ascontiguousarray(k_ds)).to(self.device)
        self.b = torch.from_numpy(self._compute_filter()).to(self.device)
        self.M_OS = torch.empty(batch_size, 0, dtype=torch.float64, device=self.device)      # code/tape.py:303-328
        self.M = torch.empty(batch_size, 0, dtype=torch.float64, device=self.device)
        self.startup_enable = startup_enable
        if startup_enable:                                                                    # code/tape.py:376-386
            self(torch.zeros(batch_size, len(np.arange(0, self.T_STARTUP, self.Ts)), dtype=torch.float64, device=self.device))

    def _compute_filter(self):
        """Playback-loss FIR (spacing, thickness and gap loss), code/tape.py:333-374: -> h float64 [N_FIR]."""
        import scipy.fft
        f = np.linspace(0, self.fs, self.N_FIR)
        k = (2 * np.pi * f[1:int(self.N_FIR / 2)]) / self.TAPE_V
        loss = np.exp(-k * self.PLAY_D) * ((1 - np.exp(-k * self.TAPE_DELTA)) / (k * self.TAPE_DELTA)) * \
            (np.sin(k * self.PLAY_G / 2) / (k * self.PLAY_G / 2))
        H = np.zeros((self.N_FIR),)
        H[0] = 1
        H[1:int(self.N_FIR / 2)] = loss
        H[int(self.N_FIR / 2):] = np.flip(H[0:int(self.N_FIR / 2)], 0)
        return np.ascontiguousarray(np.abs(scipy.fft.ifft(H)), dtype=np.float64)

    def _resample(self, x, kernel, width, up, down):
        x = x.to(torch.float64).contiguous()
        B, N = x.shape
        M = int(math.ceil(up * N / down))                                  # torchaudio: ceil(new * length / orig)
        y = torch.empty(B, M, dtype=torch.float64, device=x.device)
        rc = _lib.lib().ntm_resample_fir(ptr(x), ptr(y), B, N, M, up, down, width, ptr(kernel), _lib.current_stream())
        _lib.check(rc, "ntm_resample_fir")
        return y

    def oversample(self, I_in):
        """code/tape.py:471-474."""
        return self._resample(I_in, self._k_os, self._w_os, self._up, 1)

    def downsample(self, M_OS):
        """code/tape.py:553-558: resample the previous call's oversampled magnetisation together with this one and keep
        the new part."""
        M = self._resample(torch.cat((self.M_OS, M_OS), dim=1), self._k_ds, self._w_ds, 1, self._down)
        return M[:, self.M.shape[1]:]

    def delay(self, M):
        """Delay (dummy in the reference too, code/tape.py:560-563)."""
        return M.clone()

    def H_play(self, M):
        """Playback head, code/tape.py:565-574, with the loss filter when enabled (lfilter: FIR + clamp to [-1, 1])."""
        V = super().H_play(M)
        if not self.playback_loss_enable:
            return V
        V = V.contiguous()
        out = torch.empty_like(V)
        rc = _lib.lib().ntm_fir_f64(ptr(V), ptr(out), V.shape[0], V.shape[1], ptr(self.b), self.N_FIR, 1, _lib.current_stream())
        _lib.check(rc, "ntm_fir_f64")
        return out

    @torch.no_grad()
    def __call__(self, V_in):
        """V_in (B, N) float64 on the device -> V_out (B, N) (code/tape.py:389-464; a batch smaller than batch_size
        is padded with zero streams like the reference does)."""
        if not V_in.is_cuda:
            raise RuntimeError("Tape: this engine runs on a HIP device only (no CPU fallback)")
        n_in = V_in.shape[0]
        if n_in < self.batch_size:
            V_in = torch.cat((V_in, torch.zeros(self.batch_size - n_in, V_in.shape[1], dtype=V_in.dtype, device=V_in.device)), 0)
        I_in_OS = self.oversample(self.H_pre(V_in.to(torch.float64)))
        M_OS = self.H_mag(self.record_field(I_in_OS))
        M = self.downsample(M_OS)
        V_out = self.H_post(self.H_play(self.delay(M)))
        self.M_OS, self.M = M_OS, M
        return V_out[:n_in]
