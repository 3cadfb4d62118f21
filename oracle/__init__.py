"""CPU oracle for the tape-nonlinearity forward path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product (neural-tape-modeling_amd + libntm.so) never does; it fails loudly without its HIP
library instead of falling back to anything here.

Parity status: pinned by tests/golden/g1..g8 (generated from the reference by
tools/make_goldens.py); ESR, the MR-STFT loss and the TCN are "parity unpinned" (no reference source
exists; the MR-STFT restatement is pinned to torch.stft by g10).

Three layers, all restating code/model.py of the reference:
  * C (ntm_oracle.c via ctypes)  -- fast enough for 16x8192 / 1x65536 cases and the CPU baseline
  * numpy (np_*)                 -- tiny independent restatement used to cross-check the C
  * torch_gru_port               -- stock torch.nn.GRU + Linear on CPU, i.e. the very calls the
                                    reference makes at code/model.py:44-45,81-82 (baseline timing)
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_f32p = ctypes.POINTER(ctypes.c_float)
_f64p = ctypes.POINTER(ctypes.c_double)
_i32p = ctypes.POINTER(ctypes.c_int)


def build():
    """Compile libntm_oracle.so with gcc (idempotent)."""
    subprocess.run(["make", "-s", "-C", _HERE], check=True)


def lib():
    global _LIB
    if _LIB is None:
        path = os.environ.get("NTM_ORACLE_LIB") or os.path.join(_HERE, "libntm_oracle.so")   # override: the asan build
        if not os.path.exists(path):
            # never built lazily: a caller such as bench.py's cpu_baseline leg runs after the GPU is initialised and
            # must not spawn a compiler there -- __graft_entry__.build() / `make -C oracle` build it up front
            raise RuntimeError(f"{path} is missing: build the oracle first (`make -C {_HERE}` or __graft_entry__.build())")
        _LIB = ctypes.CDLL(path)
        _LIB.ntmo_gru_forward.argtypes = [_f32p] * 6 + [ctypes.c_int, _f32p, _f32p,
                                                        ctypes.c_int64, ctypes.c_int64, _f32p]
        _LIB.ntmo_gru_forward_mt.argtypes = _LIB.ntmo_gru_forward.argtypes + [ctypes.c_int]
        _LIB.ntmo_gru_forward_io.argtypes = [_f32p] * 6 + [ctypes.c_int] * 3 + [_f32p, _f32p, ctypes.c_int64, ctypes.c_int64, _f32p]
        _LIB.ntmo_gru_forward_f64_mt.argtypes = [_f32p] * 6 + [ctypes.c_int, _f32p, _f64p, ctypes.c_int64, ctypes.c_int64, _f64p, ctypes.c_int]
        _LIB.ntmo_delay_forward_f64.argtypes = [_f64p, _f32p, _f64p, ctypes.c_int64, ctypes.c_int64, _f64p, ctypes.c_int, ctypes.c_int]
        _LIB.ntmo_delay_forward.argtypes = [_f32p, _f32p, _f32p, ctypes.c_int64, ctypes.c_int64,
                                            _f32p, ctypes.c_int, ctypes.c_int]
        _LIB.ntmo_esr_sums.argtypes = [_f32p, _f32p, ctypes.c_int64, ctypes.c_int64,
                                       ctypes.c_int64, _f64p]
        _LIB.ntmo_esr_dcpre_sums.argtypes = [_f32p, _f32p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                             ctypes.c_float, _f64p]
        _LIB.ntmo_tape_hmag.argtypes = [_f64p, _f64p, ctypes.c_int64, ctypes.c_int64, _f64p, ctypes.c_double, _f64p]
        _LIB.ntmo_tcn_forward.argtypes = [_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _i32p,
                                          _f32p, _f32p, ctypes.c_int64, ctypes.c_int64]
        _LIB.ntmo_tcn_forward_mt.argtypes = _LIB.ntmo_tcn_forward.argtypes + [ctypes.c_int]
    return _LIB


def _p(a):
    return a.ctypes.data_as(_f32p) if a is not None else None


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class Weights:
    """The six GRU-HS parameters in the reference's state_dict layout."""

    def __init__(self, w_ih, w_hh, b_ih, b_hh, w_o, b_o=None):
        self.w_ih = _c(w_ih).reshape(-1)
        self.H = self.w_ih.size // 3
        self.w_hh = _c(w_hh).reshape(3 * self.H, self.H)
        self.b_ih = _c(b_ih).reshape(-1)
        self.b_hh = _c(b_hh).reshape(-1)
        self.w_o = _c(w_o).reshape(-1)
        self.b_o = None if b_o is None else _c(b_o).reshape(-1)

    @classmethod
    def from_state_dict(cls, sd):
        g = lambda k: np.asarray(sd[k], dtype=np.float32) if k in sd else None  # noqa: E731
        return cls(g("GRU.weight_ih_l0"), g("GRU.weight_hh_l0"), g("GRU.bias_ih_l0"),
                   g("GRU.bias_hh_l0"), g("output.weight"), g("output.bias"))


def gru_forward(w, x, h=None, threads=1):
    """x [B,T] -> (y [B,T], h_out [B,H]).  code/model.py:81-82."""
    x = _c(x)
    B, T = x.shape
    h = np.zeros((B, w.H), np.float32) if h is None else _c(h).copy()
    y = np.empty_like(x)
    args = [_p(w.w_ih), _p(w.w_hh), _p(w.b_ih), _p(w.b_hh), _p(w.w_o), _p(w.b_o), w.H, _p(x), _p(y),
            B, T, _p(h)]
    rc = lib().ntmo_gru_forward_mt(*args, threads) if threads > 1 else lib().ntmo_gru_forward(*args)
    assert rc == 0
    return y, h


def warm_state(w, n=1024):
    """Hidden state after RNN.warm_start(): 1024 zero samples from h=0 (code/model.py:58-65)."""
    _, h = gru_forward(w, np.zeros((1, n), np.float32))
    return h


def gru_predict(w, x, threads=1):
    """Batched generalisation of RNN.predict (code/model.py:218-246): warm state broadcast to all
    streams, then the whole sequence (the reference's 2048-chunking only carries state)."""
    x = _c(x)
    h0 = np.repeat(warm_state(w), x.shape[0], axis=0)
    return gru_forward(w, x, h0, threads)


def delay_forward(x, d, buf, warmup=False):
    """TimeVaryingDelayLine.forward, code/model.py:269-320.  Returns (y, new_buf); raises
    AssertionError like the reference (:284) if max(d) > D."""
    x, d = _c(x), _c(d)
    buf = _c(buf).copy()
    B, T = x.shape
    D = buf.shape[1]
    y = np.empty_like(x)
    rc = lib().ntmo_delay_forward(_p(x), _p(d), _p(y), B, T, _p(buf), D, int(bool(warmup)))
    if rc == 1:
        raise AssertionError("max_delay >= max(dt) violated")
    assert rc == 0
    return y, buf


def diffdel_forward(w, x, d, h, buf, warmup=False, threads=1):
    """DiffDelRNN.forward, code/model.py:393-424 -> (y, pre_d, h_out, buf_out)."""
    pre, h = gru_forward(w, x, h, threads)
    y, buf = delay_forward(pre, d, buf, warmup)
    return y, pre, h, buf


def diffdel_predict(w, x, d, max_delay, threads=1):
    """DiffDelRNN.predict, code/model.py:618-653, batched generalisation (same warm state for all
    streams: zero input, zero delay for 1024 samples)."""
    x, d = _c(x), _c(d)
    B = x.shape[0]
    D = int(max_delay) + 1                                  # code/model.py:372-375
    z = np.zeros((1, 1024), np.float32)
    _, _, h1, b1 = diffdel_forward(w, z, z, None, np.zeros((1, D), np.float32))
    return diffdel_forward(w, x, d, np.repeat(h1, B, 0), np.repeat(b1, B, 0), threads=threads)


def gru_forward_io(sd, x, h=None):
    """RNN.forward for ANY input_size / output_size (code/model.py:67-88), the reference's reshape semantics included:
    x (B, C, T) -> y (B, O, T) where both are reinterpretations (reshape, not transpose) of the [T][C] / [T][O] matrices the
    GRU and the head see.  `sd`: the reference's state_dict as numpy arrays.  -> (y, h_out [B,H])."""
    w_ih = _c(sd["GRU.weight_ih_l0"])
    w_hh = _c(sd["GRU.weight_hh_l0"])
    w_o = _c(sd["output.weight"])
    H, I, O = w_hh.shape[1], w_ih.shape[1], w_o.shape[0]
    b_o = _c(sd["output.bias"]) if sd.get("output.bias") is not None else None
    x = _c(x)
    B, C, T = x.shape
    assert C == I, f"input has {C} channels, the GRU takes {I}"
    h = np.zeros((B, H), np.float32) if h is None else _c(h).copy()
    y = np.empty((B, T * O), np.float32)
    assert lib().ntmo_gru_forward_io(_p(w_ih), _p(w_hh), _p(_c(sd["GRU.bias_ih_l0"])), _p(_c(sd["GRU.bias_hh_l0"])), _p(w_o), _p(b_o),
                                     H, I, O, _p(x.reshape(B, T * I)), _p(y), B, T, _p(h)) == 0
    return y.reshape(B, O, T), h


def _pd(a):
    return a.ctypes.data_as(_f64p)


def gru_forward_f64(w, x, h=None, threads=1):
    """fp64 mode of gru_forward (ntm_oracle.c gru_stream_f64): fp32 parameters and input, double state / accumulators / gates.
    x [B,T] fp32 -> (y [B,T] float64, h_out [B,H] float64).  A yardstick for rounding, not a parity target."""
    x = _c(x)
    B, T = x.shape
    h = np.zeros((B, w.H), np.float64) if h is None else np.ascontiguousarray(h, np.float64).copy()
    y = np.empty((B, T), np.float64)
    assert lib().ntmo_gru_forward_f64_mt(_p(w.w_ih), _p(w.w_hh), _p(w.b_ih), _p(w.b_hh), _p(w.w_o), _p(w.b_o), w.H, _p(x), _pd(y),
                                         B, T, _pd(h), max(1, int(threads))) == 0
    return y, h


def gru_predict_f64(w, x, threads=1):
    """gru_predict in fp64 mode (warm state from 1024 zero samples, also in double)."""
    x = _c(x)
    _, h1 = gru_forward_f64(w, np.zeros((1, 1024), np.float32))
    return gru_forward_f64(w, x, np.repeat(h1, x.shape[0], 0), threads)


def delay_forward_f64(x, d, buf, warmup=False):
    x = np.ascontiguousarray(x, np.float64)
    d = _c(d)
    buf = np.ascontiguousarray(buf, np.float64).copy()
    B, T = x.shape
    y = np.empty_like(x)
    rc = lib().ntmo_delay_forward_f64(_pd(x), _p(d), _pd(y), B, T, _pd(buf), buf.shape[1], int(bool(warmup)))
    if rc == 1:
        raise AssertionError("max_delay >= max(dt) violated")
    assert rc == 0
    return y, buf


def diffdel_predict_f64(w, x, d, max_delay, threads=1):
    """diffdel_predict in fp64 mode -> (y, pre_d, h_out, buf_out), all float64."""
    x, d = _c(x), _c(d)
    B = x.shape[0]
    D = int(max_delay) + 1
    z = np.zeros((1, 1024), np.float32)
    p1, h1 = gru_forward_f64(w, z)
    _, b1 = delay_forward_f64(p1, z, np.zeros((1, D)))
    pre, h = gru_forward_f64(w, x, np.repeat(h1, B, 0), threads)
    y, buf = delay_forward_f64(pre, d, np.repeat(b1, B, 0))
    return y, pre, h, buf


def esr_sums(y, t, skip=0):
    y, t = _c(y), _c(t)
    B, T = y.shape
    out = np.empty((B, 2), np.float64)
    assert lib().ntmo_esr_sums(_p(y), _p(t), B, T, skip, out.ctypes.data_as(_f64p)) == 0
    return out


ESR_EPS = 1e-5


def esr_dcpre_sums(y, t, skip=0, R=0.995):
    """Per-stream ESR sums after the DC blocker (1 - z^-1)/(1 - R z^-1): PARITY UNPINNED (ntm_oracle.c)."""
    y, t = _c(y), _c(t)
    B, T = y.shape
    out = np.empty((B, 2), np.float64)
    assert lib().ntmo_esr_dcpre_sums(_p(y), _p(t), B, T, skip, R, out.ctypes.data_as(_f64p)) == 0
    return out


def esr_per_segment(y, t, skip=0):
    """CoreAudioML ESRLoss per stream: mean(e^2)/(mean(t^2)+1e-5) over samples [skip,T)."""
    s = esr_sums(y, t, skip)
    n = y.shape[1] - skip
    return (s[:, 0] / n) / (s[:, 1] / n + ESR_EPS)


MRSTFT_RESOLUTIONS = ((1024, 120, 600), (2048, 240, 1200), (512, 50, 240))   # (n_fft, hop, win_length)
STFT_EPS = 1e-8


def stft_sums(y, t, skip=0, n_fft=1024, hop=120, win_length=600, eps=STFT_EPS):
    """Per-stream sums of auraloss.freq.STFTLoss (un-vendored submodule of the reference, imported at
    code/test-model.py:25 and instantiated with its defaults at :253) -- PARITY UNPINNED by the reference,
    pinned to torch.stft by tests/golden/g10 (tools/make_goldens_stft.py).  fp64 numpy restatement of
      X = torch.stft(x, n_fft, hop, win_length, hann_window(win_length))   (center, reflect padding, the window
                                                                            zero-padded to n_fft on both sides)
      mag = sqrt(clamp(re^2 + im^2, min=eps))
    over samples [skip, T) of each stream.  Returns [B,4] float64:
      sum (mag_t - mag_y)^2 | sum mag_t^2 | sum |ln mag_y - ln mag_t| | sum |mag_y - mag_t|
    and the number of (bin, frame) cells per stream."""
    y = np.asarray(y, np.float64)[:, skip:]
    t = np.asarray(t, np.float64)[:, skip:]
    B, L = y.shape
    assert L > n_fft // 2, "reflect padding needs more than n_fft/2 samples"
    n = np.arange(win_length)
    win = np.zeros(n_fft)
    left = (n_fft - win_length) // 2
    win[left:left + win_length] = 0.5 - 0.5 * np.cos(2 * np.pi * n / win_length)      # periodic Hann
    n_frames = 1 + L // hop
    idx = np.arange(n_frames)[:, None] * hop + np.arange(n_fft)[None, :]

    def mag(x):
        xp = np.pad(x, ((0, 0), (n_fft // 2, n_fft // 2)), mode="reflect")
        X = np.fft.rfft(xp[:, idx] * win, axis=-1)                                       # [B, frames, bins]
        return np.sqrt(np.maximum(X.real ** 2 + X.imag ** 2, eps))

    out = np.empty((B, 4))
    for b in range(B):                      # stream by stream: bounded memory
        my, mt = mag(y[b:b + 1]), mag(t[b:b + 1])
        out[b] = [((mt - my) ** 2).sum(), (mt ** 2).sum(), np.abs(np.log(my) - np.log(mt)).sum(), np.abs(my - mt).sum()]
    return out, n_frames * (n_fft // 2 + 1)


def mrstft_per_segment(y, t, skip=0, resolutions=MRSTFT_RESOLUTIONS):
    """auraloss MultiResolutionSTFTLoss() defaults (w_sc = w_log_mag = 1, w_lin_mag = w_phs = 0), evaluated
    per stream as the harness does (code/test-model.py:386-398): mean over resolutions of
    ||mag_t - mag_y||_F / ||mag_t||_F + mean |ln mag_y - ln mag_t|."""
    total = 0.0
    for n_fft, hop, win in resolutions:
        s, cells = stft_sums(y, t, skip, n_fft, hop, win)
        total = total + np.sqrt(s[:, 0]) / np.sqrt(s[:, 1]) + s[:, 2] / cells
    return total / len(resolutions)


SPEC_SCALES = (2048, 1024, 512, 256, 128, 64)        # code/evaluation.py:23
SPEC_LOG_FLOOR = 1e-5                                   # code/evaluation.py:44


def spec_sums(y, t, skip=0, n_fft=1024, hop=None, win_length=None, log_floor=SPEC_LOG_FLOOR):
    """Per-stream sums of the POWER-spectrogram terms of code/evaluation.py:75-84 (`TimeFreqConverter` =
    torchaudio.transforms.Spectrogram(n_fft, hop_length=n_fft//4), power 2; torchaudio is un-vendored here: its
    Spectrogram is torch.stft with a periodic Hann window of n_fft samples, centred, reflect padding -- pinned to
    torch.stft by golden g13).  Returns [B,4] float64: sum |P_y - P_t| | sum |log10 max(P_y,f) - log10 max(P_t,f)| |
    sum P_t | sum P_y, and the number of (bin, frame) cells per stream."""
    hop = n_fft // 4 if hop is None else hop
    win_length = n_fft if win_length is None else win_length
    y = np.asarray(y, np.float64)[:, skip:]
    t = np.asarray(t, np.float64)[:, skip:]
    B, L = y.shape
    win = np.zeros(n_fft)
    left = (n_fft - win_length) // 2
    win[left:left + win_length] = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(win_length) / win_length)
    n_frames = 1 + L // hop
    idx = np.arange(n_frames)[:, None] * hop + np.arange(n_fft)[None, :]

    def power(x):
        xp = np.pad(x, ((0, 0), (n_fft // 2, n_fft // 2)), mode="reflect")
        X = np.fft.rfft(xp[:, idx] * win, axis=-1)
        return X.real ** 2 + X.imag ** 2

    out = np.empty((B, 4))
    for b in range(B):
        py, pt = power(y[b:b + 1]), power(t[b:b + 1])
        out[b] = [np.abs(py - pt).sum(),
                  np.abs(np.log10(np.maximum(py, log_floor)) - np.log10(np.maximum(pt, log_floor))).sum(), pt.sum(), py.sum()]
    return out, n_frames * (n_fft // 2 + 1)


def mel_filterbank(sr=44100, n_fft=2048, n_mels=160, fmin=0.0, fmax=None):
    """`librosa.filters.mel(sr=, n_fft=, n_mels=, fmin=, fmax=)` with its defaults (htk=False: the Slaney / Auditory
    Toolbox mel scale; norm="slaney": every triangle scaled to unit area in Hz) -- the `mel_basis` of the reference's
    TimeFreqConverter (code/utilities/utilities.py:639-646).  librosa is NOT installed in the build container and
    the reference vendors no copy: this restates the published algorithm -- PARITY UNPINNED; tests check it against
    the one value librosa's own docstring prints (0.016 = mel(sr=22050, n_fft=2048)[0, 1]) and the unit-area property.
      mel(f)  = f / (200/3)                              for f < 1000 Hz
              = 15 + ln(f / 1000) / (ln(6.4) / 27)       above
      n_mels + 2 band edges equally spaced in mel between fmin and fmax (= sr/2); filter m rises from edge m to edge
      m+1 and falls to edge m+2 (piecewise linear in Hz, evaluated at the FFT bin centres k sr / n_fft), then is
      multiplied by 2 / (edge[m+2] - edge[m]).   -> float32 [n_mels, 1 + n_fft/2]"""
    fmax = sr / 2.0 if fmax is None else fmax
    f_sp, min_log_hz = 200.0 / 3.0, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, np.log(6.4) / 27.0

    def hz_to_mel(f):
        f = np.asarray(f, np.float64)
        return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-300) / min_log_hz) / logstep, f / f_sp)

    def mel_to_hz(m):
        m = np.asarray(m, np.float64)
        return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)

    fftfreqs = np.linspace(0.0, sr / 2.0, 1 + n_fft // 2)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    w = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0.0, np.minimum(lower, upper))
    w *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return w.astype(np.float32)


def mel_sums(y, t, skip=0, n_fft=2048, hop=None, n_mels=160, sr=44100, log_floor=SPEC_LOG_FLOOR):
    """Per-stream sums of the two mel entries of val_loss_supervised.forward (code/evaluation.py:86-92): the power
    spectrogram of `TimeFreqConverter(n_fft=2048, hop 512)` projected by `mel_basis` (matmul, :666), then
    |mel_y - mel_t| and |log10 max(mel_y, 1e-5) - log10 max(mel_t, 1e-5)|.  Returns [B,4] float64: the two sums,
    sum mel_t, sum mel_y, and the number of (mel band, frame) cells per stream."""
    hop = n_fft // 4 if hop is None else hop
    basis = mel_filterbank(sr, n_fft, n_mels).astype(np.float64)
    y = np.asarray(y, np.float64)[:, skip:]
    t = np.asarray(t, np.float64)[:, skip:]
    B, L = y.shape
    win = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n_fft) / n_fft)
    n_frames = 1 + L // hop
    idx = np.arange(n_frames)[:, None] * hop + np.arange(n_fft)[None, :]

    def mel(x):
        xp = np.pad(x, ((0, 0), (n_fft // 2, n_fft // 2)), mode="reflect")
        X = np.fft.rfft(xp[:, idx] * win, axis=-1)
        return (X.real ** 2 + X.imag ** 2) @ basis.T                          # [1, frames, mels]

    out = np.empty((B, 4))
    for b in range(B):
        my, mt = mel(y[b:b + 1]), mel(t[b:b + 1])
        out[b] = [np.abs(my - mt).sum(),
                  np.abs(np.log10(np.maximum(my, log_floor)) - np.log10(np.maximum(mt, log_floor))).sum(), mt.sum(), my.sum()]
    return out, n_frames * n_mels


def ms_spec_losses(y, t, scales=SPEC_SCALES):
    """`ms_spec_loss` and `ms_log_spec_loss` of val_loss_supervised.forward (code/evaluation.py:75-84) over the
    whole batch: SUM over the scales of the mean absolute difference of the power spectrograms / of their
    log10 (clamped at 1e-5)."""
    lin = log = 0.0
    for n_fft in scales:
        s, cells = spec_sums(y, t, 0, n_fft)
        lin += s[:, 0].sum() / (cells * len(s))
        log += s[:, 1].sum() / (cells * len(s))
    return lin, log


def _lin_interp(xk, yk, xn):
    """scipy.interpolate.interp1d(kind='linear')._call_linear restated: knots xk (increasing), values yk [..., K]."""
    idx = np.clip(np.searchsorted(xk, xn), 1, len(xk) - 1)
    lo, hi = idx - 1, idx
    slope = (yk[..., hi] - yk[..., lo]) / (xk[hi] - xk[lo])
    return slope * (xn - xk[lo]) + yk[..., lo]


def demodulate(output, x_idx_pulse, y_idx_pulse):
    """DelayAnalyzer.demodulate, code/utilities/utilities.py:408-465 (fp64): undo the time-varying delay of a
    (C, N) recording from its input / output pulse indices.  Pinned by golden g11 (the reference's own output)."""
    output = np.asarray(output)                 # dtype kept: with float32 knot values (what code/dataset.py:397 passes)
    if output.dtype not in (np.float32, np.float64):     # scipy forms y_hi - y_lo in float32, then a float64 slope
        output = output.astype(np.float64)
    x_idx, y_idx = np.asarray(x_idx_pulse).reshape(-1), np.asarray(y_idx_pulse).reshape(-1)
    period = int(np.mean(np.diff(x_idx)))                                   # :436
    y_hat_idx = y_idx[0] + np.arange(len(y_idx)) * period                   # :437-438  (Eq. 57)
    t = np.arange(output.shape[-1])
    y_hat = _lin_interp(y_idx.astype(np.float64), y_hat_idx.astype(np.float64), t.astype(np.float64))   # :441-447, extrapolating
    dem = _lin_interp(y_hat, output, t.astype(np.float64))                  # :450-455
    dem[:, t < y_hat[0]] = output[:, :1]                                    # fill_value = (output[:, 0], output[:, 1]):
    dem[:, t > y_hat[-1]] = output[:, 1:2]                                  # the "above" value is SAMPLE 1, as upstream
    shift = int(y_idx[0] - x_idx[0])
    if shift > 0:                                                           # :458-465
        dem = np.roll(dem, -shift, axis=1)
        dem[:, -shift:] = 0.0
    return dem


TAPE_PARAMS = (1.6e6, 1.1e3, 1.6e-3, 4.0e2, 1.7e-1)      # Ms, A, alpha, K, c  (code/tape.py:251-256)


def tape_hmag(H, state=None, Ts=1.0 / (48000 * 16), params=TAPE_PARAMS):
    """Tape.H_mag, code/tape.py:516-551.  H [B,N] f64 -> (M [B,N], state [B,3] = M_prev,H_prev,Hprime_prev)."""
    H = np.ascontiguousarray(H, dtype=np.float64)
    B, N = H.shape
    state = np.zeros((B, 3)) if state is None else np.ascontiguousarray(state, dtype=np.float64).copy()
    M = np.empty_like(H)
    par = np.asarray(params, dtype=np.float64)
    d = lambda a: a.ctypes.data_as(_f64p)                                                    # noqa: E731
    assert lib().ntmo_tape_hmag(d(H), d(M), B, N, d(state), Ts, d(par)) == 0
    return M, state


def sinc_resample(x, orig_freq, new_freq, lowpass_filter_width=6, rolloff=0.99):
    """torchaudio.transforms.Resample(orig, new, dtype=float64) as code/tape.py:330-332,471-474,553-558 uses it (sinc
    interpolation with a Hann window).  torchaudio is un-vendored AND absent from the build container: this restates
    its published `_get_sinc_resample_kernel` / `_apply_sinc_resample_kernel` -- PARITY UNPINNED; tests check DC gain,
    a band-limited sine and the round trip instead.  x [B,N] float64 -> [B, ceil(new N / orig)]."""
    import math
    x = np.asarray(x, np.float64)
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    B, N = x.shape
    out_len = int(math.ceil(new * N / orig))
    xp = np.pad(x, ((0, 0), (width, width + orig)))
    frames = (xp.shape[1] - (2 * width + orig)) // orig + 1
    y = np.zeros((B, frames * new))
    for p in range(new):
        for k in range(2 * width + orig):
            t = (-p / new + (k - width) / orig) * base
            t = min(max(t, -lowpass_filter_width), lowpass_filter_width)
            wdw = math.cos(t * math.pi / lowpass_filter_width / 2) ** 2
            tt = t * math.pi
            kv = (1.0 if tt == 0 else math.sin(tt) / tt) * wdw * (base / orig)
            y[:, p::new] += kv * xp[:, k:k + (frames - 1) * orig + 1:orig]
    return y[:, :out_len]


def fir_clamp(x, h, clamp=True):
    """torchaudio.functional.lfilter(x, a=[1,0,...], b=h) as code/tape.py:570-571 calls it: a FIR from zero state,
    output clamped to [-1, 1] (lfilter's default).  PARITY UNPINNED (torchaudio absent)."""
    x = np.asarray(x, np.float64)
    y = np.stack([np.convolve(r, np.asarray(h, np.float64))[:x.shape[1]] for r in x])
    return np.clip(y, -1.0, 1.0) if clamp else y


def tcn_forward(params, L, C, K, dil, x, threads=1):
    x = _c(x)
    B, T = x.shape
    y = np.empty_like(x)
    params = _c(params)
    dil = np.ascontiguousarray(dil, dtype=np.int32)
    args = [_p(params), L, C, K, dil.ctypes.data_as(_i32p), _p(x), _p(y), B, T]
    assert (lib().ntmo_tcn_forward_mt(*args, threads) if threads > 1 else lib().ntmo_tcn_forward(*args)) == 0
    return y


# ----------------------------------------------------------------------------- numpy restatement
def np_gru_forward(w, x, h=None):
    """Pure-numpy fp32 restatement (small cases only)."""
    x = np.asarray(x, np.float32)
    B, T = x.shape
    H = w.H
    h = np.zeros((B, H), np.float32) if h is None else np.asarray(h, np.float32).copy()
    y = np.empty((B, T), np.float32)
    sig = lambda v: (np.float32(1) / (np.float32(1) + np.exp(-v))).astype(np.float32)  # noqa: E731
    for t in range(T):
        gi = x[:, t:t + 1] * w.w_ih[None, :] + w.b_ih[None, :]
        gh = h @ w.w_hh.T + w.b_hh[None, :]
        r = sig(gi[:, :H] + gh[:, :H])
        z = sig(gi[:, H:2 * H] + gh[:, H:2 * H])
        n = np.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:]).astype(np.float32)
        h = ((h - n) * z + n).astype(np.float32)
        y[:, t] = h @ w.w_o + (w.b_o[0] if w.b_o is not None else np.float32(0))
    return y, h


def np_delay_forward(x, d, buf, warmup=False):
    """The reference's own O(T*D) formulation (code/model.py:287-315) in numpy, for tiny cases."""
    x, d, buf = (np.asarray(a, np.float32) for a in (x, d, buf))
    B, T = x.shape
    D = buf.shape[1]
    assert D >= d.max()
    xp = np.concatenate([buf, x], axis=1)
    nb = np.concatenate([buf[:, T:], x[:, -D:] if D > 0 else x[:, :0]], axis=1)
    if warmup:
        return x.copy(), nb
    unf = np.lib.stride_tricks.sliding_window_view(xp, D + 1, axis=1)        # [B,T,D+1]
    dist = np.linspace(D, 0, D + 1, dtype=np.float32)
    wts = np.maximum(np.float32(1) - np.abs(dist[None, None, :] - d[:, :, None]), np.float32(0))
    y = (wts * unf).astype(np.float32).sum(axis=2, dtype=np.float32)
    return y, nb


# ----------------------------------------------------------------------------- torch CPU port
def torch_gru_port(w):
    """torch.nn.GRU(1,H,batch_first=True)+Linear(H,1) loaded with `w`: exactly the modules the
    reference builds at code/model.py:44-45 (and :364-365), on CPU.  Returns f(x[B,T], h0[B,H]) ->
    (y, h)."""
    import torch

    gru = torch.nn.GRU(1, w.H, batch_first=True)
    lin = torch.nn.Linear(w.H, 1, bias=w.b_o is not None)
    with torch.no_grad():
        gru.weight_ih_l0.copy_(torch.from_numpy(w.w_ih).view(3 * w.H, 1))
        gru.weight_hh_l0.copy_(torch.from_numpy(w.w_hh))
        gru.bias_ih_l0.copy_(torch.from_numpy(w.b_ih))
        gru.bias_hh_l0.copy_(torch.from_numpy(w.b_hh))
        lin.weight.copy_(torch.from_numpy(w.w_o).view(1, w.H))
        if w.b_o is not None:
            lin.bias.copy_(torch.from_numpy(w.b_o))
    gru.eval()
    lin.eval()

    def f(x, h0=None):
        with torch.inference_mode():
            xt = torch.as_tensor(x, dtype=torch.float32)
            B, T = xt.shape
            h = None if h0 is None else torch.as_tensor(h0, dtype=torch.float32).view(1, B, w.H)
            o, hn = gru(xt.reshape(B, T, 1), h)
            return lin(o).reshape(B, T).numpy(), hn.view(B, w.H).numpy()

    return f
