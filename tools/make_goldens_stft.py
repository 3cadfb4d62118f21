#!/usr/bin/env python3
"""Golden vectors for the multi-resolution STFT loss of code/test-model.py:25,253 (`MultiResolutionSTFTLoss()`
from the un-vendored `auraloss` submodule -- an empty directory in the reference, so the loss itself cannot be
imported).  What CAN be pinned here is the arithmetic it is made of: `torch.stft` (the call auraloss makes) and
the published formula of auraloss.freq.{STFTLoss, SpectralConvergenceLoss, STFTMagnitudeLoss,
MultiResolutionSTFTLoss} at its defaults, restated below with stock torch ops in fp32, exactly as a caller of
the reference would evaluate it on CPU.  Dev-only, like tools/make_goldens.py.
Inputs: the GRU outputs / inputs of golden G1 (real programme material) and seeded noise-plus-tone pairs.
Output: tests/golden/g10_mrstft.npz"""
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RES = ((1024, 120, 600), (2048, 240, 1200), (512, 50, 240))
EPS = 1e-8


def stft_mag(x, n_fft, hop, win):
    X = torch.stft(x, n_fft, hop, win, torch.hann_window(win), return_complex=True)
    return torch.sqrt(torch.clamp(X.real ** 2 + X.imag ** 2, min=EPS))


def stft_loss_terms(x, y, n_fft, hop, win):
    """x = prediction, y = target, each (1, L).  -> (spectral convergence, log-magnitude L1, linear-magnitude L1)"""
    xm, ym = stft_mag(x, n_fft, hop, win), stft_mag(y, n_fft, hop, win)
    sc = torch.norm(ym - xm, p="fro") / torch.norm(ym, p="fro")
    lg = torch.nn.functional.l1_loss(torch.log(xm), torch.log(ym))
    ln = torch.nn.functional.l1_loss(xm, ym)
    return sc.item(), lg.item(), ln.item()


if __name__ == "__main__":
    g1 = np.load(os.path.join(ROOT, "tests", "golden", "g1_predict_16x8192.npz"))
    rng = np.random.default_rng(77)
    n = np.arange(8192)
    pred = [g1["y"][0], g1["y"][5], g1["y"][11]]
    targ = [g1["x"][0], g1["x"][5], 0.9 * g1["y"][11] + 0.003 * rng.standard_normal(8192)]
    for f in (220.0, 3100.0):
        a = 0.4 * np.sin(2 * np.pi * f * n / 44100) + 0.05 * rng.standard_normal(8192)
        pred.append(np.tanh(1.5 * a) * 0.6 + 0.002 * rng.standard_normal(8192))
        targ.append(a)
    pred = np.stack(pred).astype(np.float32)
    targ = np.stack(targ).astype(np.float32)
    skip = 1000                                   # the harness cuts INIT_LEN samples first (code/test-model.py:367-369)
    terms = np.zeros((len(pred), len(RES), 3))
    for b in range(len(pred)):
        x = torch.from_numpy(pred[b:b + 1, skip:])
        y = torch.from_numpy(targ[b:b + 1, skip:])
        for r, (n_fft, hop, win) in enumerate(RES):
            terms[b, r] = stft_loss_terms(x, y, n_fft, hop, win)
    loss = (terms[:, :, 0] + terms[:, :, 1]).mean(1)          # w_sc = w_log_mag = 1, mean over resolutions
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "g10_mrstft.npz"), pred=pred, targ=targ, skip=skip,
                        res=np.array(RES), terms=terms, loss=loss)
    print(terms, loss)


# ---- second part: the power-spectrogram validation metrics of code/evaluation.py:75-84 (TimeFreqConverter wraps
# torchaudio.transforms.Spectrogram(n_fft, hop_length=n_fft//4), i.e. torch.stft with a periodic Hann window of
# n_fft samples and power 2; torchaudio itself is not installed here) -> tests/golden/g13_ms_spec.npz
def ms_spec_goldens():
    g = np.load(os.path.join(ROOT, "tests", "golden", "g10_mrstft.npz"))
    out, tgt = torch.from_numpy(g["pred"]), torch.from_numpy(g["targ"])
    scales = (2048, 1024, 512, 256, 128, 64)
    lin, log, per_scale = 0.0, 0.0, []
    for n in scales:
        P = lambda x: torch.stft(x, n, n // 4, n, torch.hann_window(n), return_complex=True).abs() ** 2      # noqa: E731
        px, py = P(out), P(tgt)
        a = torch.nn.functional.l1_loss(px, py).item()
        b = torch.nn.functional.l1_loss(torch.log10(torch.clamp(px, 1e-5)), torch.log10(torch.clamp(py, 1e-5))).item()
        per_scale.append((a, b)); lin += a; log += b
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "g13_ms_spec.npz"), scales=np.array(scales),
                        per_scale=np.array(per_scale), ms_spec_loss=lin, ms_log_spec_loss=log)
    print("ms_spec", lin, log, per_scale)


if __name__ == "__main__":
    ms_spec_goldens()
