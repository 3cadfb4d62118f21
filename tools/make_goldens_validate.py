#!/usr/bin/env python3
"""Golden g18: the reference's own `RNN.validate` / `DiffDelRNN.validate` (code/model.py:163-216, :513-616), the
inference-only batched use of the path that code/train.py:242 calls every epoch.

Dev-only script (imports /root/reference, never travels).  The inputs are REGENERATED from seeds by the test
(numpy's default_rng is stable), so the fixture holds only what the reference returned:
  val_loss (python float) and, per batch, the example's prediction (row 0) -- for the DiffDel model also the
  pre-delay prediction -- plus the final carried state (hidden, delay buffer).

A "dataloader" here is what the methods actually use of one: len(), iteration over (input, target, meta) batches and,
for DiffDelRNN, `.dataset.delay_analyzer.max_delay` (seconds), `.dataset.fs`, `meta['delay_trajectory']` (seconds,
(B, T)).  Loss: the ESR of CoreAudioML as a lambda, mean(e^2) / (mean(t^2) + 1e-5).

Usage:  python tools/make_goldens_validate.py
"""
import os
import sys
import types

import numpy as np

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GDIR = os.path.join(ROOT, "tests", "golden")

for m in ["torchaudio", "soundfile", "librosa", "librosa.filters"]:
    sys.modules[m] = types.ModuleType(m)
sys.modules["librosa.filters"].mel = lambda *a, **k: None
sys.modules["librosa"].filters = sys.modules["librosa.filters"]
sys.path.insert(0, os.path.join(REF, "code"))

import torch  # noqa: E402
import model as refmodel  # noqa: E402  (the reference's code/model.py)

torch.set_num_threads(4)
FS = 44100
W_G = "GRU-HS[64]-L[DCPreESR]-DS[ReelToReel_Dataset_MiniPulse100_CHOWTAPE]_BEST"
W_D = "DiffDelGRU-HS[64]-L[DCPreESR]-DS[ReelToReel_Dataset_MiniPulse100_CHOWTAPE_WOWFLUTTER]_BEST"

# ---- the synthetic validation set: shared with tests/ through these constants (stored in the fixture) ----
SEED, N_BATCHES, B, T = 1807, 2, 6, 8192
MAX_DELAY_S = 150.2 / FS           # analyser statistic, seconds -> INIT_LEN = nextpow2(150) = 256
MODEL_MAX_DELAY = 160              # DiffDelRNN(max_delay=...) in samples -> delay line of 161 taps


sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import validate_batches  # noqa: E402  (numpy only: the test regenerates the same inputs)


def make_batches():
    return validate_batches(SEED, N_BATCHES, B, T, FS)


class Loader:
    def __init__(self, batches, with_meta):
        self.batches, self.with_meta = batches, with_meta
        self.dataset = types.SimpleNamespace(delay_analyzer=types.SimpleNamespace(max_delay=MAX_DELAY_S), fs=FS)

    def __len__(self):
        return len(self.batches)

    def __iter__(self):
        for x, t, d in self.batches:
            meta = {"delay_trajectory": torch.from_numpy(d)} if self.with_meta else {}
            yield torch.from_numpy(x), torch.from_numpy(t), meta


def esr(pred, target):
    return ((target - pred) ** 2).mean() / ((target ** 2).mean() + 1e-5)


def main():
    batches = make_batches()
    out = {"seed": SEED, "n_batches": N_BATCHES, "B": B, "T": T, "fs": FS, "max_delay_s": MAX_DELAY_S,
           "model_max_delay": MODEL_MAX_DELAY}

    m = refmodel.RNN(1, 64, 1)
    m.load_state_dict(torch.load(os.path.join(REF, "weights", W_G, "best.pth"), map_location="cpu"))
    val, ex = m.validate(Loader(batches, False), esr)
    out["rnn_val_loss"] = np.float64(val)
    out["rnn_pred"] = np.stack([e["prediction"].numpy() for e in ex])
    out["rnn_hidden"] = m.hidden.numpy()
    assert all(torch.equal(e["input"], torch.from_numpy(b[0][0, 0, 1024:])) for e, b in zip(ex, batches))
    val_ne, ex_ne = m.validate(Loader(batches, False), esr, store_examples=False)
    assert val_ne == val and ex_ne == []

    dm = refmodel.DiffDelRNN(1, 64, 1, max_delay=MODEL_MAX_DELAY)
    dm.load_state_dict(torch.load(os.path.join(REF, "weights", W_D, "best.pth"), map_location="cpu"))
    val, ex = dm.validate(Loader(batches, True), esr)
    out["dd_val_loss"] = np.float64(val)
    out["dd_pred"] = np.stack([e["prediction"].numpy() for e in ex])
    out["dd_pre_d"] = np.stack([e["prediction_pre_d"].numpy() for e in ex])
    out["dd_hidden"] = dm.hidden.numpy()
    out["dd_buffer"] = dm.diffdel.buffer.numpy()

    # detach_hidden / detach_buffer (code/model.py:54-56, :322-324, :377-380): clones with the same values
    h_before, b_before = dm.hidden, dm.diffdel.buffer
    dm.detach_hidden()
    assert dm.hidden is not h_before and torch.equal(dm.hidden, h_before)
    assert dm.diffdel.buffer is not b_before and torch.equal(dm.diffdel.buffer, b_before)

    path = os.path.join(GDIR, "g18_validate.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; rnn val", out["rnn_val_loss"], "dd val", out["dd_val_loss"])


if __name__ == "__main__":
    main()
