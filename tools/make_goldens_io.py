#!/usr/bin/env python3
"""Golden g22: the reference's own RNN.forward for input_size / output_size other than 1 (code/model.py:22,44-45,67-88),
run here by importing the reference (torchaudio / soundfile / librosa stubbed as in tools/make_goldens.py).  Per case: the
seeded state_dict, a seeded input (B, input_size, T), forward() in two calls (state carried), the final hidden state; one case
with skip=True (output_size == input_size).  Nothing of the reference travels: only inputs and outputs.
usage: python tools/make_goldens_io.py [/root/reference] -> tests/golden/g22_io_sizes.npz"""
import os
import sys
import types

import numpy as np
import torch

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for name in ("torchaudio", "soundfile", "librosa", "librosa.filters"):
    sys.modules.setdefault(name, types.ModuleType(name))
sys.modules["librosa.filters"].mel = lambda *a, **k: None
sys.modules["librosa"].filters = sys.modules["librosa.filters"]
sys.path.insert(0, os.path.join(REF, "code"))
import model as ref_model  # noqa: E402

CASES = [("i2_h24_o3", 2, 24, 3, False, 3, 150), ("i3_h64_o1", 3, 64, 1, False, 2, 200), ("i1_h16_o2", 1, 16, 2, False, 4, 97),
         ("i2_h8_o2_skip", 2, 8, 2, True, 3, 64), ("i5_h96_o4", 5, 96, 4, False, 2, 40)]
out = {"cases": np.array([c[0] for c in CASES])}
for name, I, H, O, skip, B, T in CASES:
    torch.manual_seed(1000 + I * 100 + H + O)
    m = ref_model.RNN(I, H, O, skip=skip).eval()
    g = torch.Generator().manual_seed(7 + I + H + O)
    x = torch.rand(B, I, T, generator=g) - 0.5
    cut = T // 3
    with torch.no_grad():
        m.initialize_hidden()
        y = torch.cat([m(x[:, :, :cut]), m(x[:, :, cut:])], dim=2)
    for k, v in m.state_dict().items():
        out[f"{name}__{k}"] = v.numpy().copy()
    out[f"{name}__x"], out[f"{name}__y"], out[f"{name}__h"] = x.numpy(), y.numpy(), m.hidden.numpy()[0].copy()
    out[f"{name}__meta"] = np.array([I, H, O, int(skip), cut])
    print(name, tuple(y.shape))
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "g22_io_sizes.npz"), **out)
