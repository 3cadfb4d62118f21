#!/usr/bin/env python3
"""Child process of tests/test_gpu_round6.py: the bf16x3 engine's output and final state on seeded inputs, from whichever
library NTM_LIB_PATH names (libntm.so: the hand-scheduled form; libntm_bf16x3c.so: the compiler-scheduled form) -> one .npz.
usage: bf16x3_dump.py <out.npz>"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ntm_amd  # noqa: E402

out = {}
m = ntm_amd.harness.build_model(ntm_amd.weights.W_GRU)
m.kernel_variant = "bf16x3"
for B, T in ((1040, 4096), (4112, 300), (8200, 129), (1040, 1)):
    g = torch.Generator(device="cuda").manual_seed(B + T)
    x = torch.rand(B, 1, T, generator=g, device="cuda") - 0.5
    m.initialize_hidden()
    y1 = m(x[:, :, :T // 2]) if T > 1 else None          # carried state across two calls (odd / even step counts)
    y2 = m(x[:, :, T // 2:])
    out[f"y2_{B}_{T}"] = y2.cpu().numpy()
    if y1 is not None:
        out[f"y1_{B}_{T}"] = y1.cpu().numpy()
    out[f"h_{B}_{T}"] = m.hidden.cpu().numpy()
np.savez(sys.argv[1], **out)
print("lib", os.path.basename(ntm_amd._lib.LIB_PATH), "ok")
