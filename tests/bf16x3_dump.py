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
# larger batches as bit-pattern checksums (sum and xor of the int32 views, per stream): every sample of every stream counts,
# the files stay small -- the headline shape's own generator, one group per CU (4096) and two per CU (8200: YPN = 4)
sys.path.insert(0, ROOT)
import bench  # noqa: E402
for B, T in ((4096, 16384), (8200, 4096)):
    x = bench.synth_input(B, T, torch.device("cuda", 0), seed=1234)
    y = m.predict(x)
    bits = y.view(torch.int32)[:, 0].to(torch.int64)
    out[f"sum_{B}_{T}"] = bits.sum(dim=1).cpu().numpy()
    folded = bits
    while folded.shape[1] > 1:
        half = folded.shape[1] // 2
        folded = torch.bitwise_xor(folded[:, :half], folded[:, half:2 * half]) if folded.shape[1] % 2 == 0 else torch.cat(
            [torch.bitwise_xor(folded[:, :half], folded[:, half:2 * half]), folded[:, 2 * half:]], dim=1)
    out[f"xor_{B}_{T}"] = folded[:, 0].cpu().numpy()
    out[f"hsum_{B}_{T}"] = m.hidden.view(torch.int32).to(torch.int64).sum(dim=2).cpu().numpy()
    del x, y, bits, folded
# optional soak (NTM_DUMP_SOAK=n, not part of the suite's default): n seeded random shapes and states, checksums only
n_soak = int(os.environ.get("NTM_DUMP_SOAK", "0"))
rng = np.random.default_rng(2026)
for k in range(n_soak):
    B = int(rng.choice([1, 7, 16, 17, 100, 1040, 2500, 4096, 4112, 5000, 8200]))
    T = int(rng.integers(1, 700))
    g = torch.Generator(device="cuda").manual_seed(10_000 + k)
    x = torch.rand(B, 1, T, generator=g, device="cuda") * 1.6 - 0.8
    m.hidden = (torch.rand(1, B, 64, generator=g, device="cuda") * 1.8 - 0.9) * float(rng.choice([1.0, 1e-3]))
    y = m(x)
    out[f"soak{k}_{B}_{T}"] = np.concatenate([y.view(torch.int32).to(torch.int64).sum(dim=2)[:, 0].cpu().numpy(),
                                              m.hidden.view(torch.int32).to(torch.int64).sum(dim=2)[0].cpu().numpy()])
np.savez(sys.argv[1], **out)
print("lib", os.path.basename(ntm_amd._lib.LIB_PATH), "ok")
