#!/usr/bin/env python3
"""s_memtime timeline of one iteration of tcn_block_pg_kernel<false> (the DIAGNOSTIC build in libntm_lab.so): where a wave's
time goes with two waves per SIMD (default) and alone on its SIMD (NTM_LAB_TCN_ONE_WG=1).  TCN 1 / 1000 / 1000 at
4096 x 65 536: the middle block is the stamped launch.  One JSON line."""
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ntm_amd  # noqa: E402
from ntm_amd import _lib  # noqa: E402
from ntm_amd._lib import ptr  # noqa: E402

B, T = 4096, 65536
m = ntm_amd.TCN(dilations=(1, 1000, 1000), seed=4321).to("cuda")
x = (torch.rand(B, 1, T, device="cuda") - 0.5).view(B, T).contiguous()
y = torch.empty_like(x)
params = m.packed_params().to("cuda")
lab, lib = _lib.lab(), _lib.lib()
scratch = torch.empty(lib.ntm_tcn_scratch_floats(B, T, 32), device="cuda")
dil = (ctypes.c_int * 3)(*m.dilations)
for _ in range(2):
    rc = lab.ntm_lab_tcn_forward(ptr(params), 3, 32, 13, dil, ptr(x), ptr(y), B, T, ptr(scratch), _lib.current_stream())
    assert rc == 0, rc
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (8 + 3 * 64))()
assert lab.ntm_lab_tcn_stamps(buf) == 0
v = list(buf)
names = ["issue block loads", "MFMA block (224 MFMAs)", "ring stores (vmcnt wait)", "epilogue", "barrier", "loop back-edge"]
n = v[6]
print(json.dumps({"one_wave_per_simd": bool(os.environ.get("NTM_LAB_TCN_ONE_WG")), "iterations": n,
                  "ticks_per_iteration": {k: round(v[i] / n, 1) for i, k in enumerate(names)}, "total": round(sum(v[:6]) / n, 1),
                  "first_iteration_starts_at": v[8], "iteration_cycles": [v[8 + 3 * i + 2] for i in range(n)],
                  "mfma_block_cycles": [v[8 + 3 * i + 1] for i in range(n)],
                  "y_vs_product_max_abs": float((y - m(x.view(B, 1, T)).view(B, T)).abs().max())}))
