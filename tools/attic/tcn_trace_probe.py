#!/usr/bin/env python3
"""Per-workgroup trace of the stamped tcn_block_pg_kernel<false> launch (libntm_lab.so): start / end s_memtime, CU identity and
cycles inside the iteration loop for EVERY workgroup -> lifetime statistics, gaps between successive workgroups of a CU,
workgroups resident per CU.  TCN 1 / 1000 / 1000 at 4096 x 65 536 (the middle block is traced).  One JSON line."""
import ctypes
import json
import os
import sys

import numpy as np
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
nwg = B * 32
trace = torch.zeros(nwg * 4, dtype=torch.int64, device="cuda")
assert lab.ntm_lab_tcn_trace(ptr(trace)) == 0
for _ in range(2):
    assert lab.ntm_lab_tcn_forward(ptr(params), 3, 32, 13, dil, ptr(x), ptr(y), B, T, ptr(scratch), _lib.current_stream()) == 0
torch.cuda.synchronize()
assert lab.ntm_lab_tcn_trace(None) == 0
tr = trace.cpu().numpy().reshape(nwg, 4)
t0, t1, hw, loop = tr[:, 0], tr[:, 1], tr[:, 2], tr[:, 3]
life = (t1 - t0).astype(np.float64)
cu = ((hw >> 32) & 15) * 256 + ((hw & 0xffffffff) >> 8 & 255)          # XCC, SE/SH/CU bits of HW_ID
span = float(t1.max() - t0.min())
gaps, resident = [], []
for c in np.unique(cu):
    idx = np.where(cu == c)[0]
    order = idx[np.argsort(t0[idx])]
    ends = np.sort(t1[idx])
    # gap between a workgroup's start and the latest end before it that has not been "used" (two slots per CU)
    s, e = t0[order], np.sort(t1[idx])
    if len(s) > 2:
        gaps.append(float(np.mean(s[2:] - e[:-2])))
    resident.append(float(life[idx].sum() / max(1.0, float(t1[idx].max() - t0[idx].min()))))
print(json.dumps({"workgroups": int(nwg), "cus_seen": int(len(np.unique(cu))), "kernel_span_cycles": span,
                  "lifetime_cycles": {"mean": float(life.mean()), "p5": float(np.percentile(life, 5)), "p95": float(np.percentile(life, 95))},
                  "loop_cycles": {"mean": float(loop.mean()), "p5": float(np.percentile(loop, 5)), "p95": float(np.percentile(loop, 95))},
                  "outside_loop_cycles_mean": float((life - loop).mean()),
                  "mean_gap_between_end_and_next_start_on_a_cu_cycles": float(np.mean(gaps)),
                  "mean_workgroups_resident_per_cu": float(np.mean(resident)),
                  "sum_lifetimes_over_span_per_cu": float(life.sum() / span / len(np.unique(cu)))}))
