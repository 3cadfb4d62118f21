#!/usr/bin/env python3
"""Diagnostic: time the MFMA2 kernel with pieces of the step removed (results are wrong on purpose)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ntm_amd
from ntm_amd._lib import ptr
B, T = 4096, 8192
m = ntm_amd.harness.build_model(ntm_amd.weights.W_GRU)
x = torch.rand(B, T, device="cuda") - 0.5
y = torch.empty_like(x)
g, o = m.GRU, m.output
L = ntm_amd._lib.lib()
LAB = ntm_amd._lib.lab()          # stamps / ablations: diagnostic builds in libntm_lab.so
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
NAMES = {0: "full kernel", 1: "no gates", 2: "no LDS exchange", 4: "no head", 8: "own-quarter MFMAs only (12)", 16: "no barrier",
         32: "no housekeeping", 3: "no gates, no LDS", 7: "no gates/LDS/head", 18: "no LDS, no barrier", 39: "no gates/LDS/head/hk",
         55: "no gates/LDS/head/hk/barrier", 63: "everything off but 12 MFMAs + cinit"}
def run(mask):
    ts = []
    for i in range(5):
        h = torch.zeros(B, 64, device="cuda")
        ev[0].record()
        if mask == 0:
            rc = L.ntm_gru_forward_ex(ptr(g.weight_ih_l0), ptr(g.weight_hh_l0), ptr(g.bias_ih_l0), ptr(g.bias_hh_l0),
                                      ptr(o.weight), ptr(o.bias), 64, ptr(x), ptr(y), B, T, T, T, ptr(h), 3, None)
        else:
            rc = LAB.ntm_debug_gru_ablate(ptr(g.weight_ih_l0), ptr(g.weight_hh_l0), ptr(g.bias_ih_l0), ptr(g.bias_hh_l0),
                                        ptr(o.weight), ptr(o.bias), ptr(x), ptr(y), B, T, ptr(h), mask, None)
        assert rc == 0, L.ntm_last_error()
        ev[1].record(); torch.cuda.synchronize()
        ts.append(ev[0].elapsed_time(ev[1]))
    return min(ts[1:])
base = None
for mask in [0, 1, 2, 4, 8, 16, 32, 3, 7, 18, 39, 55, 63]:
    t = run(mask)
    base = base or t
    print(f"mask {mask:2d} {NAMES[mask]:40s} {t:7.3f} ms  {t*1e6/T:7.1f} ns/step  delta vs full {(t-base)*1e6/T:+7.1f} ns/step")
