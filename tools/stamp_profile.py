#!/usr/bin/env python3
"""Diagnostic: where does one GRU step spend its cycles? (s_memtime stamps, MFMA kernel)"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ntm_amd
from ntm_amd._lib import ptr
B, T = 4096, 4096
m = ntm_amd.harness.build_model(ntm_amd.weights.W_GRU)
x = torch.rand(B, T, device="cuda") - 0.5
y = torch.empty_like(x); h = torch.zeros(B, 64, device="cuda")
st = torch.zeros((B + 15) // 16, 4, 12, dtype=torch.int64, device="cuda")   # [..., 6:12]: the phase-2 (housekeeping) steps alone
g, o = m.GRU, m.output
L = ntm_amd._lib.lib()
LAB = ntm_amd._lib.lab()          # stamps / ablations: diagnostic builds in libntm_lab.so
NAMES = {1: ["lds_read", "phaseA", "phaseB", "tail", "lds_write", "barrier"],
         3: ["mfma1-3(+bookkeeping)", "wait_own_wr", "barrier", "rd+45mfma", "hk+cinit+gates", "wr+head"],
         8: ["between_steps", "hk+x_read", "bar1+rd_hi+split+12mfma", "6mfma(z hi)+head", "20mfma(r,n rest)+seeds+r_sigm", "10mfma+n_gate+z_sigm+publish"]}
for variant in [int(v) for v in os.environ.get('NTM_STAMP_VARIANTS', '3,1,8').split(',')]:
    for _ in range(2):
        rc = LAB.ntm_debug_gru_stamps(ptr(g.weight_ih_l0), ptr(g.weight_hh_l0), ptr(g.bias_ih_l0), ptr(g.bias_hh_l0),
                                    ptr(o.weight), ptr(o.bias), ptr(x), ptr(y), B, T, ptr(h), ptr(st), variant, None)
        assert rc == 0, L.ntm_last_error()
        torch.cuda.synchronize()
    s = st.cpu().numpy().astype(np.float64)[:, :, :6] / T
    print(f"variant {variant}{' (forward + ESR sums)' if os.environ.get('NTM_LAB_STAMP_ESR') else ''}: s_memtime cycles per step, mean over workgroups (stamped build; shares, not totals)")
    for w in range(4):
        print(f"  wave {w}: " + "  ".join(f"{n}={s[:, w, k].mean():7.1f}" for k, n in enumerate(NAMES[variant]))
              + f"  total={s[:, w, :].sum(1).mean():7.1f}")
    if variant == 3:
        hk = st.cpu().numpy().astype(np.float64)[:, :, 6:] / (T // 64)
        print("  the phase-2 step of a tile alone (x-tile fetch, y-tile flush), cycles per occurrence:")
        for w in range(4):
            print(f"  wave {w}: " + "  ".join(f"{n}={hk[:, w, k].mean():7.1f}" for k, n in enumerate(NAMES[3])) + f"  total={hk[:, w, :].sum(1).mean():7.1f}")
