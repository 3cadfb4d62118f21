#!/usr/bin/env python3
"""Dev micro-benchmark (lives under tests/ because it checks against the oracle): GRU kernel time at B x T (default 4096 x 8192) + max error vs the oracle.
usage: [NTM_LIB_PATH=...] python tests/quick_bench.py [--variant mfma] [--B 4096] [--T 8192] [--iters 10]"""
import argparse, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ntm_amd, oracle
from helpers import oracle_weights
ap = argparse.ArgumentParser()
ap.add_argument("--variant", default="mfma"); ap.add_argument("--B", type=int, default=4096)
ap.add_argument("--T", type=int, default=8192); ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
m = ntm_amd.harness.build_model(ntm_amd.weights.W_GRU); m.kernel_variant = a.variant
g = torch.Generator(device="cuda"); g.manual_seed(1)
x = (torch.rand(a.B, 1, a.T, generator=g, device="cuda") - 0.5)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ts = []
for i in range(a.iters + 2):
    m.initialize_hidden(); m.warm_start(); m.hidden = m.hidden.expand(1, a.B, 64).contiguous()
    ev[0].record(); y = m(x); ev[1].record(); torch.cuda.synchronize()
    if i >= 2: ts.append(ev[0].elapsed_time(ev[1]))
n = min(16, a.B); tt = min(2048, a.T)
yo, _ = oracle.gru_predict(oracle_weights(ntm_amd.weights.W_GRU), x[:n, 0, :tt].cpu().numpy(), threads=8)
err = float(np.abs(y[:n, 0, :tt].cpu().numpy() - yo).max())
ts = np.array(ts)
sps = a.B * a.T / (ts.min() * 1e-3)
print(f"lib={os.path.basename(ntm_amd._lib.LIB_PATH)} variant={a.variant} B={a.B} T={a.T} min={ts.min():.3f} ms "
      f"med={np.median(ts):.3f} ms  {sps/1e9:.3f} Gsamples/s  fp32frac={sps*25088/157.3e12:.3f}  "
      f"ns/step={ts.min()*1e6/a.T:.1f}  maxerr={err:.2e}")
