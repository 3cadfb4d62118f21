#!/usr/bin/env python3
"""Soak test of the fused DiffDelRNN step's read-back of its own pre_d output (visibility inside the workgroup: stores at
phase 2, vmcnt(0) at phase 34, barrier, loads at phase 36): many full-size launches with delay trajectories whose taps are
the NEWEST samples (k = 0 .. 3: written one phase earlier by other waves of the workgroup), mixed with wow / white /
history-heavy ones, every output compared bit for bit with the GRU launch + streaming delay pass.
usage: python tools/fused_soak.py [iterations] [B] [T]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, ntm_amd
from ntm_amd import weights
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
T = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
dev = torch.device("cuda", 0)
D = 1847
mods = {}
for mode in ("fused", "two_pass"):
    m = ntm_amd.DiffDelRNN(1, 64, 1, max_delay=D - 1)
    m.load_state_dict(weights.load_state_dict(weights.W_DIFFDEL))
    m = m.to(dev).eval()
    m.delay_mode = mode
    if mode == "two_pass":
        m.kernel_variant = "mfma2"
    mods[mode] = m
g = torch.Generator(device=dev)
bad = 0
n = torch.arange(T, device=dev, dtype=torch.float32).unsqueeze(0)
for it in range(iters):
    g.manual_seed(1000 + it)
    x = (torch.rand(B, 1, T, generator=g, device=dev) - 0.5)
    kind = it % 4
    if kind == 0:      # taps = the newest samples
        d = (1.6 * torch.sin(n / (3.0 + 20 * torch.rand(B, 1, generator=g, device=dev)))).abs() + 0.4 * torch.rand(B, 1, generator=g, device=dev)
    elif kind == 1:    # wow around a few tiles back
        d = 100.0 + 90.0 * torch.sin(n / (200.0 + 2000 * torch.rand(B, 1, generator=g, device=dev)) + 6 * torch.rand(B, 1, generator=g, device=dev))
    elif kind == 2:    # white over the whole delay line (general form everywhere)
        d = torch.rand(B, T, generator=g, device=dev) * D
    else:              # integer part hops by several samples inside a 4-sample group
        d = 5.0 + 60.0 * ((n.long() % 4 == (it % 3)).float()) + torch.rand(B, 1, generator=g, device=dev)
    d = d.clamp_(0, D).unsqueeze(1).contiguous()
    h0 = (torch.rand(1, B, 64, generator=g, device=dev) - 0.5) * 0.6
    b0 = (torch.rand(B, 1, D, generator=g, device=dev) - 0.5) * 0.6
    out = {}
    for mode, m in mods.items():
        m.initialize_hidden(B, D - 1)
        m.hidden, m.diffdel.buffer = h0.clone(), b0.clone()
        y, pre = m(x, d)
        out[mode] = (y, pre, m.hidden, m.diffdel.buffer)
    ok = all(torch.equal(a, b) for a, b in zip(out["fused"], out["two_pass"]))
    bad += not ok
    print(f"iteration {it} kind {kind}: {'identical' if ok else 'MISMATCH'}", flush=True)
    del out, x, d
print(f"fused_soak: {iters} launches of {B} x {T}, mismatches: {bad}")
sys.exit(1 if bad else 0)
