#!/usr/bin/env python3
"""What the loss leg costs the headline step (4096 x 65 536 by default), one process, alternating:
  forward            the recurrent launch alone
  forward_esr        the launch with the ESR sums accumulated in its output flush (ntm_gru_forward_esr)
  forward + pass     the launch, then the streaming ESR pass behind it on the same stream
usage: python tools/esr_fused_probe.py [B] [T] [gru|diffdel]   (diffdel: the fused DiffDelGRU step, sums on the delayed output)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, ntm_amd
from ntm_amd import weights
from ntm_amd.model import esr_sums
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
which = sys.argv[3] if len(sys.argv) > 3 else "gru"
dev = torch.device("cuda", 0)
x = bench.synth_input(B, T, dev, 1234)
if which == "diffdel":
    m = ntm_amd.harness.build_model(weights.W_DIFFDEL, max_delay_seconds=0.0335, device=dev)
    d = bench.delay_trajectories(B, T, dev, m.max_delay)
    tgt = m.predict(x, d)[0].clone()
else:
    m = ntm_amd.harness.build_model(weights.W_GRU, device=dev)
    tgt = m.predict(x).clone()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
def run(kind, n=6):
    ts = []
    for i in range(n + 1):
        if which == "diffdel":
            m.initialize_hidden(1, m.max_delay); m.warm_start(); m.hidden = m.hidden.expand(1, B, 64).contiguous()
            m.diffdel.buffer = m.diffdel.buffer.expand(B, 1, -1).contiguous(); m.diffdel.defer_check = True
            ev[0].record()
            if kind == "forward": y = m.forward(x, d)[0]
            elif kind == "forward_esr": y, _, s = m.forward_esr(x, d, tgt, 1024)
            else: y = m.forward(x, d)[0]; s = esr_sums(y, tgt, 1024)
            ev[1].record(); torch.cuda.synchronize(); m.diffdel.defer_check = False; m.diffdel.raise_if_violated()
            if i: ts.append(ev[0].elapsed_time(ev[1]))
            continue
        m.initialize_hidden(); m.warm_start(); m.hidden = m.hidden.expand(1, B, 64).contiguous()
        ev[0].record()
        if kind == "forward": y = m.forward(x)
        elif kind == "forward_esr": y, s = m.forward_esr(x, tgt, 1024)
        else: y = m.forward(x); s = esr_sums(y, tgt, 1024)
        ev[1].record(); torch.cuda.synchronize()
        if i: ts.append(ev[0].elapsed_time(ev[1]))
    return np.mean(ts), np.min(ts)
for rep in range(2):
    for kind in ("forward", "forward_esr", "forward + pass"):
        print("%-16s mean %.3f ms  min %.3f ms" % ((kind,) + run(kind)))
