#!/usr/bin/env python3
"""What the two time-domain losses cost the GRU step (4096 x 65 536 by default), one process, alternating:
  forward                      the recurrent launch alone
  forward_esr                  the launch with the ESR sums in its output flush (ntm_gru_forward_esr)
  forward_losses               the launch with the ESR AND the DCPreESR sums in its flush (ntm_gru_forward_losses)
  forward_esr + dcpre pass     the ESR launch, then the streaming DCPreESR pass behind it (what compute_loss issued before)
  forward + both passes        the plain launch, then both streaming passes
usage: python tools/dcp_fused_probe.py [B] [T] [gru|diffdel]   (diffdel: the fused DiffDelGRU step, sums on the delayed output)"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, ntm_amd
from ntm_amd import weights
from ntm_amd.model import esr_sums, esr_dcpre_sums
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
which = sys.argv[3] if len(sys.argv) > 3 else "gru"
dev = torch.device("cuda", 0)
x = bench.synth_input(B, T, dev, 1234)
if which == "diffdel":
    m = ntm_amd.harness.build_model(weights.W_DIFFDEL, max_delay_seconds=0.0335, device=dev)
    dtr = bench.delay_trajectories(B, T, dev, m.max_delay)
    tgt = (0.9 * m.predict(x, dtr)[0] + 0.02 * x + 0.01).contiguous()
else:
    m = ntm_amd.harness.build_model(weights.W_GRU, device=dev)
    tgt = (0.9 * m.predict(x) + 0.02 * x + 0.01).contiguous()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
KINDS = ("forward", "forward_esr", "forward_losses", "forward_esr + dcpre pass", "forward + both passes")
def run(kind, n=6):
    ts = []
    for i in range(n + 1):
        if which == "diffdel":
            m.initialize_hidden(1, m.max_delay); m.warm_start(); m.hidden = m.hidden.expand(1, B, 64).contiguous()
            m.diffdel.buffer = m.diffdel.buffer.expand(B, 1, -1).contiguous(); m.diffdel.defer_check = True
            ev[0].record()
            if kind == "forward": y = m.forward(x, dtr)[0]
            elif kind == "forward_esr": y, _, s = m.forward_esr(x, dtr, tgt, 1024)
            elif kind == "forward_losses": y, _, s, d = m.forward_losses(x, dtr, tgt, 1024)
            elif kind == "forward_esr + dcpre pass": y, _, s = m.forward_esr(x, dtr, tgt, 1024); d = esr_dcpre_sums(y, tgt, 1024)
            else: y = m.forward(x, dtr)[0]; s = esr_sums(y, tgt, 1024); d = esr_dcpre_sums(y, tgt, 1024)
            ev[1].record(); torch.cuda.synchronize(); m.diffdel.defer_check = False; m.diffdel.raise_if_violated()
            if i: ts.append(ev[0].elapsed_time(ev[1]))
            continue
        m.initialize_hidden(); m.warm_start(); m.hidden = m.hidden.expand(1, B, 64).contiguous()
        ev[0].record()
        if kind == "forward": y = m.forward(x)
        elif kind == "forward_esr": y, s = m.forward_esr(x, tgt, 1024)
        elif kind == "forward_losses": y, s, d = m.forward_losses(x, tgt, 1024)
        elif kind == "forward_esr + dcpre pass": y, s = m.forward_esr(x, tgt, 1024); d = esr_dcpre_sums(y, tgt, 1024)
        else: y = m.forward(x); s = esr_sums(y, tgt, 1024); d = esr_dcpre_sums(y, tgt, 1024)
        ev[1].record(); torch.cuda.synchronize()
        if i: ts.append(ev[0].elapsed_time(ev[1]))
    return float(np.mean(ts)), float(np.min(ts))
res = {k: [] for k in KINDS}
for rep in range(3):
    for kind in KINDS:
        mean, mn = run(kind)
        res[kind].append(mean)
        print("%-26s mean %.3f ms  min %.3f ms" % (kind, mean, mn))
out = {"model": which, "B": B, "T": T, "skip": 1024, "mean_ms": {k: float(np.mean(v)) for k, v in res.items()}}
out["saved_ms_vs_esr_launch_plus_dcpre_pass"] = out["mean_ms"]["forward_esr + dcpre pass"] - out["mean_ms"]["forward_losses"]
out["cost_ms_over_esr_launch"] = out["mean_ms"]["forward_losses"] - out["mean_ms"]["forward_esr"]
# same numbers?  y and the ESR sums bit for bit, the DCPreESR sums to the fp32 evaluation order of the filter
if which == "diffdel":
    y1, _, s1, d1 = m.predict_losses(x, dtr, tgt, 1024)
    y2, _, s2 = m.predict_esr(x, dtr, tgt, 1024)
else:
    y1, s1, d1 = m.predict_losses(x, tgt, 1024)
    y2, s2 = m.predict_esr(x, tgt, 1024)
d2 = esr_dcpre_sums(y2, tgt, 1024)
out["y_and_esr_sums_identical"] = bool(torch.equal(y1, y2) and torch.equal(s1, s2))
out["dcpre_sums_max_rel_diff_vs_streaming_pass"] = float(((d1 - d2).abs() / d2.abs()).max())
print(json.dumps(out))
