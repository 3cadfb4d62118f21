#!/usr/bin/env python3
"""Where a DiffDelGRU predict() spends its time, fused step against GRU launch + delay pass (BASELINE configs[2] shape):
host time of each phase of predict (the launches are asynchronous: a phase that takes milliseconds on the host is a
stall), device time between events, per mode.  usage: python tools/diffdel_probe.py [B] [T] [steps]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, ntm_amd
from ntm_amd import weights
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
dev = torch.device("cuda", 0)
x = bench.synth_input(B, T, dev, 1234)
model = ntm_amd.harness.build_model(weights.W_DIFFDEL, max_delay_seconds=0.0335, device=dev)
d = bench.delay_trajectories(B, T, dev, model.max_delay)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
for mode in ("fused", "two_pass", "fused", "two_pass"):
    model.delay_mode = mode
    rows = []
    for i in range(steps + 2):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); ev[0].record()
        model.initialize_hidden(1, model.max_delay); model.warm_start()
        model.hidden = model.hidden.expand(1, B, 64).contiguous()
        model.diffdel.buffer = model.diffdel.buffer.expand(B, 1, -1).contiguous()
        model.diffdel.defer_check = True
        t1 = time.perf_counter(); ev[1].record()
        y, pre = model.forward(x, d, _events=ev[2:5])
        t2 = time.perf_counter()
        model.diffdel.defer_check = False
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        model.diffdel.raise_if_violated()
        if i >= 2:
            rows.append([1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t0), ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2]),
                         ev[2].elapsed_time(ev[3]), ev[3].elapsed_time(ev[4])])
        del y, pre
    r = np.mean(rows, 0)
    print(f"{mode:9s} host: state {r[0]:.3f} ms, forward call {r[1]:.3f} ms, whole step {r[2]:.3f} ms | device: state {r[3]:.3f}, "
          f"alloc/before launch {r[4]:.3f}, kernel(s) {r[5]:.3f}, second pass {r[6]:.3f}")
