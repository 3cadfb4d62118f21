#!/usr/bin/env python3
"""Event-timed TCN forward (BASELINE configs[3] shape by default) for kernel A/B builds: NTM_LIB_PATH selects the
library.  Prints one JSON line: ms per forward (median of --reps), samples/s, fraction of the fp32 matrix peak.
    python tools/tcn_probe.py [--batch 4096] [--samples 65536] [--reps 5]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ntm_amd  # noqa: E402
from ntm_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--samples", type=int, default=65536)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--dilations", default="1,10,100,1000", help="the frozen spec by default; e.g. 1,2,4,8 or 1,1,1,1")
a = ap.parse_args()
dev = torch.device("cuda", 0)
dil = tuple(int(v) for v in a.dilations.split(","))
model = ntm_amd.TCN(dilations=dil, seed=4321).to(dev)
g = torch.Generator(device=dev); g.manual_seed(1)
x = (torch.rand(a.batch, 1, a.samples, device=dev, generator=g) - 0.5)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ms = []
for i in range(a.reps + 1):
    ev0.record(); y = model(x); ev1.record(); torch.cuda.synchronize()
    if i:
        ms.append(ev0.elapsed_time(ev1))
fl = 2 * (13 * 32 + 32 + 3 * (32 * 13 * 32 + 32 * 32) + 32)
m = float(np.median(ms))
print(json.dumps({"lib": os.path.basename(_lib.LIB_PATH), "dilations": dil, "batch": a.batch, "samples": a.samples, "ms": m, "ms_all": ms,
                  "samples_per_s": a.batch * a.samples / m * 1e3, "frac_fp32_peak": fl * a.batch * a.samples / m / 1e9 / 157.3,
                  "y_abs_mean": float(y.abs().mean())}))
