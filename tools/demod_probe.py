"""Timing of ntm_demodulate (DelayAnalyzer.demodulate, code/utilities/utilities.py:408-465) on a long stereo recording:
C = 2 channels x N samples, pulses every `period` samples with a slow wow on the output side.  Algorithmic bytes:
8 per channel-sample (x gathered once, out written once) + the fp64 warped-time scratch (8 written + 8 read per sample)."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ntm_amd
from ntm_amd import feeder
N = int(sys.argv[1]) if len(sys.argv) > 1 else 44100 * 600          # ten minutes
period = 4410
x_idx = np.arange(1000, N - 5000, period)
y_idx = (x_idx + 1200 + 40 * np.sin(2 * np.pi * 0.7 * x_idx / 44100.0)).astype(np.int64)
g = torch.Generator(device="cuda").manual_seed(3)
x = 0.3 * torch.randn(2, N, device="cuda", generator=g)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ms = []
for i in range(6):
    ev[0].record(); o = feeder.demodulate(x, x_idx, y_idx); ev[1].record(); torch.cuda.synchronize()
    if i: ms.append(ev[0].elapsed_time(ev[1]))
m = sum(ms) / len(ms)
alg = 2 * N * 8 + N * 16
print(json.dumps({"C": 2, "N": N, "pulses": int(len(x_idx)), "ms": m, "algorithmic_bytes": alg, "GBps": alg / m / 1e6,
                  "frac_of_8TBps": alg / m / 1e6 / 8000.0}))
