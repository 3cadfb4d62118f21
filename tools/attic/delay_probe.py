"""Timing of the delay-line pass K2 alone (ntm_delay_forward) at the BASELINE configs[2] shape: 4096 streams x 65536
samples, D = 1847, wow-and-flutter trajectories; event-timed, prints one JSON line with the HBM roofline
(12 algorithmic bytes per sample: x and d read once, y written once; + 8 D bytes per stream for the carried buffer)."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ntm_amd
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
D = int(sys.argv[3]) if len(sys.argv) > 3 else 1847
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(5)
x = 0.3 * torch.randn(B, 1, T, device=dev, generator=g)
n = torch.arange(T, device=dev, dtype=torch.float32).unsqueeze(0)
amp = 0.002 + 0.003 * torch.rand(B, 1, generator=g, device=dev)
wv = 0.5 + 1.5 * torch.rand(B, 1, generator=g, device=dev)
d = torch.empty(B, T, device=dev)
for b0 in range(0, B, 256):
    sl = slice(b0, min(B, b0 + 256))
    d[sl] = 44100 * (0.0271 + amp[sl] * torch.sin(2 * np.pi * wv[sl] * n / 44100) + 0.0005 * torch.sin(2 * np.pi * 23 * n / 44100))
d = d.clamp_(0, D - 1).unsqueeze(1)
dl = ntm_amd.TimeVaryingDelayLine(max_delay=D)
dl.init_buffer(B, D)
dl.defer_check = True
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ms = []
for i in range(8):
    ev[0].record(); y = dl(x, d); ev[1].record(); torch.cuda.synchronize()
    if i >= 2: ms.append(ev[0].elapsed_time(ev[1]))
dl.raise_if_violated()
# yardstick with the same traffic shape (two streams read, one written, 12 B per sample): torch's elementwise add
y2 = torch.empty_like(x)
ms2 = []
for i in range(8):
    ev[0].record(); torch.add(x, d, out=y2); ev[1].record(); torch.cuda.synchronize()
    if i >= 2: ms2.append(ev[0].elapsed_time(ev[1]))
m = float(np.mean(ms))
alg = 12.0 * B * T + 8.0 * B * D
print(json.dumps({"B": B, "T": T, "D": D, "ms": m, "ms_min": min(ms), "algorithmic_bytes": alg, "GBps": alg / m / 1e6,
                  "frac_of_8TBps": alg / m / 1e6 / 8000.0, "frac_at_16B_per_sample": 16.0 * B * T / m / 1e6 / 8000.0,
                  "yardstick_torch_add_same_traffic_ms": float(np.mean(ms2)), "yardstick_GBps": 12.0 * B * T / float(np.mean(ms2)) / 1e6}))
