"""Timing of the small-hidden-size GRU kernel (K1s, H = 8 / 16 / 32) at B x T (default 4096 x 8192), random weights from the
reference-style initialisation; one line per H with ns per step and the fp32 fraction of its own flop count."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ntm_amd
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
g = torch.Generator(device="cuda"); g.manual_seed(1)
x = torch.rand(B, 1, T, generator=g, device="cuda") - 0.5
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
out = {"B": B, "T": T}
for H in (8, 16, 32):
    torch.manual_seed(H)
    m = ntm_amd.RNN(1, H, 1).to("cuda").eval()
    ms = []
    for i in range(6):
        m.initialize_hidden()
        ev[0].record(); y = m(x); ev[1].record(); torch.cuda.synchronize()
        if i: ms.append(ev[0].elapsed_time(ev[1]))
    t = min(ms)
    flop = 2 * (3 * H * H + 3 * H + H)
    out[f"H{H}"] = {"ms": t, "samples_per_s": B * T / t * 1e3, "ns_per_step": t * 1e6 / T, "frac_of_fp32_peak": B * T * flop / (t * 1e-3) / 157.3e12}
print(json.dumps(out))
