"""Timing of the GRU kernels for hidden sizes other than 64 (K1s: H = 8 / 16 / 32; round 5: any H -- zero-padded below 64, a
workgroup per stream above) at B x T (default 4096 x 8192; the plain kernel for H > 128 at T / 16), random weights from the
reference-style initialisation; per H: ns per step and the fp32 fraction of its own flop count.
usage: python tools/small_probe.py [B] [T] [H,H,...]"""
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
HS = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [8, 16, 32]
for H in HS:
    torch.manual_seed(H)
    m = ntm_amd.RNN(1, H, 1).to("cuda").eval()
    Th = T if H <= 128 else max(64, T // 16)
    xh = x[:, :, :Th]
    ms = []
    for i in range(4):
        m.initialize_hidden()
        ev[0].record(); y = m(xh); ev[1].record(); torch.cuda.synchronize()
        if i: ms.append(ev[0].elapsed_time(ev[1]))
    t = min(ms)
    flop = 2 * (3 * H * H + 3 * H + H)
    out[f"H{H}"] = {"T": Th, "ms": t, "samples_per_s": B * Th / t * 1e3, "ns_per_step": t * 1e6 / Th, "frac_of_fp32_peak": B * Th * flop / (t * 1e-3) / 157.3e12}
print(json.dumps(out))
