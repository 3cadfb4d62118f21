"""Timing of the loss-dict kernels (ESR, DC-pre-emphasised ESR, multi-resolution STFT) at the headline batch:
4096 streams x 65536 samples, INIT_LEN 1024.  Event-timed per kernel; prints one JSON line with the HBM roofline of
the two streaming kernels (8 algorithmic bytes per sample: y and t read once)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ntm_amd
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
g = torch.Generator(device="cuda").manual_seed(1)
t = 0.3 * torch.randn(B, 1, T, device="cuda", generator=g)
y = t + 0.02 * torch.randn(B, 1, T, device="cuda", generator=g)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
out = {"B": B, "T": T}
for name, fn in (("esr_sums_kernel", lambda: ntm_amd.esr_sums(y, t, 1024)),
                 ("esr_dcpre_kernel", lambda: ntm_amd.esr_dcpre_sums(y, t, 1024)),
                 ("MRSTFTLoss", lambda: ntm_amd.MRSTFTLoss().per_segment(y, t, 1024))):
    ms = []
    for i in range(6):
        ev[0].record(); r = fn(); ev[1].record(); torch.cuda.synchronize()
        if i: ms.append(ev[0].elapsed_time(ev[1]))
    m = sum(ms) / len(ms)
    out[name] = {"ms": m, "GBps_8B_per_sample": 8.0 * B * (T - 1024) / m / 1e6, "frac_of_8TBps": 8.0 * B * (T - 1024) / m / 1e6 / 8000.0}
print(json.dumps(out))
