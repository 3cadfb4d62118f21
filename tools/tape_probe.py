import sys, os, torch, numpy as np, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ntm_amd
B, N = 4096, 16384
H = 8000.0 * torch.sin(torch.arange(N, device="cuda", dtype=torch.float64)[None, :] * 0.01 + torch.rand(B, 1, device="cuda", dtype=torch.float64))
tp = ntm_amd.TapeMagnetization(batch_size=B)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for i in range(3):
    ev[0].record(); M = tp.H_mag(H); ev[1].record(); torch.cuda.synchronize()
ms = ev[0].elapsed_time(ev[1])
print(f"tape H_mag: {B} x {N} oversampled samples in {ms:.2f} ms = {B*N/ms/1e6:.2f} G oversampled samples/s = {B*N/16/ms/1e3:.1f} M audio samples/s (16x OS)")
