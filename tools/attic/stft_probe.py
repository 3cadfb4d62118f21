"""Timing of the STFT-loss kernels (N1) at the headline batch: 4096 streams x 65536 samples, INIT_LEN 1024."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ntm_amd
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
g = torch.Generator(device="cuda").manual_seed(1)
t = 0.3 * torch.randn(B, 1, T, device="cuda", generator=g)
y = t + 0.02 * torch.randn(B, 1, T, device="cuda", generator=g)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
tot = 0.0
for n_fft, hop, win in [(1024, 120, 600), (2048, 240, 1200), (512, 50, 240)]:
    for i in range(3):
        ev[0].record(); s, cells = ntm_amd.stft_sums(y, t, 1024, n_fft, hop, win); ev[1].record(); torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]); tot += ms
    frames = cells // (n_fft // 2 + 1)
    import math
    flop = B * frames * 5 * n_fft * math.log2(n_fft)
    print(f"stft {n_fft}/{hop}/{win}: {ms:.2f} ms  {B*T/ms/1e6:.2f} Gsamples/s  {B*frames/ms/1e3:.1f} Mframe-pairs/s  ~{flop/ms/1e9:.1f} TFLOP/s(5NlogN)  HBM {2*B*T*4/ms/1e6:.0f} GB/s")
loss = ntm_amd.MRSTFTLoss()
for i in range(2):
    ev[0].record(); v = loss.per_segment(y, t, 1024); ev[1].record(); torch.cuda.synchronize()
print(f"MRSTFTLoss.per_segment: {ev[0].elapsed_time(ev[1]):.2f} ms (kernels {tot:.2f} ms), mean loss {float(v.mean()):.5f}")
