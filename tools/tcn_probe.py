import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ntm_amd
B, T = 512, 65536
x = (torch.rand(B, 1, T, device="cuda") - 0.5)
for dil in [(1, 1, 1, 1), (1, 10, 100, 1000), (1, 2, 4, 8)]:
    m = ntm_amd.TCN(dilations=dil).to("cuda")
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ts = []
    for i in range(4):
        ev[0].record(); y = m(x); ev[1].record(); torch.cuda.synchronize(); ts.append(ev[0].elapsed_time(ev[1]))
    print(dil, f"{min(ts):.2f} ms  {B*T/min(ts)/1e6:.1f} Msamples/s")
