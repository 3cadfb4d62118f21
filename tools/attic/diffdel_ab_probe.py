#!/usr/bin/env python3
"""A/B of the DiffDelRNN step through the C ABI at BASELINE configs[2] (4096 x 65 536, D = 1847): GRU launch + streaming delay
pass (NTM_DIFFDEL_TWO_PASS) against the fused launch (NTM_DIFFDEL_FUSED), and the fused launch with its delay work switched
off (sticky flag preset / warm-up mode) -- what the fused interpolation itself costs.  usage: python tools/diffdel_ab_probe.py"""
import sys, ctypes, numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench, ntm_amd
from ntm_amd import weights, _lib
L = _lib.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
D = 1847
dev = torch.device("cuda", 0)
x = bench.synth_input(B, T, dev, 1234)[:, 0].contiguous()
w = {k: v.cuda() for k, v in weights.load_state_dict(weights.W_DIFFDEL).items()}
d = bench.delay_trajectories(B, T, dev, D - 1)[:, 0].contiguous()
y, pre = torch.empty_like(x), torch.empty_like(x)
h = torch.zeros(B, 64, device=dev); buf = torch.zeros(B, D, device=dev)
flag = torch.zeros(1, dtype=torch.int32, device=dev)
p = lambda t: ctypes.c_void_p(t.data_ptr())
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
def t(mode, warmup=0, preset=0, dd=None, n=5):
    ts = []
    for i in range(n + 1):
        flag.fill_(preset); h.zero_()
        ev[0].record()
        rc = L.ntm_diffdel_gru_forward_ex(p(w["GRU.weight_ih_l0"]), p(w["GRU.weight_hh_l0"]), p(w["GRU.bias_ih_l0"]), p(w["GRU.bias_hh_l0"]),
                                          p(w["output.weight"]), 64, p(x), p(d if dd is None else dd), p(y), p(pre), B, T, p(h), p(buf), D, warmup, p(flag), mode, None)
        ev[1].record(); torch.cuda.synchronize(); assert rc == 0
        if i: ts.append(ev[0].elapsed_time(ev[1]))
    return np.mean(ts)
print("two_pass            %.3f" % t(1))
print("fused               %.3f" % t(2))
print("fused, flag preset  %.3f (no delay work: structure overhead only)" % t(2, preset=1))
print("fused, warmup       %.3f (y = pre_d stored at the flush)" % t(2, warmup=1))
dc = torch.full_like(d, 1200.25)
print("fused, constant d   %.3f" % t(2, dd=dc))
print("two_pass            %.3f" % t(1))
print("fused               %.3f" % t(2))
