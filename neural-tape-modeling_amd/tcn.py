"""Builder-defined TCN contrast point (BASELINE.json configs[3]).  The reference contains NO TCN
(code/micro_tcn is an empty, un-vendored submodule: SURVEY.md §0), so spec, weights and oracle are this
build's own; parity is "unpinned" and checked against torch.nn.functional.conv1d on CPU.

Spec (after the public micro-TCN "TCN-300-C" shape): 4 causal blocks, kernel 13, dilations
1/10/100/1000, 32 channels; block: out = PReLU(causal_dilated_conv(in)) + conv1x1(in); a final 1x1 conv
to one channel.  Receptive field 1 + 12*(1+10+100+1000) = 13 333 samples.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import ptr

DEFAULT_DILATIONS = (1, 10, 100, 1000)


class TCN(torch.nn.Module):
    def __init__(self, channels=32, kernel_size=13, dilations=DEFAULT_DILATIONS, seed=4321):
        super().__init__()
        if channels != 32:
            raise ValueError("only 32 channels is compiled")
        self.channels, self.kernel_size, self.dilations = channels, kernel_size, tuple(int(d) for d in dilations)
        g = torch.Generator().manual_seed(seed)
        C, K = channels, kernel_size
        self.conv_weight, self.conv_bias, self.prelu, self.res_weight = (torch.nn.ParameterList() for _ in range(4))
        cin = 1
        for _ in self.dilations:
            k = 1.0 / np.sqrt(cin * K)
            mk = lambda *shape, s=k: torch.nn.Parameter((torch.rand(*shape, generator=g) * 2 - 1) * s,  # noqa: E731
                                                       requires_grad=False)
            self.conv_weight.append(mk(C, cin, K))
            self.conv_bias.append(mk(C))
            self.prelu.append(torch.nn.Parameter(torch.full((C,), 0.25), requires_grad=False))
            self.res_weight.append(mk(C, cin, 1, s=1.0 / np.sqrt(cin)))
            cin = C
        self.out_weight = torch.nn.Parameter((torch.rand(1, C, 1, generator=g) * 2 - 1) / np.sqrt(C),
                                             requires_grad=False)
        self.out_bias = torch.nn.Parameter(torch.zeros(1), requires_grad=False)
        self._scratch = None

    @property
    def receptive_field(self):
        return 1 + (self.kernel_size - 1) * sum(self.dilations)

    def packed_params(self):
        """Flat fp32 buffer in the order of include/ntm.h: per block W[Cin][K][C], b, alpha, R[Cin][C];
        then out_w[C], out_b[1]."""
        parts = []
        for W, b, a, R in zip(self.conv_weight, self.conv_bias, self.prelu, self.res_weight):
            parts += [W.permute(1, 2, 0).reshape(-1), b.reshape(-1), a.reshape(-1),
                      R[:, :, 0].permute(1, 0).reshape(-1)]
        parts += [self.out_weight.reshape(-1), self.out_bias.reshape(-1)]
        return torch.cat(parts).contiguous()

    @torch.no_grad()
    def forward(self, x):
        """x (B,1,T) fp32 on a HIP device -> y (B,1,T)."""
        if x.dim() != 3 or x.shape[1] != 1:
            raise RuntimeError(f"TCN.forward: expected (N_BATCHES, 1, N_SAMPLES), got {tuple(x.shape)}")
        if not x.is_cuda:
            raise RuntimeError("TCN.forward: this engine runs on a HIP device only (no CPU fallback)")
        B, T = x.shape[0], x.shape[2]
        xb = x.float().contiguous().view(B, T)
        y = torch.empty_like(xb)
        params = self.packed_params().to(x.device)
        L = _lib.lib()
        n = L.ntm_tcn_scratch_floats(B, T, self.channels)
        if self._scratch is None or self._scratch.numel() < n or self._scratch.device != x.device:
            self._scratch = torch.empty(n, device=x.device, dtype=torch.float32)
        dil = (ctypes.c_int * len(self.dilations))(*self.dilations)
        rc = L.ntm_tcn_forward(ptr(params), len(self.dilations), self.channels, self.kernel_size, dil, ptr(xb),
                               ptr(y), B, T, ptr(self._scratch), _lib.current_stream())
        _lib.check(rc, "ntm_tcn_forward")
        return y.view(B, 1, T)
