#!/usr/bin/env python3
"""Which rounding source moves a checkpoint's output?  ("if any AKAI checkpoint exceeds 5e-6, say which
op it comes from".)  CPU-only, no reference import: the network's equations (torch.nn.GRU + Linear, code/model.py:81-82) in
float64 on golden g19's input, with ONE source of float32 rounding injected at a time; prints max |y - y_f64| per source.

    python tools/rounding_attribution.py w11        # blob id of tests/golden/g19_checkpoints.npz (name_file)

sources:  prescale  -log2e / 2 log2e folded into fp32 weights (what every kernel here does so the gates start at v_exp_f32)
          dot32     the 64-term recurrent dot product rounded to fp32 once
          dotnoise  ... plus ~4 ulp of accumulation-order noise
          h32       the state rounded to fp32 every step
          exparg    the exp2 argument rounded to fp32
          r / z     sigmoid outputs with 1-ulp relative noise (v_exp_f32 -> 1 + e -> v_rcp_f32)
          nabs      tanh = 1 - 2 rcp(1 + e): ABSOLUTE noise of 1 ulp of 1.0 (the form cancels for small |n|)
          nrel      a tanh accurate to 1 ulp RELATIVE (what torch's Sleef tanhf delivers)
Finding (profiles/r04_rounding_attribution.txt): for the DiffDelGRU L[ESR] AKAI checkpoints everything but the gates stays
near 1e-6; 1-ulp noise on z gives 3e-6 .. 1.2e-5 and the absolute error of the tanh form 4e-6 .. 2.5e-5, while a relatively
accurate tanh would give ~1e-6.  A cheaper-to-fix form, (e - 1) rcp(e + 1), was emulated in fp32 (second part, `emulate`):
it improves the worst case by 1.7x only (z's noise remains), not enough to change which checkpoints exceed 1e-5.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import load, state_dict_np  # noqa: E402

ULP = 2.0 ** -24
L2 = 1.4426950408889634


def main(blob, seeds=8):
    g = load("g19_checkpoints.npz")
    names, nf = [str(n) for n in g["names"]], [str(f) for f in g["name_file"]]
    name = names[nf.index(blob + ".bin")]
    sd = state_dict_np(name)
    x = (g["x_int16"].astype(np.float64) / 32768.0)[:4096]
    xx = np.concatenate([np.zeros(1024), x])
    Wih, Whh = sd["GRU.weight_ih_l0"][:, 0].astype(np.float64), sd["GRU.weight_hh_l0"].astype(np.float64)
    bih, bhh = sd["GRU.bias_ih_l0"].astype(np.float64), sd["GRU.bias_hh_l0"].astype(np.float64)
    wo = sd["output.weight"][0].astype(np.float64)
    bo = float(sd["output.bias"][0]) if "output.bias" in sd else 0.0
    H = 64
    rng = np.random.default_rng(0)

    def run(variant, S):
        h = np.zeros((S, H))
        ys = np.empty((S, len(xx)))
        W, wi, bi, bh = Whh, Wih, bih, bhh
        if variant == "prescale":
            sc = np.concatenate([np.full(2 * H, -L2), np.full(H, 2 * L2)])
            W = np.float32(Whh * sc[:, None]).astype(np.float64) / sc[:, None]
            wi, bi, bh = (np.float32(a * sc).astype(np.float64) / sc for a in (Wih, bih, bhh))
        for t, xv in enumerate(xx):
            gh = h @ W.T + bh
            gi = xv * wi + bi
            if variant == "dot32":
                gh = np.float32(gh).astype(np.float64)
            if variant == "dotnoise":
                gh = gh * (1 + 4 * ULP * rng.uniform(-1, 1, gh.shape))
            ar, az = gi[:H] + gh[:, :H], gi[H:2 * H] + gh[:, H:2 * H]
            r, z = 1 / (1 + np.exp(-ar)), 1 / (1 + np.exp(-az))
            if variant == "r":
                r = r * (1 + 2 * ULP * rng.uniform(-1, 1, r.shape))
            if variant == "z":
                z = z * (1 + 2 * ULP * rng.uniform(-1, 1, z.shape))
            pn = gi[2 * H:] + r * gh[:, 2 * H:]
            if variant == "exparg":
                r = 1 / (1 + np.exp(-np.float32(ar * L2).astype(np.float64) / L2))
                z = 1 / (1 + np.exp(-np.float32(az * L2).astype(np.float64) / L2))
                pn = np.float32((gi[2 * H:] + r * gh[:, 2 * H:]) * 2 * L2).astype(np.float64) / (2 * L2)
            n = np.tanh(pn)
            if variant == "nabs":
                n = n + 2 * ULP * rng.uniform(-1, 1, n.shape)
            if variant == "nrel":
                n = n * (1 + 2 * ULP * rng.uniform(-1, 1, n.shape))
            h = (h - n) * z + n
            if variant == "h32":
                h = np.float32(h).astype(np.float64)
            ys[:, t] = h @ wo + bo
        return ys[:, 1024:]

    base = run("none", 1)[0]
    key = "_y" if name.startswith("GRU") else "_pre"
    print(name)
    print(f"reference fp32 vs float64: {np.abs(g[blob + key].astype(np.float64) - base).max():.2e}")
    for v in ("prescale", "dot32", "dotnoise", "h32", "exparg", "r", "z", "nabs", "nrel"):
        det = v in ("prescale", "dot32", "h32", "exparg")
        d = np.abs(run(v, 1 if det else seeds) - base).max(axis=1)
        print(f"{v:9s}", " ".join(f"{e:.2e}" for e in d))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "w11")
