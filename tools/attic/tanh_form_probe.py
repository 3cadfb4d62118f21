#!/usr/bin/env python3
"""EXPERIMENT (round 4): what the (1 - t) rcp(1 + t) form of tanh in the matrix-pipe GRU kernel would buy and cost, measured on the
MI355X.  `make -C neural-tape-modeling_amd/csrc exp` builds libntm_tanh1.so (the product library with that one change); this tool runs
the same measurements through libntm.so and through it (child processes, NTM_LIB_PATH):
  * step time: the headline launch, 4096 x 65536, kernel_variant mfma2, best / median of 5;
  * accuracy against golden g19 (the reference's own predict) and its float64 truth for the two ill-conditioned DiffDelGRU L[ESR]
    AKAI checkpoints (w11, w9), the marginal GRU (w18) and three well-behaved ones, stream tiled to 1040 streams: predict and
    teacher-forced.
    python tools/tanh_form_probe.py            -> table on stdout
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BLOBS = ["w11", "w9", "w18", "w0", "w2", "w21"]


def child():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import torch
    import ntm_amd
    from helpers import load
    out = {"lib": os.path.basename(ntm_amd._lib.LIB_PATH)}
    m = ntm_amd.harness.build_model(ntm_amd.weights.W_GRU)
    m.kernel_variant = "mfma2"
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.rand(4096, 1, 65536, generator=g, device="cuda") - 0.5
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ts = []
    for i in range(7):
        m.initialize_hidden(); m.warm_start(); m.hidden = m.hidden.expand(1, 4096, 64).contiguous()
        ev[0].record(); m(x); ev[1].record(); torch.cuda.synchronize()
        if i >= 2:
            ts.append(ev[0].elapsed_time(ev[1]))
    out["ms_min"], out["ms_median"] = float(min(ts)), float(sorted(ts)[len(ts) // 2])
    del x
    gd = load("g19_checkpoints.npz")
    names, nf = [str(n) for n in gd["names"]], [str(f) for f in gd["name_file"]]
    xs = (gd["x_int16"].astype(np.float32) / 32768.0)[:4096]
    X = torch.from_numpy(np.broadcast_to(xs, (1040, 1, 4096)).copy()).cuda()
    rows = {}
    for k in BLOBS:
        name = names[nf.index(k + ".bin")]
        sd = ntm_amd.weights.load_state_dict(name)
        if name.startswith("GRU"):
            mm = ntm_amd.RNN(1, 64, 1)
            key = "_y"
        else:
            mm = ntm_amd.DiffDelRNN(1, 64, 1, max_delay=300)
            key = "_pre"
        mm.load_state_dict(sd)
        mm = mm.to("cuda").eval()
        mm.kernel_variant = "mfma2"
        ref, y64 = gd[k + key], gd[k + key + "64"].astype(np.float64)
        # the GRU + head only (pre_d for the DiffDelGRU): _gru on [B,T]
        def run(h0):
            mm.hidden = h0
            return mm._gru(X[:, 0].contiguous())[0].cpu().numpy()
        z = torch.zeros(1, 1024, device="cuda")
        mm.hidden = None
        mm._gru(z)
        hw = mm.hidden.expand(1, 1040, 64).contiguous()
        yp = run(hw)
        yf = run(torch.from_numpy(np.broadcast_to(gd[k + "_hwarm"], (1, 1040, 64)).copy()).cuda())
        rows[k] = {"name": name[:18] + ".." + name[-14:], "predict_vs_ref": float(np.abs(yp - ref).max()), "predict_vs_f64": float(np.abs(yp - y64).max()),
                   "forced_vs_ref": float(np.abs(yf - ref).max()), "ref_vs_f64": float(np.abs(ref - y64).max())}
    out["rows"] = rows
    print("JSON" + json.dumps(out))


def main():
    libs = [os.path.join(ROOT, "neural-tape-modeling_amd", n) for n in ("libntm.so", "libntm_tanh1.so", "libntm.so", "libntm_tanh1.so")]
    res = []
    for lib in libs:
        env = dict(os.environ, NTM_LIB_PATH=lib)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True, timeout=900)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("JSON")]
        if r.returncode != 0 or not line:
            print(f"{lib}: failed\n{r.stderr[-2000:]}")
            continue
        res.append(json.loads(line[0][4:]))
    for o in res:
        print(f"{o['lib']:18s} headline launch 4096 x 65536 (mfma2): min {o['ms_min']:.3f} ms  median {o['ms_median']:.3f} ms")
    print()
    print(f"{'checkpoint':36s} {'lib':16s} predict|ref32  predict|f64   forced|ref32   (reference's own fp32 vs f64)")
    for k in BLOBS:
        for o in res[:2]:
            r = o["rows"][k]
            print(f"{k:4s} {r['name']:31s} {o['lib']:16s} {r['predict_vs_ref']:.2e}      {r['predict_vs_f64']:.2e}     {r['forced_vs_ref']:.2e}      {r['ref_vs_f64']:.2e}")


if __name__ == "__main__":
    child() if "--child" in sys.argv else main()
