#!/usr/bin/env python3
"""Goldens at the reference's own operating point, from ALL shipped checkpoints (dev-only: imports the reference
through tools/make_goldens.py's stubs; only its OUTPUT -- data -- travels).

  neural-tape-modeling_amd/weights/w<i>.bin + manifest.json   all 44 `weights/<name>/best.pth` of the reference (32 distinct
                                                              files; names that share bytes share a blob); w0..w3 keep
                                                              their round-1 numbering
  tests/golden/g19_checkpoints.npz     every distinct checkpoint through `predict` (code/model.py:218-246, :618-653,
                                       loaded as code/test-model.py:217-233 does):
                                         GRU        x (1,1,4096)                         -> y
                                         DiffDelGRU x, toy wow trajectory, max_delay 1846 -> y, pre_d
                                         DiffDelGRU trained on the real AKAI tape, additionally at the real-tape delay
                                                    length max_delay = 11000 (code/test-model.py:222-230 with a measured
                                                    delay of 0.2 s), trajectory around 8400 samples, T = 12288
  tests/golden/g20_operating_point.npz the harness' shapes (scripts/test-model-loss.sh:22 SEGMENT_LENGTH = 441000):
                                         GRU  CHOWTAPE + GRU AKAI   predict, T = 441000 (int16 programme-like input)
                                         DiffDelGRU CHOWTAPE_WOWFLUTTER  predict, T = 65536, D = 1847
                                         DiffDelGRU AKAI                 predict, T = 20000, D = 11001

Usage:  python tools/make_goldens_checkpoints.py            (about two minutes of torch-CPU)
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_goldens as mg  # noqa: E402  (stubs + imports the reference's code/model.py)

torch = mg.torch
FS = mg.FS
REF_W = os.path.join(mg.REF, "weights")


def all_names():
    return sorted(n for n in os.listdir(REF_W) if "-HS[" in n)


def export_all():
    """All 44 names -> manifest; blobs are shared between names whose best.pth have identical bytes."""
    with open(os.path.join(mg.WDIR, "manifest.json")) as f:
        manifest = json.load(f)
    by_sha = {v["source_sha256"]: v["file"] for v in manifest.values()}
    nxt = 1 + max(int(v["file"][1:-4]) for v in manifest.values())
    for name in all_names():
        src = os.path.join(REF_W, name, "best.pth")
        sha = mg.sha256(src)
        sd = mg.load_sd(name)
        blobs, entries, off = [], [], 0
        for k in mg.KEYS:
            if k not in sd:
                continue
            a = sd[k].detach().numpy().astype("<f4").ravel()
            entries.append({"key": k, "shape": list(sd[k].shape), "offset": off, "count": int(a.size)})
            off += a.size
            blobs.append(a)
        if sha not in by_sha:
            by_sha[sha] = f"w{nxt}.bin"
            nxt += 1
            np.concatenate(blobs).tofile(os.path.join(mg.WDIR, by_sha[sha]))
        else:
            assert np.array_equal(np.fromfile(os.path.join(mg.WDIR, by_sha[sha]), dtype="<f4"), np.concatenate(blobs))
        manifest[name] = {"file": by_sha[sha], "source_sha256": sha, "tensors": entries}
    with open(os.path.join(mg.WDIR, "manifest.json"), "w") as f:
        json.dump(dict(sorted(manifest.items())), f, indent=1)
    return manifest


def programme(T, seed, peak=0.7):
    """Programme-like test signal on the int16 grid of a WAV file (what the harness feeds: code/dataset.py reads PCM):
    three gliding partials under a slow envelope plus a noise floor.  Returned as int16; x = int16 / 32768 exactly."""
    rng = np.random.default_rng(seed)
    n = np.arange(T)
    env = 0.55 + 0.45 * np.sin(2 * np.pi * 0.7 * n / FS + rng.uniform(0, 6.28))
    sig = np.zeros(T)
    for f0, a in ((110.0, 0.5), (587.0, 0.3), (3100.0, 0.15)):
        f = f0 * (1.0 + 0.2 * np.sin(2 * np.pi * 0.31 * n / FS + rng.uniform(0, 6.28)))
        sig += a * np.sin(2 * np.pi * np.cumsum(f) / FS)
    sig = peak * env * sig / 0.95 + 0.02 * rng.standard_normal(T)
    return np.clip(np.round(sig * 32768.0), -32767, 32767).astype(np.int16)


def real_tape_traj(T, base=8400.0, wow=380.0, psi=0.4):
    """Delay trajectory in samples at the real-tape scale (record -> playback head distance of 0.19 s at 7.5 ips,
    wow of a few ms): stays inside [8000, 8800], far below max_delay = 11000 and far above the toy 1846."""
    n = np.arange(T)
    return (base + wow * np.sin(2 * np.pi * 1.1 * n / FS * 8 + psi)
            + 20.0 * np.sin(2 * np.pi * 23 * n / FS)).astype(np.float32)


def f64_truth(name, x, d=None, max_delay=None):
    """The same network evaluated in float64: torch.nn.GRU / nn.Linear (the modules code/model.py:44-45, :364-365 build)
    cast to double, 1024 zero samples then x from h = 0 (what predict() computes, code/model.py:58-65, :218-246); for the
    DiffDelGRU the delay line's closed form (bit-exact against code/model.py:287-315 in fp32) in
    float64 on the float64 pre_d.  |y32 - y64| is the reference's OWN fp32 rounding noise on this input: the yardstick
    for checkpoints whose dynamics amplify rounding differences beyond 1e-5."""
    sd = mg.load_sd(name)
    gru = torch.nn.GRU(1, 64, batch_first=True).double()
    lin = torch.nn.Linear(64, 1, bias="output.bias" in sd).double()
    gru.load_state_dict({k[4:]: v.double() for k, v in sd.items() if k.startswith("GRU.")})
    lin.load_state_dict({k[7:]: v.double() for k, v in sd.items() if k.startswith("output.")})
    T = x.shape[-1]
    xx = torch.cat([torch.zeros(1, 1024, 1, dtype=torch.float64), torch.from_numpy(x.reshape(1, T, 1)).double()], 1)
    o, _ = gru(xx)
    pre_all = lin(o)[0, :, 0].numpy()
    pre = pre_all[1024:]
    if d is None:
        return pre
    D = int(max_delay) + 1
    assert D >= 1024
    xp = np.concatenate([np.zeros(D - 1024), pre_all])          # history: the warm-up filled the buffer with its pre_d (d = 0)
    dd = d.reshape(-1).astype(np.float64)
    n = np.arange(T)
    k = np.floor(dd)
    wa = np.maximum(1.0 - np.abs(k - dd), 0.0)
    wb = np.where(k + 1 <= D, np.maximum(1.0 - np.abs(k + 1 - dd), 0.0), 0.0)
    i = D + n - k.astype(np.int64)
    return wa * xp[i] + wb * xp[np.maximum(i - 1, 0)], pre


def gate_noise(name, x, seeds=32):
    """How far 1-ulp rounding noise in the GATES moves this checkpoint's output: the network's own equations
    (code/model.py:81-82 = torch.nn.GRU + Linear, restated in float64) with every gate value perturbed by what a float32
    sigmoid / tanh accurate to about one unit in the last place carries -- r, z relative U(-1,1) 2^-23, n absolute U(-1,1) 2^-23
    (rms 7e-8: the measured rms of exp2 -> 1 + e -> rcp on fp32 hardware units) -- 1024 zero samples then x, `seeds` independent draws.
    -> max |y_perturbed - y| per draw.  A property of the checkpoint (and input), not of any implementation: where it exceeds
    1e-5 no float32 implementation with 1-ulp transcendentals can be expected within 1e-5 of another one."""
    sd = {k: v.double().numpy() for k, v in mg.load_sd(name).items()}
    Wih, Whh = sd["GRU.weight_ih_l0"][:, 0], sd["GRU.weight_hh_l0"]
    bih, bhh, wo = sd["GRU.bias_ih_l0"], sd["GRU.bias_hh_l0"], sd["output.weight"][0]
    H = Whh.shape[1]
    rng = np.random.default_rng(2323)
    xx = np.concatenate([np.zeros(1024), np.asarray(x, np.float64).reshape(-1)])
    U = 2.0 ** -23
    h = np.zeros((seeds + 1, H))                       # row 0: unperturbed
    ys = np.empty((seeds + 1, xx.size))
    mask = np.ones((seeds + 1, 1)); mask[0] = 0.0
    WT = Whh.T.copy()
    for t, xv in enumerate(xx):
        gh = h @ WT + bhh
        gi = xv * Wih + bih
        e = rng.uniform(-U, U, (3, seeds + 1, H)) * mask
        r = (1.0 / (1.0 + np.exp(-(gi[:H] + gh[:, :H])))) * (1.0 + e[0])
        z = (1.0 / (1.0 + np.exp(-(gi[H:2 * H] + gh[:, H:2 * H])))) * (1.0 + e[1])
        n = np.tanh(gi[2 * H:] + r * gh[:, 2 * H:]) + e[2]
        h = (h - n) * z + n
        ys[:, t] = h @ wo
    return np.abs(ys[1:, 1024:] - ys[0, 1024:]).max(axis=1).astype(np.float32)


def ref_batch_noise(name, x):
    """max |row of a B = 3 forward - the B = 1 forward| of the REFERENCE model itself (same torch, same thread count):
    how far the reference is from reproducing itself."""
    m = mg.make_rnn(name)
    m.initialize_hidden(); m.warm_start()
    h = m.hidden.clone()
    y1 = m.forward(torch.from_numpy(x)).numpy()
    m.hidden = h.repeat(1, 3, 1)
    y3 = m.forward(torch.from_numpy(x).repeat(3, 1, 1)).numpy()
    return float(np.abs(y3 - y1).max())


def g19(manifest):
    names = all_names()
    files = sorted({manifest[n]["file"] for n in names}, key=lambda s: int(s[1:-4]))
    rep = {f: next(n for n in names if manifest[n]["file"] == f) for f in files}      # one name per distinct blob
    T, TL = 4096, 12288
    x16 = programme(TL, 1919)
    x = (x16.astype(np.float32) / 32768.0).reshape(1, 1, TL)
    d_toy = mg.wow_traj(T).reshape(1, 1, T)
    d_real = real_tape_traj(TL).reshape(1, 1, TL)
    assert d_real.min() >= 8000 and d_real.max() <= 8900
    out = {"x_int16": x16, "d_toy": d_toy[0, 0], "d_real": d_real[0, 0], "T": T, "T_long": TL,
           "max_delay_toy": 1846, "max_delay_real": 11000,
           "names": np.array(names), "name_file": np.array([manifest[n]["file"] for n in names]),
           "files": np.array(files)}
    with torch.inference_mode():
        for f in files:
            name = rep[f]
            t0 = time.time()
            if mg.parse_model(name) == "GRU":
                m = mg.make_rnn(name)
                out[f"{f[:-4]}_y"] = m.predict(torch.from_numpy(x[:, :, :T])).numpy()[0, 0]
                m.initialize_hidden(); m.warm_start()
                out[f"{f[:-4]}_hwarm"] = m.hidden.numpy()[0, 0].copy()
                out[f"{f[:-4]}_y64"] = f64_truth(name, x[:, :, :T]).astype(np.float32)
                out[f"{f[:-4]}_b3"] = ref_batch_noise(name, x[:, :, :T])
                out[f"{f[:-4]}_gate_noise"] = gate_noise(name, x[:, :, :T])
            else:
                m = mg.make_ddr(name, 1846)
                y, pre = m.predict(torch.from_numpy(x[:, :, :T]), torch.from_numpy(d_toy))
                out[f"{f[:-4]}_y"], out[f"{f[:-4]}_pre"] = y.numpy()[0, 0], pre.numpy()[0, 0]
                m.initialize_hidden(1, m.max_delay); m.warm_start()
                assert not m.diffdel.buffer.numpy()[0, 0, :-1024].any()
                out[f"{f[:-4]}_hwarm"] = m.hidden.numpy()[0, 0].copy()
                out[f"{f[:-4]}_bwarm"] = m.diffdel.buffer.numpy()[0, 0, -1024:].copy()     # the rest of the buffer is zero
                y64, p64 = f64_truth(name, x[:, :, :T], d_toy, 1846)
                out[f"{f[:-4]}_y64"], out[f"{f[:-4]}_pre64"] = y64.astype(np.float32), p64.astype(np.float32)
                out[f"{f[:-4]}_gate_noise"] = gate_noise(name, x[:, :, :T])
                if "AKAI" in name:
                    m = mg.make_ddr(name, 11000)
                    y, pre = m.predict(torch.from_numpy(x), torch.from_numpy(d_real))
                    out[f"{f[:-4]}_y_real"], out[f"{f[:-4]}_pre_real"] = y.numpy()[0, 0], pre.numpy()[0, 0]
                    assert m.diffdel.max_delay == 11001
                    y64, p64 = f64_truth(name, x, d_real, 11000)
                    out[f"{f[:-4]}_y64_real"], out[f"{f[:-4]}_pre64_real"] = y64.astype(np.float32), p64.astype(np.float32)
                    out[f"{f[:-4]}_gate_noise_real"] = gate_noise(name, x)
            print(f"  g19 {f} {name[:60]}  {time.time() - t0:.1f}s", flush=True)
    np.savez_compressed(os.path.join(mg.GDIR, "g19_checkpoints.npz"), **out)


W_G_AKAI = "GRU-HS[64]-L[DCPreESR]-DS[ReelToReel_Dataset_MiniPulse100_AKAI_IPS[7.5]_MAXELL]_BEST"
W_D_AKAI = "DiffDelGRU-HS[64]-L[DCPreESR]-DS[ReelToReel_Dataset_MiniPulse100_AKAI_IPS[7.5]_MAXELL]_BEST"


def g20():
    out = {}
    T = 441000                                       # scripts/test-model-loss.sh:22
    x16 = programme(T, 2020)
    x = torch.from_numpy((x16.astype(np.float32) / 32768.0).reshape(1, 1, T))
    out["gru_x_int16"] = x16
    with torch.inference_mode():
        for tag, name in (("chow", mg.W_G), ("akai", W_G_AKAI)):
            t0 = time.time()
            out[f"gru_{tag}_weights"] = name
            out[f"gru_{tag}_y"] = mg.make_rnn(name).predict(x).numpy()[0, 0]
            out[f"gru_{tag}_y64"] = f64_truth(name, x.numpy()).astype(np.float32)
            out[f"gru_{tag}_b3"] = ref_batch_noise(name, x.numpy())
            out[f"gru_{tag}_gate_noise"] = gate_noise(name, x.numpy(), seeds=16)
            print(f"  g20 GRU {tag} T={T}  {time.time() - t0:.1f}s", flush=True)
        # DiffDelGRU, toy delay length at the bench's sequence length
        T2 = 65536
        x2_16 = programme(T2, 2021)
        d2 = mg.wow_traj(T2)
        m = mg.make_ddr(mg.W_D, 1846)
        t0 = time.time()
        y, pre = m.predict(torch.from_numpy((x2_16.astype(np.float32) / 32768.0).reshape(1, 1, T2)),
                           torch.from_numpy(d2.reshape(1, 1, T2)))
        y64, p64 = f64_truth(mg.W_D, (x2_16.astype(np.float32) / 32768.0).reshape(1, 1, T2), d2, 1846)
        out.update(dd_toy_y64=y64.astype(np.float32), dd_toy_pre64=p64.astype(np.float32),
                   dd_toy_gate_noise=gate_noise(mg.W_D, x2_16.astype(np.float32) / 32768.0, seeds=16))
        out.update(dd_toy_weights=mg.W_D, dd_toy_max_delay=1846, dd_toy_x_int16=x2_16, dd_toy_d=d2,
                   dd_toy_y=y.numpy()[0, 0], dd_toy_pre=pre.numpy()[0, 0],
                   dd_toy_buffer=m.diffdel.buffer.numpy()[0, 0].copy(), dd_toy_hidden=m.hidden.numpy()[0, 0].copy())
        print(f"  g20 DiffDel toy T={T2} D=1847  {time.time() - t0:.1f}s", flush=True)
        # DiffDelGRU, real-tape delay length
        T3 = 20000
        x3_16 = programme(T3, 2022)
        d3 = real_tape_traj(T3, base=8300.0, wow=450.0, psi=1.0)
        m = mg.make_ddr(W_D_AKAI, 11000)
        t0 = time.time()
        y, pre = m.predict(torch.from_numpy((x3_16.astype(np.float32) / 32768.0).reshape(1, 1, T3)),
                           torch.from_numpy(d3.reshape(1, 1, T3)))
        y64, p64 = f64_truth(W_D_AKAI, (x3_16.astype(np.float32) / 32768.0).reshape(1, 1, T3), d3, 11000)
        out.update(dd_real_y64=y64.astype(np.float32), dd_real_pre64=p64.astype(np.float32),
                   dd_real_gate_noise=gate_noise(W_D_AKAI, x3_16.astype(np.float32) / 32768.0, seeds=16))
        out.update(dd_real_weights=W_D_AKAI, dd_real_max_delay=11000, dd_real_x_int16=x3_16, dd_real_d=d3,
                   dd_real_y=y.numpy()[0, 0], dd_real_pre=pre.numpy()[0, 0],
                   dd_real_buffer=m.diffdel.buffer.numpy()[0, 0].copy(), dd_real_hidden=m.hidden.numpy()[0, 0].copy())
        assert m.diffdel.max_delay == 11001
        print(f"  g20 DiffDel real T={T3} D=11001  {time.time() - t0:.1f}s", flush=True)
    np.savez_compressed(os.path.join(mg.GDIR, "g20_operating_point.npz"), **out)


if __name__ == "__main__":
    man = export_all()
    print("exported", len(man), "names,", len({v['file'] for v in man.values()}), "distinct blobs")
    g19(man)
    g20()
    for f in ("g19_checkpoints.npz", "g20_operating_point.npz"):
        print(f, os.path.getsize(os.path.join(mg.GDIR, f)), "bytes")
