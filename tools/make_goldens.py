#!/usr/bin/env python3
"""Generate golden vectors + exported weights by importing the REFERENCE in this container.

Dev-only script: it is the one place that touches /root/reference.  It never travels to the
GPU box as a runtime dependency; what travels is its OUTPUT (data):

  neural-tape-modeling_amd/weights/<name>.bin + manifest.json   raw little-endian fp32 parameters
  tests/golden/*.npz                                            inputs + reference outputs

How the reference is imported: `code/model.py:13-15` pulls in torchaudio,
soundfile and librosa at module top although RNN / TimeVaryingDelayLine / DiffDelRNN use none
of them, so four empty stub modules are placed in sys.modules first.

Golden sets (ids follow SURVEY.md §8(c)):
  G1  RNN.predict (code/model.py:218-246), B=1, 16 segments x 8192 of real programme material
  G2  RNN.forward batched + state carry (code/model.py:67-88)
  G3  hidden state after warm_start (code/model.py:58-65, 382-391)
  G4  TimeVaryingDelayLine.forward edge battery (code/model.py:269-320)
  G5  DiffDelRNN.predict (code/model.py:618-653)
  G6  RNN.predict on 65 536 samples (drift check)
  G7  weights-dir name -> (model, hidden, loss) (code/utilities/utilities.py:872-914)
  G8  DiffDelRNN batched validate-style use: warmup=True then chunked forward (code/model.py:555-590)

Usage:  python tools/make_goldens.py
"""
import hashlib
import json
import os
import sys
import types
import wave

import numpy as np

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WDIR = os.path.join(ROOT, "neural-tape-modeling_amd", "weights")
GDIR = os.path.join(ROOT, "tests", "golden")

for m in ["torchaudio", "soundfile", "librosa", "librosa.filters"]:
    sys.modules[m] = types.ModuleType(m)
sys.modules["librosa.filters"].mel = lambda *a, **k: None
sys.modules["librosa"].filters = sys.modules["librosa.filters"]
sys.path.insert(0, os.path.join(REF, "code"))

import torch  # noqa: E402
import model as refmodel  # noqa: E402  (the reference's code/model.py)
from utilities.utilities import (nextpow2, parse_hidden_size, parse_loss,  # noqa: E402
                                 parse_model)

torch.set_num_threads(4)
torch.manual_seed(0)

W_G = "GRU-HS[64]-L[DCPreESR]-DS[ReelToReel_Dataset_MiniPulse100_CHOWTAPE]_BEST"
W_G_ESR = "GRU-HS[64]-L[ESR]-DS[ReelToReel_Dataset_MiniPulse100_CHOWTAPE]_BEST"
W_D = "DiffDelGRU-HS[64]-L[DCPreESR]-DS[ReelToReel_Dataset_MiniPulse100_CHOWTAPE_WOWFLUTTER]_BEST"
W_D_ESR = "DiffDelGRU-HS[64]-L[ESR]-DS[ReelToReel_Dataset_MiniPulse100_CHOWTAPE_WOWFLUTTER]_BEST"
EXPORT = [W_G, W_G_ESR, W_D, W_D_ESR]
KEYS = ["GRU.weight_ih_l0", "GRU.weight_hh_l0", "GRU.bias_ih_l0", "GRU.bias_hh_l0",
        "output.weight", "output.bias"]
FS = 44100


def sha256(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        h.update(f.read())
    return h.hexdigest()


def load_sd(name):
    return torch.load(os.path.join(REF, "weights", name, "best.pth"), map_location="cpu")


def export_weights():
    os.makedirs(WDIR, exist_ok=True)
    manifest = {}
    for i, name in enumerate(EXPORT):
        sd = load_sd(name)
        blobs, entries, off = [], [], 0
        for k in KEYS:
            if k not in sd:
                continue
            a = sd[k].detach().numpy().astype("<f4").ravel()
            entries.append({"key": k, "shape": list(sd[k].shape), "offset": off, "count": int(a.size)})
            off += a.size
            blobs.append(a)
        fname = f"w{i}.bin"
        np.concatenate(blobs).tofile(os.path.join(WDIR, fname))
        manifest[name] = {
            "file": fname,
            "source_sha256": sha256(os.path.join(REF, "weights", name, "best.pth")),
            "tensors": entries,
        }
    with open(os.path.join(WDIR, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1)


def read_wav(path):
    with wave.open(path, "rb") as w:
        assert w.getsampwidth() == 2
        nch = w.getnchannels()
        a = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2").astype(np.float32) / 32768.0
        return a.reshape(-1, nch)[:, 0].copy(), w.getframerate()


def make_rnn(name):
    m = refmodel.RNN(input_size=1, hidden_size=parse_hidden_size(name), output_size=1, skip=False)
    m.load_state_dict(load_sd(name))
    m.eval()
    return m


def make_ddr(name, max_delay):
    m = refmodel.DiffDelRNN(input_size=1, hidden_size=parse_hidden_size(name), output_size=1,
                            skip=False, max_delay=max_delay)
    m.load_state_dict(load_sd(name))
    m.eval()
    return m


def g1():
    """16 x 8192 segments of the shipped EXP1A input WAVs -> RNN.predict per segment (B=1)."""
    m = make_rnn(W_G)
    segs = []
    ids = [21, 52, 77, 78, 109]
    for i in ids:
        a, fs = read_wav(os.path.join(REF, "results/EXP1A_ToyData/predictions", f"{i}_input.wav"))
        assert fs == FS
        # 3 or 4 non-overlapping windows from the loud part of each 5-s clip
        nseg = 4 if i == 21 else 3
        start = FS // 2
        for s in range(nseg):
            segs.append(a[start + s * 16384:start + s * 16384 + 8192])
    x = np.stack(segs).astype(np.float32)
    assert x.shape == (16, 8192)
    ys = []
    with torch.inference_mode():
        for b in range(16):
            ys.append(m.predict(torch.from_numpy(x[b]).view(1, 1, -1)).numpy().reshape(-1))
    np.savez_compressed(os.path.join(GDIR, "g1_predict_16x8192.npz"), weights=W_G, x=x,
                        y=np.stack(ys).astype(np.float32))


def g2():
    m = make_rnn(W_G)
    rng = np.random.default_rng(1234)
    x = rng.uniform(-0.5, 0.5, size=(4, 1, 4096)).astype(np.float32)
    with torch.inference_mode():
        m.initialize_hidden()
        y0 = m.forward(torch.from_numpy(x[:, :, :1500]))
        y1 = m.forward(torch.from_numpy(x[:, :, 1500:]))
        hid = m.hidden.numpy().copy()
    # also an fp64 input (cast path, code/model.py:76)
    with torch.inference_mode():
        m.initialize_hidden()
        y64 = m.forward(torch.from_numpy(x[:2, :, :256].astype(np.float64)))
    np.savez_compressed(os.path.join(GDIR, "g2_forward_carry.npz"), weights=W_G, x=x,
                        y=np.concatenate([y0.numpy(), y1.numpy()], axis=2), hidden=hid,
                        y64=y64.numpy())


def g3():
    out = {}
    with torch.inference_mode():
        for tag, name in [("wg", W_G), ("wg_esr", W_G_ESR)]:
            m = make_rnn(name)
            m.initialize_hidden()
            m.warm_start()
            out[f"{tag}_hidden"] = m.hidden.numpy().copy()
        for tag, name in [("wd", W_D), ("wd_esr", W_D_ESR)]:
            m = make_ddr(name, 300)
            m.initialize_hidden(1, m.max_delay)
            m.warm_start()
            out[f"{tag}_hidden"] = m.hidden.numpy().copy()
            out[f"{tag}_buffer"] = m.diffdel.buffer.numpy().copy()
    np.savez_compressed(os.path.join(GDIR, "g3_warm_start.npz"), **out)


def g4():
    """Delay-line battery: D=37, N=3 streams, three 50-sample chunks + one 20-sample chunk (T<D)."""
    D = 37
    rng = np.random.default_rng(99)
    T = 170
    x = rng.standard_normal((3, 1, T)).astype(np.float32)
    d = rng.uniform(0, D, size=(3, 1, T)).astype(np.float32)
    d[0, 0, :10] = np.arange(10)                 # integer delays
    d[0, 0, 10] = D                              # d == D exactly (w_b tap dropped)
    d[0, 0, 11] = D - 1
    d[0, 0, 12] = 0.0
    d[1, 0, :8] = [0.25, 0.5, 36.5, 36.999, 1e-6, 17.0, 17.5, 3.75]
    d[2, 0, 50:60] = D                           # first samples of 2nd chunk read deep history
    d[2, 0, 100:105] = [-0.25, -0.5, -0.999, 0.0, 0.5]   # -1<d<0: future tap silently missing
    dl = refmodel.TimeVaryingDelayLine(max_delay=D)
    dl.init_buffer(3, D)
    ys, bufs = [], []
    bounds = [0, 50, 100, 150, 170]
    with torch.inference_mode():
        for a, b in zip(bounds[:-1], bounds[1:]):
            ys.append(dl.forward(torch.from_numpy(x[:, :, a:b]), torch.from_numpy(d[:, :, a:b])).numpy())
            bufs.append(dl.buffer.numpy().copy())
        # warmup=True path: returns x unchanged, only updates the buffer
        dl2 = refmodel.TimeVaryingDelayLine(max_delay=D)
        dl2.init_buffer(3, D)
        yw = dl2.forward(torch.from_numpy(x[:, :, :50]), torch.from_numpy(d[:, :, :50]), warmup=True).numpy()
        bw = dl2.buffer.numpy().copy()
        yw2 = dl2.forward(torch.from_numpy(x[:, :, 50:100]), torch.from_numpy(d[:, :, 50:100])).numpy()
    np.savez_compressed(os.path.join(GDIR, "g4_delay_line.npz"), D=D, x=x, d=d,
                        bounds=np.array(bounds), y=np.concatenate(ys, axis=2),
                        buf_after_each=np.stack(bufs), y_warm=yw, buf_warm=bw, y_after_warm=yw2)


def wow_traj(T, fs=FS, a=0.004, w=1.3, psi=0.0, base=0.0271):
    n = np.arange(T)
    return (fs * (base + a * np.sin(2 * np.pi * w * n / fs + psi)
                  + 0.0005 * np.sin(2 * np.pi * 23 * n / fs))).astype(np.float32)


def g5():
    max_delay = int(1.25 * 0.0335 * FS)   # code/test-model.py:223 with delay_analyzer.max_delay=33.5 ms
    assert max_delay == 1846
    m = make_ddr(W_D, max_delay)
    rng = np.random.default_rng(555)
    T = 6000
    x = rng.uniform(-0.5, 0.5, size=(1, 1, T)).astype(np.float32)
    d = wow_traj(T).reshape(1, 1, T)
    with torch.inference_mode():
        y, pre = m.predict(torch.from_numpy(x), torch.from_numpy(d))
        buf = m.diffdel.buffer.numpy().copy()
        hid = m.hidden.numpy().copy()
    np.savez_compressed(os.path.join(GDIR, "g5_diffdel_predict.npz"), weights=W_D,
                        max_delay=max_delay, x=x, d=d, y=y.numpy(), pre_d=pre.numpy(),
                        buffer=buf, hidden=hid, D_effective=m.diffdel.max_delay)


def g6():
    m = make_rnn(W_G)
    rng = np.random.default_rng(66)
    x = rng.uniform(-0.5, 0.5, size=(1, 1, 65536)).astype(np.float32)
    with torch.inference_mode():
        y = m.predict(torch.from_numpy(x)).numpy()
    np.savez_compressed(os.path.join(GDIR, "g6_long_65536.npz"), weights=W_G, x=x, y=y)


def g7():
    rows = []
    for name in sorted(os.listdir(os.path.join(REF, "weights"))):
        if "-HS[" not in name:
            continue
        rows.append({"name": name, "model": parse_model(name), "hidden": parse_hidden_size(name),
                     "loss": parse_loss(name)})
    tab = {"names": rows, "nextpow2": {str(n): int(nextpow2(n)) for n in
                                       [1, 2, 3, 255, 256, 257, 1000, 1024, 1477, 1846, 65535, 65537]}}
    with open(os.path.join(GDIR, "g7_name_parsers.json"), "w") as f:
        json.dump(tab, f, indent=1)


def g8():
    """Batched DiffDelRNN as the reference's validate() drives it (code/model.py:555-590)."""
    max_delay = 300
    m = make_ddr(W_D, max_delay)
    rng = np.random.default_rng(88)
    B, T, INIT = 3, 3000, 512
    x = rng.uniform(-0.5, 0.5, size=(B, 1, T)).astype(np.float32)
    d = np.stack([wow_traj(T, a=0.0005 * (b + 1), w=1.0 + b, psi=b, base=0.004) for b in range(B)])
    d = d.reshape(B, 1, T).astype(np.float32)
    assert d.max() <= max_delay and d.min() >= 0
    ys, pres = [], []
    with torch.inference_mode():
        m.initialize_hidden(B, m.max_delay)
        y0, p0 = m.forward(torch.from_numpy(x[:, :, :INIT]), torch.from_numpy(d[:, :, :INIT]), warmup=True)
        ys.append(y0.numpy()); pres.append(p0.numpy())
        off = INIT
        while off < T:
            y1, p1 = m.forward(torch.from_numpy(x[:, :, off:off + 1000]), torch.from_numpy(d[:, :, off:off + 1000]))
            ys.append(y1.numpy()); pres.append(p1.numpy())
            off += 1000
        buf = m.diffdel.buffer.numpy().copy()
        hid = m.hidden.numpy().copy()
    np.savez_compressed(os.path.join(GDIR, "g8_diffdel_batched.npz"), weights=W_D, max_delay=max_delay,
                        init_len=INIT, chunk=1000, x=x, d=d, y=np.concatenate(ys, axis=2),
                        pre_d=np.concatenate(pres, axis=2), buffer=buf, hidden=hid)


if __name__ == "__main__":
    os.makedirs(GDIR, exist_ok=True)
    export_weights()
    for fn in [g1, g2, g3, g4, g5, g6, g7, g8]:
        fn()
        print("done", fn.__name__)
    tot = sum(os.path.getsize(os.path.join(GDIR, f)) for f in os.listdir(GDIR))
    print("golden bytes:", tot)
