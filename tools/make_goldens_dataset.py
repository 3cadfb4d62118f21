#!/usr/bin/env python3
"""Golden g15: the reference's `VADataset` (code/dataset.py:36-125 construction, :174-293 create_patches, :348-429
__getitem__) run HERE over a small synthetic dataset tree, so that ntm_amd.feeder.SegmentFeeder can be pinned to it.
Dev-only; the reference never travels -- what is committed is DATA: the WAV/side-car-free tree is rebuilt by the test
from the arrays stored in the .npz (the test writes the same float32 WAV files with scipy), plus what the reference
returned for it.

torchaudio / soundfile are absent in this container; `VADataset` needs three calls from them:
  torchaudio.info(path)                  -> sample_rate, num_frames        (code/dataset.py:154-157,194-201)
  torchaudio.load(path, normalize=False) -> (tensor [C,N] native dtype, fs) (code/dataset.py:425-432)
  soundfile.read(path, always_2d=True)   -> (float64 [N,C], fs)             (code/utilities/utilities.py:623-625)
They are stubbed on scipy.io.wavfile; the files are float32 WAVs, for which all three are exact.

The tree (pins what a synthetic test of one's own would not): a '[' ']' pair in the dataset path (glob escaping,
code/dataset.py:133), ids 3 and 10 (sorted as STRINGS: input_10_ < input_3_), stereo audio + pilot pulse channel,
a last file that is not a whole number of segments, `sync` > 0, subset "test" -> directory "Test".
"""
import os
import shutil
import sys
import tempfile
import types

import numpy as np
from scipy.io import wavfile

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

ta = types.ModuleType("torchaudio")
sfm = types.ModuleType("soundfile")
for m in ["librosa", "librosa.filters"]:
    sys.modules[m] = types.ModuleType(m)
sys.modules["librosa.filters"].mel = lambda *a, **k: None
sys.modules["librosa"].filters = sys.modules["librosa.filters"]
sys.modules["torchaudio"], sys.modules["soundfile"] = ta, sfm

import torch  # noqa: E402


def _read(path):
    fs, a = wavfile.read(path)
    if a.ndim == 1:
        a = a[:, None]
    return fs, a


def _info(path):
    fs, a = _read(path)
    return types.SimpleNamespace(sample_rate=fs, num_frames=a.shape[0], num_channels=a.shape[1])


def _load(path, normalize=True, num_frames=-1, frame_offset=0):
    fs, a = _read(path)
    assert a.dtype == np.float32, "stub covers float32 WAVs (native dtype == normalised dtype)"
    return torch.from_numpy(np.ascontiguousarray(a.T)), fs


ta.info, ta.load, ta.set_audio_backend = _info, _load, lambda *a, **k: None
sfm.read = lambda path, always_2d=True: (lambda fs, a: (a.astype(np.float64), fs))(*_read(path))
sys.path.insert(0, os.path.join(REF, "code"))
import dataset as refdataset  # noqa: E402  (the reference's code/dataset.py)

from make_goldens_demod import pulse_train  # noqa: E402  (the pilot-track generator of golden g12)

FS = 44100


def make_file(rng, N, wow, drop_out=()):
    n = np.arange(N)
    audio_in = (0.3 * np.sin(2 * np.pi * 220 * n / FS) + 0.05 * rng.standard_normal(N)).astype(np.float32)
    audio_out = (0.25 * np.roll(audio_in, 1200) + 0.01 * rng.standard_normal(N)).astype(np.float32)
    pin = pulse_train(rng, N, FS, 441, lambda k: 0.0).astype(np.float32)
    pout = pulse_train(rng, N, FS, 441, lambda k: 1200.0 + wow * np.sin(2 * np.pi * 1.1 * k / FS), drop=drop_out,
                       amp=0.5, width=4, noise=2e-4).astype(np.float32)
    return np.stack([audio_in, pin], 1), np.stack([audio_out, pout], 1)


def main():
    rng = np.random.default_rng(15)
    tmp = tempfile.mkdtemp()
    root = os.path.join(tmp, "Toy[Set]_A")
    test = os.path.join(root, "Test")
    os.makedirs(test)
    os.makedirs(os.path.join(root, "Train"))
    files = {"3_first": make_file(rng, 40000, 25.0), "10_second": make_file(rng, 27000, 10.0, drop_out=(20,))}
    out = {"fs": FS, "names": np.array(sorted(files))}
    for name, (xi, xt) in files.items():
        wavfile.write(os.path.join(test, f"input_{name}.wav"), FS, xi)
        wavfile.write(os.path.join(test, f"target_{name}.wav"), FS, xt)
        out[f"file_{name}_input"], out[f"file_{name}_target"] = xi, xt
    # a file of another subset with a colliding id: must not be picked up by subset "test"
    wavfile.write(os.path.join(root, "Train", "input_3_other.wav"), FS, files["3_first"][0][:9000])
    wavfile.write(os.path.join(root, "Train", "target_3_other.wav"), FS, files["3_first"][1][:9000])
    out["file_train_3_other_len"] = 9000

    cases = {"a": dict(length=8000, sync=0.0, demodulate=False), "b": dict(length=6000, sync=0.25, demodulate=False),
             "c": dict(length=9000, sync=0.0, demodulate=True)}
    for tag, kw in cases.items():
        ds = refdataset.VADataset(root, subset="test", shuffle=False, preload=True, return_full=True, **kw)
        out[f"{tag}_length"], out[f"{tag}_sync"], out[f"{tag}_demodulate"] = kw["length"], kw["sync"], kw["demodulate"]
        out[f"{tag}_n"] = len(ds)
        out[f"{tag}_input_files"] = np.array([os.path.basename(p) for p in ds.input_files])
        out[f"{tag}_examples"] = np.array([[e["idx"], e["offset"]] for e in ds.examples], np.int64)
        out[f"{tag}_delay_stats"] = np.array([ds.delay_analyzer.min_delay, ds.delay_analyzer.mean_delay, ds.delay_analyzer.max_delay])
        out[f"{tag}_minutes"] = ds.minutes
        full = {0, len(ds) // 2, len(ds) - 1}            # three items with their audio; names / pulses / sums for all
        for i in range(len(ds)):
            inp, tgt, meta = ds[i]
            e = ds.examples[i]
            fname = os.path.basename(e["input_file"])[len("input_"):-len(".wav")]
            o = e["offset"]
            if not kw["demodulate"]:
                # verified HERE against the reference's return values, so the .npz need not repeat the audio: an item
                # is the [offset, offset + length) slice of both channels of its file pair
                assert np.array_equal(inp.numpy(), files[fname][0].T[:, o:o + kw["length"]])
                assert np.array_equal(tgt.numpy(), files[fname][1].T[:, o:o + kw["length"]])
            elif i in full:
                out[f"{tag}_{i}_input"], out[f"{tag}_{i}_target"] = inp.numpy(), tgt.numpy()
            if i in full:
                out[f"{tag}_{i}_traj"] = np.asarray(meta["delay_trajectory"])
            out[f"{tag}_{i}_shape_sums"] = np.array([inp.shape[-1], tgt.shape[-1], len(meta["delay_trajectory"]),
                                                     float(inp.double().sum()), float(tgt.double().sum()),
                                                     float(np.sum(meta["delay_trajectory"]))])
            out[f"{tag}_{i}_names"] = np.array([meta["input_name"], meta["target_name"]])
            for k in ("input_peaks", "output_peaks"):
                v = meta[k]
                out[f"{tag}_{i}_{k}"] = np.asarray(v if v is not None else [-1], np.int64)
        print(tag, len(ds), out[f"{tag}_examples"].tolist(), out[f"{tag}_delay_stats"])
    # the side-cars the reference wrote on first use (their pulse indices / trajectories pin the feeder's own analysis)
    for name in files:
        d = np.load(os.path.join(test, f"trajectory_{name}.npy"), allow_pickle=True).item()
        out[f"sidecar_{name}_input_peaks"], out[f"sidecar_{name}_output_peaks"] = d["input_peaks"], d["output_peaks"]
        out[f"sidecar_{name}_traj"] = d["delay_trajectory"]
    shutil.rmtree(tmp)
    path = os.path.join(ROOT, "tests", "golden", "g15_vadataset.npz")
    np.savez_compressed(path, **out)
    print("g15 bytes:", os.path.getsize(path))


if __name__ == "__main__":
    main()
