"""CPU: the segment feeder (file layout + slicing of the reference's VADataset, code/dataset.py:129-293)."""
import os

import numpy as np
import pytest
from scipy.io import wavfile

from ntm_amd.feeder import SegmentFeeder, read_wav


def make_dataset(root, n_files=3, fs=44100, frames=(10000, 12345, 8000), stereo=False, traj=False):
    d = os.path.join(root, "Toy[1]_Set", "Test")
    os.makedirs(d)
    rng = np.random.default_rng(0)
    data = []
    for i in range(n_files):
        x = (rng.uniform(-0.5, 0.5, frames[i]) * 32767).astype(np.int16)
        t = (rng.uniform(-0.5, 0.5, frames[i]) * 32767).astype(np.int16)
        xi = np.stack([x, np.zeros_like(x)], 1) if stereo else x
        wavfile.write(os.path.join(d, f"input_{i + 7}_.wav"), fs, xi)
        wavfile.write(os.path.join(d, f"target_{i + 7}_.wav"), fs, t)
        if traj:
            np.save(os.path.join(d, f"trajectory_{i + 7}_.npy"), np.full(frames[i], 0.01 * (i + 1), np.float32))
        data.append((x.astype(np.float32) / 32768, t.astype(np.float32) / 32768))
    return os.path.join(root, "Toy[1]_Set"), data


def test_slicing_matches_reference_semantics(tmp_path):
    root, data = make_dataset(str(tmp_path), stereo=True, traj=True)
    f = SegmentFeeder(root, subset="test", length=3000, sync=0.01)          # START_OFFSET = 441
    assert f.fs == 44100
    want = [(0, 441 + 3000 * k) for k in range((10000 - 441) // 3000)] + \
           [(1, 441 + 3000 * k) for k in range((12345 - 441) // 3000)] + \
           [(2, 441 + 3000 * k) for k in range((8000 - 441) // 3000)]
    assert [(e["idx"], e["offset"]) for e in f.examples] == want and len(f) == 3 + 3 + 2
    x, t, meta = f[4]
    idx, off = want[4]
    assert x.shape == (2, 3000) and t.shape == (1, 3000)
    assert np.array_equal(x[0].numpy(), data[idx][0][off:off + 3000])
    assert np.array_equal(t[0].numpy(), data[idx][1][off:off + 3000])
    assert meta["input_name"] == f"input_{idx + 7}__[{off}:{off + 3000}].wav"
    assert meta["target_name"] == f"target_{idx + 7}__[{off}:{off + 3000}].wav"
    assert np.allclose(meta["delay_trajectory"].numpy(), 0.01 * (idx + 1))
    # whole-file segments when length is None (code/dataset.py:156-157)
    with pytest.raises(ValueError):
        SegmentFeeder(root, subset="test", length=9000)                     # longer than the shortest file
    batches = list(f.batches(3, device="cpu"))
    assert [b[0].shape[0] for b in batches] == [3, 3, 2] and batches[0][0].shape == (3, 1, 3000)
    assert batches[0][2].shape == (3, 1, 3000)
    # rank sharding covers every segment exactly once
    seen = [m["input_name"] for r in range(3) for b in f.batches(2, "cpu", r, 3) for m in b[3]]
    assert sorted(seen) == sorted(f[i][2]["input_name"] for i in range(len(f)))


def test_id_mismatch_and_wav_scaling(tmp_path):
    root, _ = make_dataset(str(tmp_path), n_files=1, frames=(5000,))
    a, fs = read_wav(os.path.join(root, "Test", "input_7_.wav"))
    assert a.dtype == np.float32 and a.shape == (1, 5000) and fs == 44100 and np.abs(a).max() <= 0.5
    os.rename(os.path.join(root, "Test", "target_7_.wav"), os.path.join(root, "Test", "target_8_.wav"))
    with pytest.raises(RuntimeError, match="non-matching file ids"):
        SegmentFeeder(root, subset="test", length=1000)
