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


def test_reference_sidecar_format_stats_and_segment_peaks(tmp_path):
    """The reference's DelayAnalyzer caches a PICKLED DICT per file (code/utilities/utilities.py:327-335); the feeder
    reads it, keeps max/min/mean delay like the analyser (:191-193, :296-300, :341) and cuts the pulse indices per
    segment as code/dataset.py:262-279 does."""
    from ntm_amd.feeder import load_trajectory, segment_peaks
    root, _ = make_dataset(str(tmp_path), n_files=2, frames=(10000, 9000), stereo=True)
    d = os.path.join(root, "Test")
    pk_in = np.arange(50, 9000, 441)
    for i, (n, d0) in enumerate([(10000, 0.020), (9000, 0.030)]):
        traj = d0 + 0.002 * np.sin(np.arange(n) / 500.0)
        np.save(os.path.join(d, f"trajectory_{i + 7}_.npy"),
                {"input_peaks": pk_in, "input_meta": {"reconstruction_percentage": 0.0, "wiggle_percentage": 0.0},
                 "output_peaks": pk_in + 900 + i, "output_meta": {"reconstruction_percentage": 0.0, "wiggle_percentage": 0.0},
                 "delay_trajectory": traj})
    t0 = load_trajectory(os.path.join(d, "trajectory_7_.npy"))
    assert t0["delay_trajectory"].shape == (10000,) and np.array_equal(t0["output_peaks"], pk_in + 900)
    f = SegmentFeeder(root, subset="test", length=4000)
    trajs = [0.020 + 0.002 * np.sin(np.arange(10000) / 500.0), 0.030 + 0.002 * np.sin(np.arange(9000) / 500.0)]
    assert abs(f.max_delay - max(t.max() for t in trajs)) < 1e-12 and abs(f.min_delay - min(t.min() for t in trajs)) < 1e-12
    assert abs(f.mean_delay - np.mean([t.mean() for t in trajs])) < 1e-12
    x, t, meta = f[1]                                                  # file 0, offset 4000
    assert np.allclose(meta["delay_trajectory"].numpy(), trajs[0][4000:8000], atol=1e-7)
    first = np.where(pk_in >= 4000)[0][0]; last = np.where(pk_in <= 8000)[0][-1]
    assert np.array_equal(meta["input_peaks"], pk_in[first:last] - 4000)
    want_out = (pk_in + 900)[first:last] - 4000
    assert np.array_equal(meta["output_peaks"], want_out[want_out <= 4000])
    # segment past the last pulse: the reference then slices with the last pulse's VALUE (code/dataset.py:269-270)
    pi, po = segment_peaks(pk_in, pk_in + 900, 8000, 12000, 4000)
    assert np.array_equal(pi, pk_in[np.where(pk_in >= 8000)[0][0]:pk_in[-1]] - 8000)
    assert segment_peaks(pk_in, pk_in + 900, 9500, 13500, 4000) == (None, None)      # offset beyond the last pulse
    # peaks stay out of the collated batch metadata
    b = next(f.batches(2, device="cpu"))
    assert b[2].shape == (2, 1, 4000) and all("input_peaks" not in m for m in b[3])


def test_g11_demodulate_oracle_is_the_reference():
    """N3: the oracle's demodulate against the reference's own DelayAnalyzer.demodulate output (bit-exact)."""
    import oracle
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g11_demodulate.npz"))
    for i in range(int(g["n"])):
        dem = oracle.demodulate(g[f"out{i}"], g[f"x{i}"], g[f"y{i}"])
        assert np.array_equal(dem, g[f"dem{i}"]), i


def test_g12_pulse_analysis_matches_the_reference():
    """DelayAnalyzer.analyze_delay / _get_pulse_indices (code/utilities/utilities.py:343-406, :466-610) restated in
    feeder.find_pulses / analyze_delay: same pulse indices (incl. re-inserted and re-timed pulses), same metadata,
    same cubic-interpolated trajectory as the reference's own methods (golden g12)."""
    from ntm_amd.feeder import analyze_delay
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g12_delay_analysis.npz"))
    fs = int(g["fs"])
    for i in range(int(g["n"])):
        xi, yi, T, xm, ym = analyze_delay(g[f"in{i}"].astype(np.float64), g[f"out{i}"].astype(np.float64), fs)
        assert np.array_equal(xi, g[f"x_idx{i}"]) and np.array_equal(yi, g[f"y_idx{i}"]), i
        assert np.allclose([xm["reconstruction_percentage"], xm["wiggle_percentage"],
                            ym["reconstruction_percentage"], ym["wiggle_percentage"]], g[f"meta{i}"])
        assert T.shape == g[f"T{i}"].shape and np.abs(T - g[f"T{i}"]).max() < 1e-15, i


def test_stereo_dataset_without_sidecars_is_analysed_and_cached(tmp_path):
    """First use of a stereo dataset: pilots analysed, side-car written in the reference's format, statistics and
    per-segment trajectories available; second use loads the cache and gives the same numbers."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g12_delay_analysis.npz"))
    fs = int(g["fs"])
    d = tmp_path / "Set" / "Val"
    d.mkdir(parents=True)
    rng = np.random.default_rng(1)
    N = len(g["in0"])
    audio = (0.1 * rng.standard_normal(N)).astype(np.float32)
    wavfile.write(str(d / "input_3_a.wav"), fs, np.stack([audio, g["in0"]], 1))
    wavfile.write(str(d / "target_3_a.wav"), fs, np.stack([0.5 * audio, g["out0"]], 1))
    f = SegmentFeeder(str(tmp_path / "Set"), subset="val", length=20000)
    assert os.path.exists(str(d / "trajectory_3_a.npy"))
    assert abs(f.max_delay - g["T0"].max()) < 1e-12 and abs(f.mean_delay - g["T0"].mean()) < 1e-12
    x, t, meta = f[1]
    assert np.allclose(meta["delay_trajectory"].numpy(), g["T0"][20000:40000], atol=1e-8)
    f2 = SegmentFeeder(str(tmp_path / "Set"), subset="val", length=20000, analyze=False)      # from the cache
    assert abs(f2.max_delay - f.max_delay) < 1e-15 and np.array_equal(f2[1][2]["input_peaks"], meta["input_peaks"])


# ----------------------------------------------------------------------------- pinned to the reference (golden g15)
def rebuild_g15_tree(g, root):
    """The dataset tree tools/make_goldens_dataset.py ran the reference's VADataset over, rebuilt from the arrays in
    the golden file (same float32 WAVs; no side-cars: the feeder must analyse the pilot channels itself)."""
    ds = os.path.join(root, "Toy[Set]_A")
    os.makedirs(os.path.join(ds, "Test"))
    os.makedirs(os.path.join(ds, "Train"))
    fs = int(g["fs"])
    for name in g["names"]:
        wavfile.write(os.path.join(ds, "Test", f"input_{name}.wav"), fs, g[f"file_{name}_input"])
        wavfile.write(os.path.join(ds, "Test", f"target_{name}.wav"), fs, g[f"file_{name}_target"])
    n = int(g["file_train_3_other_len"])
    wavfile.write(os.path.join(ds, "Train", "input_3_other.wav"), fs, g["file_3_first_input"][:n])
    wavfile.write(os.path.join(ds, "Train", "target_3_other.wav"), fs, g["file_3_first_target"][:n])
    return ds


@pytest.mark.parametrize("tag", ["a", "b"])
def test_g15_feeder_equals_reference_vadataset(tmp_path, tag):
    """SegmentFeeder against what the reference's VADataset returned for the same tree (code/dataset.py:174-293 slicing,
    :348-429 items; DelayAnalyzer statistics; per-segment pulse indices; meta strings): '[' ']' in the path, ids that
    sort as strings (input_10_ before input_3_), stereo audio + pilot, a last file that is not a whole number of
    segments, sync > 0 (case b), a colliding id in another subset directory."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g15_vadataset.npz"))
    ds = rebuild_g15_tree(g, str(tmp_path))
    L, sync = int(g[f"{tag}_length"]), float(g[f"{tag}_sync"])
    f = SegmentFeeder(ds, subset="test", length=L, sync=sync)
    assert [os.path.basename(p) for p in f.input_files] == list(g[f"{tag}_input_files"])
    assert len(f) == int(g[f"{tag}_n"])
    assert [[e["idx"], e["offset"]] for e in f.examples] == g[f"{tag}_examples"].tolist()
    mn, mean, mx = g[f"{tag}_delay_stats"]
    assert abs(f.min_delay - mn) < 1e-12 and abs(f.mean_delay - mean) < 1e-12 and abs(f.max_delay - mx) < 1e-12
    assert abs(f.minutes - float(g[f"{tag}_minutes"])) < 1e-12
    for name in g["names"]:          # the feeder's own pulse analysis == the side-car the reference wrote
        sc = np.load(os.path.join(ds, "Test", f"trajectory_{name}.npy"), allow_pickle=True).item()
        assert np.array_equal(sc["input_peaks"], g[f"sidecar_{name}_input_peaks"])
        assert np.array_equal(sc["output_peaks"], g[f"sidecar_{name}_output_peaks"])
        assert np.abs(sc["delay_trajectory"] - g[f"sidecar_{name}_traj"]).max() < 1e-12
    for i in range(len(f)):
        x, t, meta = f[i]
        idx, off = g[f"{tag}_examples"][i]
        fname = os.path.basename(f.input_files[idx])[len("input_"):-len(".wav")]
        # the generator verified that the reference's item is exactly this slice of the file pair
        assert np.array_equal(x.numpy(), g[f"file_{fname}_input"].T[:, off:off + L])
        assert np.array_equal(t.numpy(), g[f"file_{fname}_target"].T[:, off:off + L])
        nin, ntg, ntr, sin_, stg, str_ = g[f"{tag}_{i}_shape_sums"]
        assert (x.shape[-1], t.shape[-1], len(meta["delay_trajectory"])) == (nin, ntg, ntr)
        assert abs(float(x.double().sum()) - sin_) < 1e-9 and abs(float(t.double().sum()) - stg) < 1e-9
        assert abs(float(meta["delay_trajectory"].double().sum()) - str_) < 1e-6 * abs(str_)     # fp32 store of the trajectory
        assert [meta["input_name"], meta["target_name"]] == list(g[f"{tag}_{i}_names"])
        for k in ("input_peaks", "output_peaks"):
            want = g[f"{tag}_{i}_{k}"]
            got = meta[k]
            assert (got is None and list(want) == [-1]) or np.array_equal(np.asarray(got), want), (i, k)
        if f"{tag}_{i}_traj" in g.files:
            assert np.abs(meta["delay_trajectory"].numpy() - g[f"{tag}_{i}_traj"]).max() < 1e-8


def test_sidecar_lookup_is_by_file_name_not_by_id(tmp_path):
    """subset "full": equal ids in Train/ and Test/ must each get their OWN trajectory side-car (the reference derives
    the side-car path from the input file's name and directory, code/utilities/utilities.py:273-275)."""
    root = os.path.join(str(tmp_path), "Set")
    fs = 44100
    rng = np.random.default_rng(3)
    for sub, val in (("Train", 0.01), ("Test", 0.02)):
        os.makedirs(os.path.join(root, sub))
        a = rng.uniform(-0.3, 0.3, (6000, 2)).astype(np.float32)
        wavfile.write(os.path.join(root, sub, "input_5_x.wav"), fs, a)
        wavfile.write(os.path.join(root, sub, "target_5_x.wav"), fs, a)
        np.save(os.path.join(root, sub, "trajectory_5_x.npy"), np.full(6000, val, np.float32))
    f = SegmentFeeder(root, subset="full", length=3000, analyze=False)
    got = {os.path.basename(os.path.dirname(f.input_files[f.examples[i]["idx"]])): float(f[i][2]["delay_trajectory"][0])
           for i in range(len(f))}
    assert abs(got["Train"] - 0.01) < 1e-7 and abs(got["Test"] - 0.02) < 1e-7


def test_fraction_and_shuffle_select_examples_like_create_fractional_patches(tmp_path):
    """code/dataset.py:295-341 for one device configuration: int(len * fraction) examples -- the FIRST ones without shuffling,
    drawn WITH replacement by np.random.randint when shuffling (repeatable here through `seed`; the global generator, as
    upstream, without); a fraction that selects nothing raises ValueError; batches / sharding follow the selection."""
    root, data = make_dataset(str(tmp_path))
    full = SegmentFeeder(root, subset="test", length=2000)
    assert len(full) == 5 + 6 + 4
    half = SegmentFeeder(root, subset="test", length=2000, fraction=0.5)
    assert len(half) == 7 and [(e["idx"], e["offset"]) for e in half.examples] == [(e["idx"], e["offset"]) for e in full.examples[:7]]
    a = SegmentFeeder(root, subset="test", length=2000, fraction=0.6, shuffle=True, seed=11)
    b = SegmentFeeder(root, subset="test", length=2000, fraction=0.6, shuffle=True, seed=11)
    pick = np.random.RandomState(11).randint(0, high=15, size=9)
    assert len(a) == 9 and [(e["idx"], e["offset"]) for e in a.examples] == [(full.examples[i]["idx"], full.examples[i]["offset"]) for i in pick]
    assert [(e["idx"], e["offset"]) for e in a.examples] == [(e["idx"], e["offset"]) for e in b.examples]
    np.random.seed(5)
    c = SegmentFeeder(root, subset="test", length=2000, shuffle=True)          # upstream's unseeded draw = the global generator
    np.random.seed(5)
    assert [(e["idx"], e["offset"]) for e in c.examples] == [(full.examples[i]["idx"], full.examples[i]["offset"])
                                                             for i in np.random.randint(0, high=15, size=15)]
    with pytest.raises(ValueError, match="set too low"):
        SegmentFeeder(root, subset="test", length=2000, fraction=0.01)
    # a shuffled selection is not a set of runs of consecutive segments: batches still deliver exactly the selected items
    got = np.concatenate([x.numpy()[:, 0] for x, _, _, _ in a.batches(4, device="cpu")])
    want = np.stack([data[e["idx"]][0][e["offset"]:e["offset"] + 2000] for e in a.examples])
    assert np.array_equal(got, want)


@pytest.mark.parametrize("dtype", [np.int16, np.int32, np.uint8, np.float32, np.float64])
@pytest.mark.parametrize("channels", [1, 2])
def test_read_wav_formats_are_decoded_as_before(tmp_path, dtype, channels):
    """read_wav maps the file and de-interleaves it in one torch pass (round 5: the decode bounds the evaluation command);
    the values must be those of the plain numpy definition -- integer PCM scaled by 2^(bits-1) like
    torchaudio.load(normalize=True), uint8 offset by 128, float64 rounded to float32 -- for mono and stereo files."""
    from scipy.io import wavfile
    from ntm_amd.feeder import read_wav
    rng = np.random.default_rng(7)
    N = 5003
    if dtype == np.uint8:
        a = rng.integers(0, 256, (N, channels)).astype(dtype)
        want = (a.astype(np.float32) - 128.0) / 128.0
    elif dtype in (np.int16, np.int32):
        info = np.iinfo(dtype)
        a = rng.integers(info.min, info.max, (N, channels), endpoint=True).astype(dtype)
        a[0], a[1] = info.min, info.max
        want = a.astype(np.float32) / np.float32(-float(info.min))
    else:
        a = rng.uniform(-1, 1, (N, channels)).astype(dtype)
        want = a.astype(np.float32)
    path = str(tmp_path / "f.wav")
    wavfile.write(path, 22050, a[:, 0] if channels == 1 else a)
    got, fs = read_wav(path)
    assert fs == 22050 and got.dtype == np.float32 and got.shape == (channels, N) and got.flags["C_CONTIGUOUS"]
    assert np.array_equal(got, want.T)
