"""Round 4: parity where the reference actually operates -- all 44 shipped checkpoints (32 distinct files), 10-second
segments (scripts/test-model-loss.sh:22: 441000 samples, not a multiple of the kernel's 64-sample tile), real-tape delay
lengths (code/test-model.py:222-230: max_delay = int(1.25 * measured * fs) ~ 11000) -- against goldens made by the
reference itself (tools/make_goldens_checkpoints.py).

Tolerance.  north_star: 1e-5 abs against the reference's PyTorch-CPU fp32 output.  Two of the shipped checkpoints do not
reproduce THEMSELVES to 1e-5 in the reference (fp32 vs the same network in fp64; batch of 3 vs batch of 1 in the same torch:
golden fields `*_y64`, `*_b3`): their dynamics amplify rounding differences (the zero-input warm-up of
GRU-...CHOWTAPE]_1 passes through a sensitive regime: 1.5e-2; the AKAI GRU drifts 2.9e-5 inside a 10-s segment).  The bar
is therefore, per golden:
    (1) teacher-forced: forward() from the REFERENCE's warm state                      |hip - ref32| < 1e-5   always
    (2) predict():  |hip - ref32| < 1e-5,  or -- only where the reference's own fp32 result is further than 2e-6 from the
        fp64 evaluation -- as close to the fp64 truth as the reference is:  |hip - y64| <= 2 |ref32 - y64|
Every number lands in gpurun_out/r04_checkpoint_parity.jsonl (copied to profiles/ and tabulated in DESIGN.md).
"""
import json
import os

import numpy as np
import pytest
import torch

from helpers import ROOT, load

pytestmark = pytest.mark.gpu

TOL = 1e-5
SELF_NOISE_FLOOR = 2e-6          # below this the reference reproduces itself and bar (2) is the plain 1e-5
TILE_B = 1040                    # > 1024 streams: "auto" takes the matrix-pipe kernel / the fused DiffDel step
LOG = os.path.join(ROOT, "gpurun_out", "r04_checkpoint_parity.jsonl")


@pytest.fixture(scope="module")
def ntm():
    import ntm_amd
    assert torch.cuda.is_available(), "these tests need the MI355X"
    ntm_amd._lib.lib()
    return ntm_amd


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def record(**row):
    os.makedirs(os.path.dirname(LOG), exist_ok=True)
    with open(LOG, "a") as f:
        f.write(json.dumps({k: (float(v) if isinstance(v, (np.floating, float)) else v) for k, v in row.items()}) + "\n")


def bar(y, ref32, y64, what):
    """-> (|y - ref32|, |y - y64|, |ref32 - y64|), asserting the module docstring's bar (2)."""
    d32 = float(np.abs(y - ref32).max())
    d64 = float(np.abs(y - y64).max())
    s = float(np.abs(ref32.astype(np.float64) - y64).max())
    if s <= SELF_NOISE_FLOOR:
        assert d32 < TOL, f"{what}: |hip - ref32| = {d32:.2e} (reference self-noise {s:.1e})"
    else:
        assert d32 < TOL or d64 <= 2.0 * s, f"{what}: |hip - ref32| = {d32:.2e}, |hip - f64| = {d64:.2e}, reference's own |ref32 - f64| = {s:.2e}"
    return d32, d64, s


def build(ntm, name, max_delay=None):
    sd = ntm.weights.load_state_dict(name)
    if name.startswith("GRU"):
        m = ntm.RNN(1, ntm.parse_hidden_size(name), 1, skip=False)
    else:
        m = ntm.DiffDelRNN(1, ntm.parse_hidden_size(name), 1, skip=False, max_delay=max_delay)
    m.load_state_dict(sd)
    return m.to("cuda").eval()


def _names():
    g = np.load(os.path.join(ROOT, "tests", "golden", "g19_checkpoints.npz"), allow_pickle=False)
    return [str(n) for n in g["names"]]


def tile(a, B=TILE_B):
    return dev(np.broadcast_to(a.reshape(1, 1, -1), (B, 1, a.size)).copy())


ROWS = [0, 519, TILE_B - 1]


@pytest.mark.parametrize("name", _names())
def test_g19_every_shipped_checkpoint(ntm, name):
    g = load("g19_checkpoints.npz")
    k = str(g["name_file"][list(g["names"]).index(name)])[:-4]
    T, TL = int(g["T"]), int(g["T_long"])
    xl = g["x_int16"].astype(np.float32) / 32768.0
    x = xl[:T]
    row = {"golden": "g19", "name": name, "blob": k}
    if name.startswith("GRU"):
        ref, y64, hw = g[k + "_y"], g[k + "_y64"].astype(np.float64), g[k + "_hwarm"]
        m = build(ntm, name)
        # B = 1: exactly the reference call ("auto" = the low-latency kernel at this batch size)
        y1 = m.predict(dev(x.reshape(1, 1, T))).cpu().numpy()[0, 0]
        row["lat_d32"], row["lat_d64"], row["ref_self"] = bar(y1, ref, y64, "predict B=1")
        # the golden stream tiled to 1040 streams: "auto" and forced mfma2 both run the matrix-pipe kernel
        for variant in ("auto", "mfma2"):
            m.kernel_variant = variant
            yb = m.predict(tile(x)).cpu().numpy()[ROWS, 0]
            assert np.array_equal(yb[0], yb[1]) and np.array_equal(yb[0], yb[2])         # same input -> same bits in every stream
            row[f"{variant}_d32"], row[f"{variant}_d64"], _ = bar(yb[0], ref, y64, f"predict B={TILE_B} {variant}")
        # teacher-forced: forward() from the reference's own warm state
        m.kernel_variant = "auto"
        for B in (1, TILE_B):
            m.initialize_hidden()
            m.hidden = dev(np.broadcast_to(hw, (1, B, 64)).copy())
            yt = m(tile(x, B)).cpu().numpy()[0, 0]
            row[f"forced_B{B}"] = float(np.abs(yt - ref).max())
            assert row[f"forced_B{B}"] < TOL, row
        row["ref_b3_vs_b1"] = float(g[k + "_b3"])
    else:
        cases = [("toy", int(g["max_delay_toy"]), T, g["d_toy"], "")]
        if k + "_y_real" in g.files:
            cases.append(("real", int(g["max_delay_real"]), TL, g["d_real"], "_real"))
        for tag, md, Tc, d, suf in cases:
            ref_y, ref_p = g[k + "_y" + suf], g[k + "_pre" + suf]
            y64, p64 = g[k + "_y64" + suf].astype(np.float64), g[k + "_pre64" + suf].astype(np.float64)
            m = build(ntm, name, md)
            assert m.diffdel.max_delay == md + 1
            y1, p1 = m.predict(dev(xl[:Tc].reshape(1, 1, Tc)), dev(d.reshape(1, 1, Tc)))
            row[f"{tag}_lat_y"], _, row[f"{tag}_ref_self"] = bar(y1.cpu().numpy()[0, 0], ref_y, y64, f"{tag} predict B=1 y")
            row[f"{tag}_lat_pre"], _, _ = bar(p1.cpu().numpy()[0, 0], ref_p, p64, f"{tag} predict B=1 pre_d")
            outs = {}
            for mode in ("fused", "two_pass"):
                m.delay_mode = mode
                yb, pb = m.predict(tile(xl[:Tc]), tile(d))
                yb, pb = yb.cpu().numpy()[ROWS, 0], pb.cpu().numpy()[ROWS, 0]
                assert np.array_equal(yb[0], yb[1]) and np.array_equal(yb[0], yb[2])
                outs[mode] = (yb[0], pb[0])
                row[f"{tag}_{mode}_y"], _, _ = bar(yb[0], ref_y, y64, f"{tag} predict B={TILE_B} {mode} y")
                row[f"{tag}_{mode}_pre"], _, _ = bar(pb[0], ref_p, p64, f"{tag} predict B={TILE_B} {mode} pre_d")
            # teacher-forced from the reference's warm state (hidden + the 1024 samples the warm-up left in the buffer; the
            # same for both delay lengths): 1e-5 for every checkpoint, and from the SAME state the fused step and the two-pass
            # step give the same bits (their predict()s differ in the last place only because a forced mode also picks the
            # GRU kernel of the B = 1 warm-up)
            forced = {}
            for mode, B in (("auto", 1), ("fused", TILE_B), ("two_pass", TILE_B)):
                m.delay_mode = mode
                m.initialize_hidden(B, m.max_delay)
                m.hidden = dev(np.broadcast_to(g[k + "_hwarm"], (1, B, 64)).copy())
                m.diffdel.buffer[:, 0, -1024:] = dev(g[k + "_bwarm"])
                yt, pt = m(tile(xl[:Tc], B), tile(d, B))
                forced[mode] = (yt.cpu().numpy()[ROWS if B > 1 else [0], 0], pt.cpu().numpy()[ROWS if B > 1 else [0], 0], m.diffdel.buffer.cpu().numpy()[0, 0])
                e = max(float(np.abs(forced[mode][0][0] - ref_y).max()), float(np.abs(forced[mode][1][0] - ref_p).max()))
                row[f"{tag}_forced_{mode}_B{B}"] = e
                assert e < TOL, row
            for a, b in zip(forced["fused"], forced["two_pass"]):
                assert np.array_equal(a, b)
    record(**row)


@pytest.mark.parametrize("tag", ["chow", "akai"])
def test_g20_gru_ten_second_segment(ntm, tag):
    """scripts/test-model-loss.sh:22: SEGMENT_LENGTH = 441000 = 6890 x 64 + 40 -- the ragged last tile of every kernel."""
    g = load("g20_operating_point.npz")
    name = str(g[f"gru_{tag}_weights"])
    x = g["gru_x_int16"].astype(np.float32) / 32768.0
    T = x.size
    assert T == 441000 and T % 64 == 40
    ref, y64 = g[f"gru_{tag}_y"], g[f"gru_{tag}_y64"].astype(np.float64)
    m = build(ntm, name)
    row = {"golden": "g20", "name": name, "T": T, "ref_b3_vs_b1": float(g[f"gru_{tag}_b3"])}
    y1 = m.predict(dev(x.reshape(1, 1, T))).cpu().numpy()[0, 0]
    row["lat_d32"], row["lat_d64"], row["ref_self"] = bar(y1, ref, y64, "B=1")
    # the reference's own chunk loop (code/model.py:236-244): 2048-sample forwards, bit-identical to one launch
    y1c = m.predict(dev(x.reshape(1, 1, T)), segment_length=2048).cpu().numpy()[0, 0]
    assert np.array_equal(y1c, y1)
    for variant in ("auto", "mfma2"):
        m.kernel_variant = variant
        yb = m.predict(tile(x)).cpu().numpy()[ROWS, 0]
        assert np.array_equal(yb[0], yb[1]) and np.array_equal(yb[0], yb[2])
        row[f"{variant}_d32"], row[f"{variant}_d64"], _ = bar(yb[0], ref, y64, f"B={TILE_B} {variant}")
    # forward + ESR sums in the same launch on the ragged length, against the reference's output as the target
    m.kernel_variant = "auto"
    tgt = tile(ref)
    yb, s = m.predict_esr(tile(x), tgt, skip=1024)
    e2 = ((ref[1024:].astype(np.float64) - yb[0, 0, 1024:].cpu().numpy().astype(np.float64)) ** 2).sum()
    t2 = (ref[1024:].astype(np.float64) ** 2).sum()
    got = s.cpu().numpy()
    assert abs(got[5, 0] / e2 - 1) < 1e-9 and abs(got[5, 1] / t2 - 1) < 1e-9
    row["esr_vs_reference_output"] = float((e2 / (T - 1024)) / (t2 / (T - 1024) + 1e-5))
    record(**row)


@pytest.mark.parametrize("tag", ["toy", "real"])
def test_g20_diffdel_long_and_real_tape_delay(ntm, tag):
    """DiffDelRNN.predict (code/model.py:618-653) at T = 65536 / D = 1847 and at the real-tape delay length
    T = 20000 / D = 11001 (code/test-model.py:222-230), through the fused step and the two-pass step (B = 1040, bit-identical
    to each other) and through the B = 1 path; final hidden state and delay buffer too."""
    g = load("g20_operating_point.npz")
    name = str(g[f"dd_{tag}_weights"])
    md = int(g[f"dd_{tag}_max_delay"])
    x = g[f"dd_{tag}_x_int16"].astype(np.float32) / 32768.0
    d = g[f"dd_{tag}_d"]
    T = x.size
    ref_y, ref_p = g[f"dd_{tag}_y"], g[f"dd_{tag}_pre"]
    y64, p64 = g[f"dd_{tag}_y64"].astype(np.float64), g[f"dd_{tag}_pre64"].astype(np.float64)
    m = build(ntm, name, md)
    row = {"golden": "g20", "name": name, "T": T, "D": md + 1, "d_min": float(d.min()), "d_max": float(d.max())}
    y1, p1 = m.predict(dev(x.reshape(1, 1, T)), dev(d.reshape(1, 1, T)))
    row["lat_y"], _, row["ref_self"] = bar(y1.cpu().numpy()[0, 0], ref_y, y64, "B=1 y")
    row["lat_pre"], _, _ = bar(p1.cpu().numpy()[0, 0], ref_p, p64, "B=1 pre_d")
    assert np.abs(m.diffdel.buffer.cpu().numpy()[0, 0] - g[f"dd_{tag}_buffer"]).max() < TOL
    assert np.abs(m.hidden.cpu().numpy()[0, 0] - g[f"dd_{tag}_hidden"]).max() < 10 * TOL          # 64 raw state values, not the head's mix
    # the reference's chunk loop: 2048-sample forwards, same bits
    y1c, p1c = m.predict(dev(x.reshape(1, 1, T)), dev(d.reshape(1, 1, T)), segment_length=2048)
    assert torch.equal(y1c, y1) and torch.equal(p1c, p1)
    outs = {}
    for mode in ("fused", "two_pass"):
        m.delay_mode = mode
        yb, pb = m.predict(tile(x), tile(d))
        buf = m.diffdel.buffer[ROWS, 0].cpu().numpy()
        yb, pb = yb.cpu().numpy()[ROWS, 0], pb.cpu().numpy()[ROWS, 0]
        assert np.array_equal(yb[0], yb[1]) and np.array_equal(yb[0], yb[2])
        outs[mode] = (yb[0], pb[0], buf[0])
        row[f"{mode}_y"], _, _ = bar(yb[0], ref_y, y64, f"B={TILE_B} {mode} y")
        row[f"{mode}_pre"], _, _ = bar(pb[0], ref_p, p64, f"B={TILE_B} {mode} pre_d")
        assert np.abs(buf[0] - g[f"dd_{tag}_buffer"]).max() < TOL
    # from the SAME warm state the two forms of the step give the same bits (outputs, hidden state, delay buffer)
    m.delay_mode = "two_pass"
    m.initialize_hidden(1, m.max_delay)
    m.warm_start()
    h0, b0 = m.hidden.clone(), m.diffdel.buffer.clone()
    same = {}
    for mode in ("fused", "two_pass"):
        m.delay_mode = mode
        m.initialize_hidden(TILE_B, m.max_delay)
        m.hidden = h0.expand(1, TILE_B, 64).contiguous()
        m.diffdel.buffer = b0.expand(TILE_B, 1, -1).contiguous()
        yb, pb = m(tile(x), tile(d))
        same[mode] = (yb[ROWS].cpu(), pb[ROWS].cpu(), m.hidden[:, ROWS].cpu(), m.diffdel.buffer[ROWS].cpu())
    for a, b in zip(same["fused"], same["two_pass"]):
        assert torch.equal(a, b)
    record(**row)


W_D = "DiffDelGRU-HS[64]-L[DCPreESR]-DS[ReelToReel_Dataset_MiniPulse100_CHOWTAPE_WOWFLUTTER]_BEST"


@pytest.mark.parametrize("bad", [302.5, float("nan")])
def test_warmup_call_checks_the_delay_range_in_every_mode(ntm, bad):
    """code/model.py:284 asserts max_delay >= max(dt) BEFORE the warm-up branch (:288).  Round 3's fused launch skipped the
    check in warm-up mode (it reads no delays there); warm-up calls now take the two-pass form in every mode.  A violating
    (or NaN) trajectory raises and leaves hidden-state-independent state -- the delay buffer -- untouched; a legal warm-up
    gives the same bits in all three modes: y == pre_d, buffer = the last D samples."""
    rng = np.random.default_rng(3)
    B, T = TILE_B, 512
    x = dev(rng.uniform(-0.5, 0.5, (B, 1, T)).astype(np.float32))
    d_ok = dev(np.full((B, 1, T), 100.25, np.float32))
    d_bad = d_ok.clone()
    d_bad[7, 0, 33] = bad
    outs = []
    for mode in ("auto", "fused", "two_pass"):
        m = build(ntm, W_D, 300)
        m.delay_mode = mode
        m.initialize_hidden(B, m.max_delay)
        buf0 = m.diffdel.buffer.clone()
        with pytest.raises(AssertionError):
            m(x, d_bad, warmup=True)
        assert torch.equal(m.diffdel.buffer, buf0)
        m.initialize_hidden(B, m.max_delay)
        y, pre = m(x, d_ok, warmup=True)
        assert torch.equal(y, pre) and torch.equal(m.diffdel.buffer[:, 0], pre[:, 0, -301:])
        y2, pre2 = m(x, d_ok)                       # and the step after the warm-up reads that buffer
        outs.append((pre.cpu(), y2.cpu(), pre2.cpu(), m.diffdel.buffer.cpu()))
    for o in outs[1:]:
        for a, b in zip(o, outs[0]):
            assert torch.equal(a, b)


def test_tcn_stream_chunks_bit_identical_and_bounded(ntm):
    """ntm_tcn_forward works through the batch in stream chunks (<= 1e9 floats per activation buffer): 6 streams of 2^23
    samples = 2 chunks of 3; same bits as each stream alone (one chunk each), scratch as ntm_tcn_scratch_floats says."""
    L = ntm._lib.lib()
    B, T = 6, 1 << 23
    assert L.ntm_tcn_chunk_streams(B, T, 32) == 3
    tcn = ntm.TCN().to("cuda")
    g = torch.Generator(device="cuda").manual_seed(9)
    x = torch.rand(B, 1, T, generator=g, device="cuda") - 0.5
    y = tcn(x)
    assert tcn._scratch.numel() == L.ntm_tcn_scratch_floats(B, T, 32) == 2 * (3 * T * 32 + 512)
    for b in (0, 2, 3, 5):
        assert torch.equal(tcn(x[b:b + 1]), y[b:b + 1])
    # and the bench's shape: 4096 x 65536 needs 2 x 3.8 GB now (9 chunks of 456 streams), 32768 x 65536 the same
    assert L.ntm_tcn_scratch_floats(32768, 65536, 32) == L.ntm_tcn_scratch_floats(4096, 65536, 32) <= 2 * 10**9 + 1024
