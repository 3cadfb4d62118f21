"""Round 4: parity where the reference actually operates -- all 44 shipped checkpoints (32 distinct files), 10-second
segments (scripts/test-model-loss.sh:22: 441000 samples, not a multiple of the kernel's 64-sample tile), real-tape delay
lengths (code/test-model.py:222-230: max_delay = int(1.25 * measured * fs) ~ 11000) -- against goldens made by the
reference itself (tools/make_goldens_checkpoints.py).

Tolerance.  north_star: 1e-5 abs against the reference's PyTorch-CPU fp32 output.  Three of the 32 distinct checkpoints do
not allow that from ANY independent float32 implementation on these inputs (two more sit at the edge: 1.0e-5 and 1.3e-5):
their dynamics amplify rounding.  The goldens
measure it twice, from the reference network itself and independently of this build: `*_y64` (the same torch modules in
float64: |ref32 - y64| is the reference's own rounding error) and `*_gate_noise` (max output change when every gate value
carries the error of a 1-ulp float32 sigmoid / tanh, 32 draws, tools/make_goldens_checkpoints.py gate_noise):
    GRU-...-L[DCPreESR]-DS[...CHOWTAPE]_1        zero-input warm-up passes a sensitive regime: 1.5e-2 either way
    DiffDelGRU-...-L[ESR]-DS[...AKAI...]_3       1-ulp gate noise -> 1.6e-5 (median), 2.7e-5 (max)
    GRU-...-L[DCPreESR]-DS[...AKAI...]_BEST      fine on 4096 samples, 2.9e-5 / 3.1e-5 inside a 10-s segment
The bar, per golden:
    (1) teacher-forced: forward() from the REFERENCE's warm state, 4096 samples        |hip - ref32| < 1e-5
        for every checkpoint but the one whose 1-ulp gate noise alone exceeds it (..._AKAI..._3: bar (2))
    (2) predict():  |hip - ref32| < 1e-5;  where the checkpoint's own measures exceed 1e-5 -- max(2 |ref32 - y64|, gate noise)
        -- the result must instead be as close to the float64 truth as those measures: |hip - y64| <= max(2 |ref32 - y64|,
        gate_noise.max()).  For the other 28 checkpoints (2) IS the plain 1e-5.
Every number lands in gpurun_out/r06_checkpoint_parity.jsonl (copied to profiles/; summarised in DESIGN.md 2 and 4).
"""
import json
import os

import numpy as np
import pytest
import torch

from helpers import ROOT, bench_record, load

pytestmark = pytest.mark.gpu

TOL = 1e-5
TILE_B = 1040                    # > 1024 streams: "auto" takes the matrix-pipe kernel / the fused DiffDel step
LOG = os.path.join(ROOT, "gpurun_out", "r06_checkpoint_parity.jsonl")


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


def bar(y, ref32, y64, gate_noise, what):
    """-> (|y - ref32|, |y - y64|, the checkpoint's own allowance), asserting the module docstring's bar (2)."""
    d32 = float(np.abs(y - ref32).max())
    d64 = float(np.abs(y - y64).max())
    own = max(2.0 * float(np.abs(ref32.astype(np.float64) - y64).max()), float(np.max(gate_noise)))
    if own <= TOL:
        assert d32 < TOL, f"{what}: |hip - ref32| = {d32:.2e}"
    else:
        assert d32 < TOL or d64 <= own, f"{what}: |hip - ref32| = {d32:.2e}, |hip - f64| = {d64:.2e}, the checkpoint's own allowance {own:.2e}"
    return d32, d64, own


def opt_in_engine(m, x, ref32, y64, own, exact_d64, what, engine="f16x3"):
    """The opt-in split engines through the same goldens as the exact engine: f16x3 (NTM_GRU_F16X3: W.h as three fp16 hi/lo
    products, the 2^-22 lo.lo term dropped -- NARROWER than fp32, never the headline) and, round 6, bf16x3 (NTM_GRU_BF16X3: W and h
    as three bf16 pieces each, i.e. the fp32 operands exactly, eight of the nine partial products -- operand-exact).  Recorded:
    |hip - ref32| and |hip - f64| beside the exact engine's.  Asserted: no further from the float64 truth than the exact-fp32
    engine is allowed to be -- inside 1e-5 of the reference, or inside the checkpoint's own allowance, or within a factor 4
    (f16x3) / 2 (bf16x3) of what the exact engine measures on the same golden (the sensitive checkpoints amplify ANY rounding
    change alike)."""
    m.kernel_variant = engine
    yb = m.predict(tile(x)).cpu().numpy()[ROWS, 0]
    m.kernel_variant = "auto"
    assert np.array_equal(yb[0], yb[1]) and np.array_equal(yb[0], yb[2])
    d32, d64 = float(np.abs(yb[0] - ref32).max()), float(np.abs(yb[0] - y64).max())
    slack = 4.0 if engine == "f16x3" else 2.0
    assert d32 < TOL or d64 <= max(own, slack * exact_d64), f"{what} {engine}: |hip - ref32| = {d32:.2e}, |hip - f64| = {d64:.2e} (exact engine {exact_d64:.2e}, allowance {own:.2e})"
    return d32, d64


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
        ref, y64, hw, gn = g[k + "_y"], g[k + "_y64"].astype(np.float64), g[k + "_hwarm"], g[k + "_gate_noise"]
        m = build(ntm, name)
        # B = 1: exactly the reference call ("auto" = the low-latency kernel at this batch size)
        y1 = m.predict(dev(x.reshape(1, 1, T))).cpu().numpy()[0, 0]
        row["lat_d32"], row["lat_d64"], row["own"] = bar(y1, ref, y64, gn, "predict B=1")
        # the golden stream tiled to 1040 streams: "auto" and forced mfma2 both run the matrix-pipe kernel
        for variant in ("auto", "mfma2"):
            m.kernel_variant = variant
            yb = m.predict(tile(x)).cpu().numpy()[ROWS, 0]
            assert np.array_equal(yb[0], yb[1]) and np.array_equal(yb[0], yb[2])         # same input -> same bits in every stream
            row[f"{variant}_d32"], row[f"{variant}_d64"], _ = bar(yb[0], ref, y64, gn, f"predict B={TILE_B} {variant}")
        row["f16x3_d32"], row["f16x3_d64"] = opt_in_engine(m, x, ref, y64, row["own"], row["auto_d64"], f"{name} predict B={TILE_B}")
        row["bf16x3_d32"], row["bf16x3_d64"] = opt_in_engine(m, x, ref, y64, row["own"], row["auto_d64"], f"{name} predict B={TILE_B}", "bf16x3")
        # teacher-forced: forward() from the reference's own warm state
        m.kernel_variant = "auto"
        for B in (1, TILE_B):
            m.initialize_hidden()
            m.hidden = dev(np.broadcast_to(hw, (1, B, 64)).copy())
            yt = m(tile(x, B)).cpu().numpy()[0, 0]
            row[f"forced_B{B}"] = float(np.abs(yt - ref).max())
            assert row[f"forced_B{B}"] < TOL, row
        row["ref_b3_vs_b1"], row["ref_self"] = float(g[k + "_b3"]), float(np.abs(ref - y64).max())
        row["gate_noise_median"], row["gate_noise_max"] = float(np.median(gn)), float(gn.max())
    else:
        cases = [("toy", int(g["max_delay_toy"]), T, g["d_toy"], "")]
        if k + "_y_real" in g.files:
            cases.append(("real", int(g["max_delay_real"]), TL, g["d_real"], "_real"))
        for tag, md, Tc, d, suf in cases:
            ref_y, ref_p = g[k + "_y" + suf], g[k + "_pre" + suf]
            y64, p64 = g[k + "_y64" + suf].astype(np.float64), g[k + "_pre64" + suf].astype(np.float64)
            gn = g[k + "_gate_noise" + suf]
            row[f"{tag}_ref_self"], row[f"{tag}_gate_noise_median"], row[f"{tag}_gate_noise_max"] = float(np.abs(ref_p - p64).max()), float(np.median(gn)), float(gn.max())
            m = build(ntm, name, md)
            assert m.diffdel.max_delay == md + 1
            y1, p1 = m.predict(dev(xl[:Tc].reshape(1, 1, Tc)), dev(d.reshape(1, 1, Tc)))
            row[f"{tag}_lat_y"], _, row[f"{tag}_own"] = bar(y1.cpu().numpy()[0, 0], ref_y, y64, gn, f"{tag} predict B=1 y")
            row[f"{tag}_lat_pre"], _, _ = bar(p1.cpu().numpy()[0, 0], ref_p, p64, gn, f"{tag} predict B=1 pre_d")
            outs = {}
            for mode in ("fused", "two_pass"):
                m.delay_mode = mode
                yb, pb = m.predict(tile(xl[:Tc]), tile(d))
                yb, pb = yb.cpu().numpy()[ROWS, 0], pb.cpu().numpy()[ROWS, 0]
                assert np.array_equal(yb[0], yb[1]) and np.array_equal(yb[0], yb[2])
                outs[mode] = (yb[0], pb[0])
                row[f"{tag}_{mode}_y"], _, _ = bar(yb[0], ref_y, y64, gn, f"{tag} predict B={TILE_B} {mode} y")
                row[f"{tag}_{mode}_pre"], _, _ = bar(pb[0], ref_p, p64, gn, f"{tag} predict B={TILE_B} {mode} pre_d")
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
                e, _, _ = bar(forced[mode][1][0], ref_p, p64, gn, f"{tag} forced {mode} pre_d")
                e = max(e, bar(forced[mode][0][0], ref_y, y64, gn, f"{tag} forced {mode} y")[0])
                row[f"{tag}_forced_{mode}_B{B}"] = e
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
    ref, y64, gn = g[f"gru_{tag}_y"], g[f"gru_{tag}_y64"].astype(np.float64), g[f"gru_{tag}_gate_noise"]
    m = build(ntm, name)
    row = {"golden": "g20", "name": name, "T": T, "ref_b3_vs_b1": float(g[f"gru_{tag}_b3"]), "ref_self": float(np.abs(ref - y64).max()),
           "gate_noise_median": float(np.median(gn)), "gate_noise_max": float(gn.max())}
    y1 = m.predict(dev(x.reshape(1, 1, T))).cpu().numpy()[0, 0]
    row["lat_d32"], row["lat_d64"], row["own"] = bar(y1, ref, y64, gn, "B=1")
    # the reference's own chunk loop (code/model.py:236-244): 2048-sample forwards, bit-identical to one launch
    y1c = m.predict(dev(x.reshape(1, 1, T)), segment_length=2048).cpu().numpy()[0, 0]
    assert np.array_equal(y1c, y1)
    for variant in ("auto", "mfma2"):
        m.kernel_variant = variant
        yb = m.predict(tile(x)).cpu().numpy()[ROWS, 0]
        assert np.array_equal(yb[0], yb[1]) and np.array_equal(yb[0], yb[2])
        row[f"{variant}_d32"], row[f"{variant}_d64"], _ = bar(yb[0], ref, y64, gn, f"B={TILE_B} {variant}")
    row["f16x3_d32"], row["f16x3_d64"] = opt_in_engine(m, x, ref, y64, row["own"], row["auto_d64"], f"{name} 441000 samples B={TILE_B}")
    row["bf16x3_d32"], row["bf16x3_d64"] = opt_in_engine(m, x, ref, y64, row["own"], row["auto_d64"], f"{name} 441000 samples B={TILE_B}", "bf16x3")
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
    y64, p64, gn = g[f"dd_{tag}_y64"].astype(np.float64), g[f"dd_{tag}_pre64"].astype(np.float64), g[f"dd_{tag}_gate_noise"]
    m = build(ntm, name, md)
    row = {"golden": "g20", "name": name, "T": T, "D": md + 1, "d_min": float(d.min()), "d_max": float(d.max())}
    y1, p1 = m.predict(dev(x.reshape(1, 1, T)), dev(d.reshape(1, 1, T)))
    row["lat_y"], _, row["own"] = bar(y1.cpu().numpy()[0, 0], ref_y, y64, gn, "B=1 y")
    row["lat_pre"], _, _ = bar(p1.cpu().numpy()[0, 0], ref_p, p64, gn, "B=1 pre_d")
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
        row[f"{mode}_y"], _, _ = bar(yb[0], ref_y, y64, gn, f"B={TILE_B} {mode} y")
        row[f"{mode}_pre"], _, _ = bar(pb[0], ref_p, p64, gn, f"B={TILE_B} {mode} pre_d")
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
    """ntm_tcn_forward works through a batch whose activations exceed 8 GB in stream chunks on two lanes (own buffers, own
    HIP stream each, forked from / joined to the caller's stream): 8 streams of 2^23 samples = 8 chunks of 1; same bits as
    each stream alone (one chunk, the caller's stream), scratch as ntm_tcn_scratch_floats says, and the call is ordered on
    the caller's stream like any other (x written just before, y read right after, on a side stream)."""
    L = ntm._lib.lib()
    B, T = 8, 1 << 23
    assert L.ntm_tcn_chunk_streams(B, T, 32) == 1 and L.ntm_tcn_chunk_streams(3, T, 32) == 3
    tcn = ntm.TCN().to("cuda")
    g = torch.Generator(device="cuda").manual_seed(9)
    x = torch.rand(B, 1, T, generator=g, device="cuda") - 0.5
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        x2 = x * 0.5                                   # produced on `side` right before the call ...
        y2 = tcn(x2)
        s2 = y2.double().sum(dim=2)                    # ... and consumed on `side` right after it
    torch.cuda.current_stream().wait_stream(side)
    assert tcn._scratch.numel() == L.ntm_tcn_scratch_floats(B, T, 32) == 4 * (T * 32 + 512)
    y = tcn(x)
    for b in (0, 2, 3, 7):
        assert torch.equal(tcn(x[b:b + 1]), y[b:b + 1])
        assert torch.equal(tcn(x2[b:b + 1]), y2[b:b + 1])
    assert torch.equal(s2, y2.double().sum(dim=2))
    y3 = tcn(x[:3])                                    # 3 streams: one chunk
    assert torch.equal(y3, y[:3])
    assert L.ntm_tcn_scratch_floats(32768, 65536, 32) <= 2 * 10**9 + 2048 >= L.ntm_tcn_scratch_floats(4096, 65536, 32)


# ----------------------------------------------------------------------------- BASELINE configs[4] at its real shapes, on the one GPU
FULL = pytest.mark.skipif(os.environ.get("NTM_SKIP_FULL") == "1", reason="full-size passes skipped on request")


def _bench(args, **env_extra):
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    return bench_record(r.stdout)[1]            # (the compact line's length and agreement with the record are checked there)


@FULL
def test_full_size_eight_rank_dry_run_on_one_gpu():
    """The first time BASELINE configs[4]'s REAL shapes meet the launcher must not be on the driver's 8-GPU node: eight ranks
    x 4096 x 65 536 (weak: 32 768 segments) and four ranks x 8192 (strong: --total-batch 32768), all sharing this box's GPU
    over gloo (8 x 5.4 GB resident), next to the one-rank job.  Checked: rank/device table, segment totals, throughput
    arithmetic, non-zero job ESR identical in every timed step, bitwise determinism of the output; rank 0 of the weak job holds
    the one-rank job's data (stream 0 = golden g6 against the REFERENCE's output, same number in both).
    Reference counterpart: none (scripts/sbatch-train-exp1a.sh:7 runs replicas only)."""
    common = ["--steps", "2", "--warmup", "1", "--no-extra", "--other", "off"]
    common = common + ["--cpu-sample", "small"]      # (the CPU leg's 64 x 8192 shape and second repetitions: the default line has them)
    weak = _bench(["--gpus", "8"] + common, NTM_DIST_BACKEND="gloo")
    assert weak["n_gpus"] == 8 and weak["ranks"] == 8 and weak["backend"] == "gloo" and weak["rccl_ranks"] == 0
    assert [d["rank"] for d in weak["rank_devices"]] == list(range(8)) and all(d["device"] == 0 for d in weak["rank_devices"])
    assert weak["config"]["segments_total"] == 32768 == weak["checks"]["segments"] and weak["config"]["segments_rank0"] == 4096
    assert weak["checks"]["job_esr"] > 0 and weak["checks"]["every_timed_step_same_loss"] is True
    assert weak["checks"]["last_output_equals_first_pass_bitwise"] is True and weak["scaling"] == "weak"
    assert abs(weak["value"] - 32768 * 65536 / (weak["ms_per_step"] * 1e-3)) < 1e-6 * weak["value"]
    strong = _bench(["--gpus", "4", "--scaling", "strong", "--total-batch", "32768"] + common, NTM_DIST_BACKEND="gloo")
    assert strong["ranks"] == 4 and strong["config"]["segments_total"] == 32768 and strong["config"]["segments_rank0"] == 8192
    assert strong["checks"]["segments"] == 32768 and strong["scaling"] == "strong"
    one = _bench(["--gpus", "1"] + common)
    assert one["checks"]["segments"] == 4096 and one["rccl_ranks"] == 0 and one["backend"].startswith("none")
    # rank 0 of the weak job holds exactly the one-rank job's data: the per-rank part of the loss must agree; the job-wide
    # sums are sums over ranks, so the weak job's err^2 is larger than one rank's
    assert weak["checks"]["job_sum_err2"] > one["checks"]["job_sum_err2"] > 0
    assert weak["checks"]["stream0_vs_reference_max_abs"] == one["checks"]["stream0_vs_reference_max_abs"] < TOL
    # round 5: an N > 1 line is GRADEABLE by its own keys -- the CPU path timed in the same run, rank 0's streams against the
    # oracle, the per-GPU roofline with counter traffic of rank 0's launch (measured after the process group is gone) and the
    # job-wide aggregate; the metric and the first 120 characters of the workload name N, the scaling mode and the per-GPU batch
    for line, n, b in ((weak, 8, 4096), (strong, 4, 8192), (one, 1, 4096)):
        assert line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["cores"] >= 1 and line["cpu_baseline"]["kind"] == "port"
        assert line["speedup_vs_cpu_baseline"] == line["value"] / line["cpu_baseline"]["value"]
        c = line["checks"]
        assert c["streams_vs_oracle"]["max_abs"] < TOL and len(c["streams_vs_oracle"]["rows"]) == 8 and c["streams_vs_oracle"]["samples_each"] == 65536
        assert c["esr_sums_vs_oracle"]["max_rel"] < 1e-9
        r = line["roofline"]
        assert r["segments_in_launch"] == b and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0.05 < r["frac"] < 1.0
        assert abs(r["achieved"] - 25088.0 * b * 65536 / (r["kernel_ms"] * 1e-3) / 1e12) < 1e-6 * r["achieved"]
        assert r["traffic"] is not None and 1.0 <= r["traffic"] / (12.0 * b * 65536) < 1.01, r["traffic_source"]
        assert r["algorithmic_bytes"] == 12.0 * b * 65536
        g = r["aggregate"]
        assert g["n_gpus"] == n and g["peak"] == n * r["peak"] and len(g["kernel_ms_by_rank"]) == n
        assert abs(g["frac"] - g["achieved"] / g["peak"]) < 1e-12
        slow = max(k["kernel_ms"] for k in g["kernel_ms_by_rank"])
        assert abs(g["achieved"] - 25088.0 * 32768 / 8 * (8 if n > 1 else 1) * 65536 / (slow * 1e-3) / 1e12) < 1e-6 * g["achieved"]
        assert f"{n} GPU" in line["metric"] and line["metric"].startswith("audio samples/sec (44.1 kHz) GRU-HS[64], batch=")
        head = line["config"]["workload"][:120]
        assert f"on {n} GPU" in head and "predict+ESR fused" in head and "warm-cache on" in head and "f32" in head
        assert len(line["config"]["workload"]) <= 120
        assert line["value_no_warm_cache"] > 0 and line["build"]["compiler"]["hip"] and line["build"]["kernels"]
        assert all(k["scratch_bytes"] == 0 for k in line["build"]["kernels"])
    assert "4096/GPU" in weak["config"]["workload"] and "weak" in weak["metric"] and "32768 segments" in weak["metric"]
    assert "8192 on rank 0" in strong["config"]["workload"] and "strong" in strong["metric"] and "batch=8192x65536 per GPU" in strong["metric"]
    assert one["metric"] == "audio samples/sec (44.1 kHz) GRU-HS[64], batch=4096x65536, 1 GPU"
    assert abs(one["roofline"]["aggregate"]["frac"] - one["roofline"]["frac"]) < 1e-9


def test_warm_cache_contract_under_data_writes(ntm):
    """The warm-start cache is keyed on torch's version counters, which writes through `.data` by-pass.  Contract: the class
    default is OFF (a model built by hand always recomputes: `.data` writes are seen), harness.build_model turns it ON for the
    evaluation path, and there a `.data` write needs invalidate_warm_cache() -- shown here both ways against the oracle."""
    import oracle
    from helpers import state_dict_np
    W_G = "GRU-HS[64]-L[DCPreESR]-DS[ReelToReel_Dataset_MiniPulse100_CHOWTAPE]_BEST"
    rng = np.random.default_rng(8)
    x = rng.uniform(-0.5, 0.5, (2, 600)).astype(np.float32)

    def want(scale):
        sd = state_dict_np(W_G)
        sd["GRU.bias_hh_l0"] = (sd["GRU.bias_hh_l0"] * np.float32(scale)).astype(np.float32)
        return oracle.gru_predict(oracle.Weights.from_state_dict(sd), x)[0]

    by_hand = build(ntm, W_G)
    assert by_hand.warm_cache is False
    assert np.abs(by_hand.predict(dev(x).unsqueeze(1)).cpu().numpy()[:, 0] - want(1.0)).max() < TOL
    by_hand.GRU.bias_hh_l0.data.mul_(1.5)                         # a write torch's version counter does not see
    assert np.abs(by_hand.predict(dev(x).unsqueeze(1)).cpu().numpy()[:, 0] - want(1.5)).max() < TOL and by_hand._warm is None

    cached = ntm.harness.build_model(W_G)
    assert cached.warm_cache is True
    cached.predict(dev(x).unsqueeze(1))
    v = cached.GRU.bias_hh_l0._version
    cached.GRU.bias_hh_l0.data.mul_(1.5)
    assert cached.GRU.bias_hh_l0._version == v                    # ... which is why the key cannot follow it
    stale = cached.predict(dev(x).unsqueeze(1)).cpu().numpy()[:, 0]
    assert np.abs(stale - want(1.5)).max() > 1e-4                 # new weights, OLD warm state: the documented failure
    cached.invalidate_warm_cache()
    assert np.abs(cached.predict(dev(x).unsqueeze(1)).cpu().numpy()[:, 0] - want(1.5)).max() < TOL
    with torch.no_grad():
        cached.GRU.bias_hh_l0.mul_(1.0 / 1.5)                     # an in-place op on the Parameter itself IS seen
    assert np.abs(cached.predict(dev(x).unsqueeze(1)).cpu().numpy()[:, 0] - want(1.0)).max() < 2 * TOL


@FULL
def test_fused_diffdel_full_size_at_the_real_tape_delay_length(ntm):
    """BASELINE configs[2]'s batch (4096 distinct streams x 65 536 samples) at the REAL-tape delay-line length D = 11 001
    (code/test-model.py:222-230) with the AKAI checkpoint: the fused launch against the two-pass step over the whole batch bit
    for bit (outputs, hidden state, 180 MB delay buffer), scattered streams against the oracle.  Trajectories: 8000 ... 8800
    samples with wow; some streams sweep the whole range 0 ... D (taps in the carried history, k = D, the fast and the
    general path of the fused delay stage inside one tile)."""
    import sys
    import oracle
    from helpers import oracle_weights
    sys.path.insert(0, ROOT)
    import bench
    W = "DiffDelGRU-HS[64]-L[DCPreESR]-DS[ReelToReel_Dataset_MiniPulse100_AKAI_IPS[7.5]_MAXELL]_BEST"
    B, T, MD = 4096, 65536, 11000
    devc = torch.device("cuda", 0)
    x = bench.synth_input(B, T, devc, seed=4321)
    g = torch.Generator(device=devc).manual_seed(5)
    n = torch.arange(T, device=devc, dtype=torch.float32).unsqueeze(0)
    base = 8000.0 + 400.0 * torch.rand(B, 1, generator=g, device=devc)
    wow = 100.0 + 300.0 * torch.rand(B, 1, generator=g, device=devc)
    rate = 0.5 + 4.0 * torch.rand(B, 1, generator=g, device=devc)
    d = torch.empty(B, T, device=devc)
    for b0 in range(0, B, 512):
        sl = slice(b0, b0 + 512)
        d[sl] = base[sl] + wow[sl] * torch.sin(2 * np.pi * rate[sl] * n / 44100.0) + 20.0 * torch.sin(2 * np.pi * 23.0 * n / 44100.0)
    sweep = [7, 16, 2049, 4095]                       # the whole range, ends included
    for k, r in enumerate(sweep):
        d[r] = (0.5 + 0.5 * torch.sin(2 * np.pi * (0.7 + k) * n[0] / 44100.0 + k)) * (MD + 1)
    d[7, :64] = float(MD + 1)                         # d == D exactly (the w_b tap is dropped)
    d[16, 100:164] = 0.0
    d = d.clamp_(0, MD + 1).unsqueeze(1)
    res = {}
    for mode in ("auto", "two_pass"):
        m = build(ntm, W, MD)
        m.warm_cache = True
        m.delay_mode = mode
        y, p = m.predict(x, d)
        res[mode] = (y, p, m.hidden.clone(), m.diffdel.buffer.clone())
        assert m.diffdel.buffer.shape == (B, 1, MD + 1)
    for a, b in zip(res["auto"], res["two_pass"]):
        assert torch.equal(a, b)
    rows = sweep + [0, 15, 2047, 2048]
    yo, po, ho, bo = oracle.diffdel_predict(oracle_weights(W), x[rows, 0].cpu().numpy(), d[rows, 0].cpu().numpy(), MD, threads=8)
    y, p, h, buf = res["auto"]
    assert np.abs(p[rows, 0].cpu().numpy() - po).max() < TOL and np.abs(y[rows, 0].cpu().numpy() - yo).max() < TOL
    assert np.abs(buf[rows, 0].cpu().numpy() - bo).max() < TOL
    # the delay line itself is exact: the GPU's own pre_d through the oracle's delay line gives the GPU's y bit for bit
    # (the warm buffer comes from the GPU's own warm-up, not the oracle's: take it from a B = 1 run)
    m1 = build(ntm, W, MD)
    m1.initialize_hidden(1, MD)
    m1.warm_start()
    bw = m1.diffdel.buffer[0].cpu().numpy()
    yd, _ = oracle.delay_forward(p[rows, 0].cpu().numpy(), d[rows, 0].cpu().numpy(), np.repeat(bw, len(rows), 0))
    assert np.array_equal(yd, y[rows, 0].cpu().numpy())


@pytest.mark.gpu
def test_real_tape_command_line_at_the_operating_point(tmp_path, monkeypatch):
    """scripts/test-model-loss.sh:87-93 (the REAL branch) as issued for TAPE = MAXELL, IPS = 7.5, MODEL = DiffDelGRU, LOSS = DCPreESR:
    a dataset directory with brackets in its name (glob escaping, code/dataset.py:133), 10-second segments (441 000 samples),
    side-car trajectories around 0.19 s so that the delay line gets the real-tape length int(1.25 * max * fs) and INIT_LEN =
    16 384 (code/test-model.py:222-230, :323-324); three segments, losses against the oracle."""
    import oracle
    from helpers import oracle_weights
    from scipy.io import wavfile
    from test_cli import cli_module
    from ntm_amd.feeder import write_sidecar
    cli = cli_module("ntm_cli_r4_real")
    ds = "ReelToReel_Dataset_MiniPulse100_AKAI_IPS[7.5]_MAXELL"
    W = f"DiffDelGRU-HS[64]-L[DCPreESR]-DS[{ds}]_BEST"
    fs, L, nseg = 44100, 441000, 3
    N = nseg * L + 1234
    rng = np.random.default_rng(75)
    n = np.arange(N)
    audio = (0.4 * np.sin(2 * np.pi * np.cumsum(220.0 * (1 + 0.3 * np.sin(2 * np.pi * 0.2 * n / fs))) / fs)
             * (0.6 + 0.4 * np.sin(2 * np.pi * 0.5 * n / fs)) + 0.02 * rng.standard_normal(N)).astype(np.float32)
    traj = (0.190 + 0.004 * np.sin(2 * np.pi * 1.3 * n / fs) + 0.0004 * np.sin(2 * np.pi * 23 * n / fs))          # seconds
    tgt = (0.5 * np.tanh(2.0 * np.roll(audio, int(0.19 * fs)))).astype(np.float32)
    d = tmp_path / "audio" / ds / "Test"
    d.mkdir(parents=True)
    pilot = np.zeros(N, np.float32)
    wavfile.write(str(d / "input_3_.wav"), fs, np.stack([audio, pilot], 1))
    wavfile.write(str(d / "target_3_.wav"), fs, np.stack([tgt, pilot], 1))
    peaks = np.arange(1000, N - 10000, 4410)
    write_sidecar(str(d / "trajectory_3_.npy"), peaks, peaks + 8379, traj, {"reconstruction_percentage": 0.0, "wiggle_percentage": 0.0},
                  {"reconstruction_percentage": 0.0, "wiggle_percentage": 0.0})
    (tmp_path / "scripts").mkdir()
    monkeypatch.chdir(tmp_path / "scripts")
    got = cli.main(["--MODEL", "DiffDelGRU", "--WEIGHTS", W, "--DATASET", ds, "--SUBSET", "Test", "--NO_SHUFFLE", "--SEGMENT_LENGTH", str(L),
                    "--ADD_DELAY", "--COMPUTE_LOSS", "--SAVE_AUDIO", "--DESCRIPTIVE_NAME", "LOSS", "--IDX", "2", "--DELAY_TYPE", "True"])
    max_delay_n = int(1.25 * traj.max() * fs)
    init = 1 << (int(traj.max() * fs) - 1).bit_length()
    assert 10000 < max_delay_n < 11000 and init == 16384
    X = np.stack([audio[k * L:(k + 1) * L] for k in range(nseg)])
    Dt = np.stack([traj[k * L:(k + 1) * L].astype(np.float32) * np.float32(fs) for k in range(nseg)])
    Tg = np.stack([tgt[k * L:(k + 1) * L] for k in range(nseg)])
    yo, _, _, _ = oracle.diffdel_predict(oracle_weights(W), X, Dt, max_delay_n, threads=3)
    want = float(np.mean(oracle.esr_per_segment(yo, Tg, init)))
    sd = oracle.esr_dcpre_sums(yo, Tg, init)
    want_dc = float(np.mean((sd[:, 0] / (L - init)) / (sd[:, 1] / (L - init) + 1e-5)))
    assert abs(got["ESR"] - want) < 1e-4 * want and abs(got["DCPreESR"] - want_dc) < 1e-4 * want_dc, (got, want, want_dc)
    assert "MultiSTFT" in got
    fsr, pred = wavfile.read(str(tmp_path / "results" / f"{ds}_LOSS_prediction_Supervised 2.wav"))
    assert fsr == fs and len(pred) == L - init and np.abs(pred.astype(np.float64) / 32767 - yo[2, init:]).max() < 1e-5 + 0.51 / 32767


# ----------------------------------------------------------------------------- the one collective, without torch.distributed
def test_loss_scalars_and_rccl_allreduce_through_ctypes(ntm):
    """ntm_loss_scalars (this rank's four fp64 loss scalars from per-stream ESR rows, code/test-model.py:386-398) against numpy and
    against distributed.local_loss_sums; then the SUM all-reduce of include/ntm_rccl.h on a communicator of one rank (the box has
    one GPU): RCCL really initialises, reduces on the caller's stream and tears down; bit-reproducible from call to call."""
    import ctypes
    L = ntm._lib.lib()
    R = ctypes.CDLL(os.path.join(os.path.dirname(ntm._lib.LIB_PATH), "libntm_rccl.so"))
    R.ntm_rccl_last_error.restype = ctypes.c_char_p
    rng = np.random.default_rng(12)
    for B, n in ((1, 1000), (300, 64512), (5000, 441000 - 16384), (32768, 64512)):
        rows = np.stack([rng.uniform(0.1, 5.0, B) * n * 1e-3, rng.uniform(0.5, 2.0, B) * n * 1e-2], 1)
        dr = dev(rows)
        out = torch.empty(4, dtype=torch.float64, device="cuda")
        assert L.ntm_loss_scalars(ctypes.c_void_p(dr.data_ptr()), B, n, 1e-5, ctypes.c_void_p(out.data_ptr()), ntm._lib.current_stream()) == 0
        got = out.cpu().numpy()
        per = (rows[:, 0] / n) / (rows[:, 1] / n + 1e-5)
        want = np.array([per.sum(), B, rows[:, 0].sum(), rows[:, 1].sum()])
        assert np.abs(got / want - 1).max() < 1e-12
        tl = ntm.distributed.local_loss_sums(dev(per), dr).cpu().numpy()
        assert np.abs(got / tl - 1).max() < 1e-12
        out2 = torch.empty_like(out)
        L.ntm_loss_scalars(ctypes.c_void_p(dr.data_ptr()), B, n, 1e-5, ctypes.c_void_p(out2.data_ptr()), ntm._lib.current_stream())
        assert torch.equal(out, out2)
    zero = torch.full((4,), 7.0, dtype=torch.float64, device="cuda")
    assert L.ntm_loss_scalars(None, 0, 10, 1e-5, ctypes.c_void_p(zero.data_ptr()), ntm._lib.current_stream()) == 0
    assert zero.cpu().tolist() == [0.0, 0.0, 0.0, 0.0]
    idb = (ctypes.c_ubyte * 128)()
    comm = ctypes.c_void_p()
    assert R.ntm_rccl_unique_id(idb) == 0, R.ntm_rccl_last_error()
    assert R.ntm_rccl_comm_create(ctypes.byref(comm), 1, 0, idb) == 0, R.ntm_rccl_last_error()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        v = out.clone()
        assert R.ntm_rccl_allreduce_f64(ctypes.c_void_p(v.data_ptr()), ctypes.c_int64(4), comm, ctypes.c_void_p(side.cuda_stream)) == 0, R.ntm_rccl_last_error()
    side.synchronize()
    assert torch.equal(v, out)                                   # one rank: the sum is the input
    assert R.ntm_rccl_comm_destroy(comm) == 0


def test_sharded_evaluation_from_a_plain_cpp_process(tmp_path):
    """tools/cabi/cabi_demo `reduce` mode: what ONE rank of a torch-free multi-GPU evaluation runs -- ntm_gru_forward_esr (forward +
    ESR sums in one call), ntm_loss_scalars, the RCCL all-reduce of the four scalars (a communicator of one rank here) -- against
    the oracle: job ESR = mean over segments of the per-segment ESR (code/test-model.py:386-398)."""
    import subprocess
    import oracle
    from helpers import oracle_weights
    exe = os.path.join(ROOT, "tools", "cabi", "cabi_demo.bin")
    if not os.path.exists(exe):
        subprocess.run(["make", "-s", "-C", os.path.dirname(exe)], check=True)
    W_G = "GRU-HS[64]-L[DCPreESR]-DS[ReelToReel_Dataset_MiniPulse100_CHOWTAPE]_BEST"
    rng = np.random.default_rng(31)
    B, T, skip = 1045, 900, 64                      # 1045 > 1024: the matrix-pipe kernel, the sums ride in its launch
    x = rng.uniform(-0.5, 0.5, (B, T)).astype(np.float32)
    t = (0.4 * np.tanh(2 * x) + 0.01 * rng.standard_normal((B, T))).astype(np.float32)
    x.tofile(str(tmp_path / "x.f32")); t.tofile(str(tmp_path / "t.f32"))
    wfile = os.path.join(ROOT, "neural-tape-modeling_amd", "weights", "w0.bin")
    r = subprocess.run([exe, "reduce", wfile, str(tmp_path / "x.f32"), str(tmp_path / "t.f32"), str(B), str(T), str(skip)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr + r.stdout
    vals = {kv.split("=")[0]: float(kv.split("=")[1]) for kv in r.stdout.split() if "=" in kv}
    yo, _ = oracle.gru_forward(oracle_weights(W_G), x, threads=4)               # h_0 = 0, as the demo passes it
    so = oracle.esr_sums(yo, t, skip)
    per = (so[:, 0] / (T - skip)) / (so[:, 1] / (T - skip) + 1e-5)
    assert vals["ranks"] == 1 and vals["segments"] == B
    assert abs(vals["sum_tgt2"] / so[:, 1].sum() - 1) < 1e-12 and abs(vals["sum_err2"] / so[:, 0].sum() - 1) < 1e-4
    assert abs(vals["job_esr"] / per.mean() - 1) < 1e-4 and abs(vals["sum_esr"] / per.sum() - 1) < 1e-4
