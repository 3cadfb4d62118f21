"""GPU parity, round 2: BASELINE's full-size configurations with 4096 DISTINCT streams (every stream of the batch has
its own input; scattered streams are compared with the C oracle over all 65 536 samples, carried state included), the
small hidden sizes of the reference's own defaults (golden g16), the `skip` connection, the one-pass delay line's
deferred range check, and bench.py's own multi-rank launcher.  Tolerance for the GRU/TCN paths: 1e-5 abs fp32
(BASELINE.json north_star); the delay line is bit-exact."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import oracle
from helpers import ROOT, bench_record, load, oracle_weights

pytestmark = pytest.mark.gpu

TOL = 1e-5
W_G = "GRU-HS[64]-L[DCPreESR]-DS[ReelToReel_Dataset_MiniPulse100_CHOWTAPE]_BEST"
W_D = "DiffDelGRU-HS[64]-L[DCPreESR]-DS[ReelToReel_Dataset_MiniPulse100_CHOWTAPE_WOWFLUTTER]_BEST"
FULL = pytest.mark.skipif(os.environ.get("NTM_SKIP_FULL") == "1", reason="full 4096x65536 pass skipped on request")
THREADS = max(1, min(32, len(os.sched_getaffinity(0))))


@pytest.fixture(scope="module")
def ntm():
    import ntm_amd
    assert torch.cuda.is_available(), "these tests need the MI355X"
    ntm_amd._lib.lib()       # raises if libntm.so is missing: no silent fallback
    return ntm_amd


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def scattered_rows(B, n_random=24, seed=5):
    """Group boundaries of the 16-stream workgroups, first/last rows, and seeded random rows."""
    fixed = [0, 1, 15, 16, 17, 255, 256, 2047, 2048, B - 17, B - 16, B - 1]
    rnd = np.random.default_rng(seed).choice(B, n_random, replace=False).tolist()
    return sorted(set(int(r) for r in fixed + rnd if 0 <= r < B))


def distinct_input(B, T):
    sys.path.insert(0, ROOT)
    import bench
    return bench.synth_input(B, T, torch.device("cuda", 0), seed=1234)     # the headline workload's own generator


# ----------------------------------------------------------------------------- configs[1]
@FULL
def test_full_size_cfg2_4096_distinct_streams(ntm):
    """BASELINE configs[1] exactly as bench.py runs it: 4096 distinct streams x 65 536 samples.  36 scattered streams
    (workgroup boundaries, first / last, random) against the oracle over ALL samples, and their carried hidden state;
    a row-addressing slip (stream b served from row b +- k) cannot hide here as it could behind replicated inputs."""
    B, T = 4096, 65536
    x = distinct_input(B, T)
    m = ntm.harness.build_model(W_G)
    y = m.predict(x)
    rows = scattered_rows(B)
    assert len(rows) >= 32
    yo, ho = oracle.gru_predict(oracle_weights(W_G), x[rows, 0].cpu().numpy(), threads=THREADS)
    assert np.abs(y[rows, 0].cpu().numpy() - yo).max() < TOL
    assert np.abs(m.hidden[0, rows].cpu().numpy() - ho).max() < TOL
    # the streams really are distinct, and so are the outputs
    assert len({float(v) for v in y[rows, 0, -1].cpu()}) == len(rows)
    # chunked == one-shot, bit for bit, on the distinct batch (state carried across launches)
    m.initialize_hidden(); m.warm_start(); m.hidden = m.hidden.expand(1, B, 64).contiguous()
    cut = 30000 + 7
    ya, yb = m(x[:, :, :cut]), m(x[:, :, cut:])
    assert torch.equal(ya, y[:, :, :cut]) and torch.equal(yb, y[:, :, cut:])


# ----------------------------------------------------------------------------- configs[2]
def wow_trajectories(B, T, max_delay, seed=77):
    dev0 = torch.device("cuda", 0)
    g = torch.Generator(device=dev0); g.manual_seed(seed)
    amp = 0.002 + 0.003 * torch.rand(B, 1, generator=g, device=dev0)
    wv = 0.5 + 1.5 * torch.rand(B, 1, generator=g, device=dev0)
    psi = 2 * np.pi * torch.rand(B, 1, generator=g, device=dev0)
    n = torch.arange(T, device=dev0, dtype=torch.float32).unsqueeze(0)
    d = torch.empty(B, T, device=dev0)
    for b0 in range(0, B, 256):
        sl = slice(b0, min(B, b0 + 256))
        d[sl] = 44100 * (0.0271 + amp[sl] * torch.sin(2 * np.pi * wv[sl] * n / 44100 + psi[sl])
                         + 0.0005 * torch.sin(2 * np.pi * 23 * n / 44100))
    return d.clamp_(0, max_delay).unsqueeze(1)


@FULL
def test_full_size_cfg3_4096_distinct_streams(ntm):
    """BASELINE configs[2]: DiffDelGRU, 4096 distinct (signal, wow trajectory) pairs x 65 536, D = 1847.  Scattered
    streams against the oracle over all samples: pre_d within 1e-5; y within 1e-5 AND bit-identical to the oracle's
    delay line applied to the GPU's own pre_d (the delay line itself is exact); both carried states."""
    B, T = 4096, 65536
    x = distinct_input(B, T)
    m = ntm.harness.build_model(W_D, max_delay_seconds=0.0335)
    d = wow_trajectories(B, T, m.max_delay)
    y, pre = m.predict(x, d)
    rows = scattered_rows(B)
    xs, ds = x[rows, 0].cpu().numpy(), d[rows, 0].cpu().numpy()
    yo, preo, ho, bo = oracle.diffdel_predict(oracle_weights(W_D), xs, ds, m.max_delay, threads=THREADS)
    pre_g, y_g = pre[rows, 0].cpu().numpy(), y[rows, 0].cpu().numpy()
    assert np.abs(pre_g - preo).max() < TOL and np.abs(y_g - yo).max() < TOL
    assert np.abs(m.hidden[0, rows].cpu().numpy() - ho).max() < TOL
    assert np.abs(m.diffdel.buffer[rows, 0].cpu().numpy() - bo).max() < TOL
    # exactness of K2 in isolation: the oracle's delay line on the GPU's own pre_d, starting from the GPU's own
    # warm-start buffer (predict() builds it the same, deterministic way), must give the GPU's y bit for bit
    buf_end = m.diffdel.buffer[rows, 0].cpu().numpy()
    m.initialize_hidden(1, m.max_delay)
    m.warm_start()
    b1 = m.diffdel.buffer[:, 0].cpu().numpy()
    y_exact, b_exact = oracle.delay_forward(pre_g, ds, np.repeat(b1, len(rows), 0))
    assert np.array_equal(y_g, y_exact)
    assert np.array_equal(buf_end, b_exact)


def test_diffdel_error_budget_where_the_3e6_comes_from(ntm):
    """Round 1 measured 3.1e-6 on a DiffDelGRU stream against 8e-7 for the GRU: the delay line is exact (see above), the
    interpolation weights are <= 1 and sum to <= 1, so |y - y_oracle| can never exceed the largest |pre_d - pre_d_oracle|:
    the difference comes from the GRU with these weights (the WOWFLUTTER checkpoint has the larger recurrent gain),
    not from the fp32 `d` arithmetic."""
    T = 65536
    rng = np.random.default_rng(3)
    x = rng.uniform(-0.5, 0.5, (4, T)).astype(np.float32)
    n = np.arange(T)
    d = (44100 * (0.0271 + 0.004 * np.sin(2 * np.pi * 1.3 * n / 44100))).astype(np.float32)[None].repeat(4, 0)
    m = ntm.harness.build_model(W_D, max_delay_seconds=0.0335)
    y, pre = m.predict(dev(x).unsqueeze(1), dev(d).unsqueeze(1))
    yo, preo, _, _ = oracle.diffdel_predict(oracle_weights(W_D), x, d, m.max_delay, threads=4)
    e_pre = np.abs(pre[:, 0].cpu().numpy() - preo).max()
    e_y = np.abs(y[:, 0].cpu().numpy() - yo).max()
    assert e_y <= e_pre * (1 + 1e-6) + 1e-9 and e_pre < TOL


# ----------------------------------------------------------------------------- configs[3]
@FULL
def test_full_size_cfg4_tcn_4096_distinct_streams(ntm):
    """BASELINE configs[3]: TCN, 4096 distinct streams x 65 536; 10 scattered streams against the oracle at 1e-5."""
    B, T = 4096, 65536
    x = distinct_input(B, T)
    m = ntm.TCN().to("cuda")
    y = m(x)
    rows = [0, 1, 15, 16, 2047, 2048, 4094, 4095] + np.random.default_rng(9).choice(B, 4, replace=False).tolist()
    rows = sorted(set(int(r) for r in rows))
    yo = oracle.tcn_forward(m.packed_params().cpu().numpy(), len(m.dilations), m.channels, m.kernel_size, m.dilations,
                            x[rows, 0].cpu().numpy(), threads=THREADS)
    assert np.abs(y[rows, 0].cpu().numpy() - yo).max() < TOL
    del y
    torch.cuda.empty_cache()


# ----------------------------------------------------------------------------- hidden sizes 8 / 16 / 32 (golden g16)
def _load_g16(ntm, g, prefix, H, cls, **kw):
    m = cls(1, H, 1, **kw) if H != 8 or kw else cls()            # H = 8: the reference's all-default constructor
    sd = {k[len(prefix):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(prefix)}
    m.load_state_dict(sd)                                         # a reference-format state_dict, strict
    return m.to("cuda").eval()


@pytest.mark.parametrize("variant", ["auto", "lat", "valu"])
@pytest.mark.parametrize("H", [8, 16, 32])
def test_g16_hidden_sizes_vs_reference(ntm, H, variant):
    g = load("g16_hidden_sizes.npz")
    m = _load_g16(ntm, g, f"sd_{H}_", H, ntm.RNN)
    m.kernel_variant = variant
    assert m.hidden_size == H
    x = dev(g[f"x_{H}"])
    y = m.predict(x)                                              # batched predict == the reference's per-stream predict
    assert np.abs(y.cpu().numpy() - g[f"y_{H}_predict"]).max() < TOL
    for b in range(3):                                            # and B = 1, literally the reference's call
        assert np.abs(m.predict(x[b:b + 1]).cpu().numpy() - g[f"y_{H}_predict"][b:b + 1]).max() < TOL
    m.initialize_hidden()
    y0, y1 = m(x[:, :, :700]), m(x[:, :, 700:])
    assert np.abs(torch.cat([y0, y1], 2).cpu().numpy() - g[f"y_{H}_carry"]).max() < TOL
    assert np.abs(m.hidden.cpu().numpy() - g[f"h_{H}_carry"]).max() < TOL
    # chunked == one-shot bit for bit (reference chunk length 2048 and an odd one)
    assert torch.equal(m.predict(x, segment_length=2048), y) and torch.equal(m.predict(x, segment_length=333), y)


@pytest.mark.parametrize("H", [8, 16, 32])
@pytest.mark.parametrize("B,T", [(1, 1), (7, 63), (9, 64), (64, 129), (130, 700), (4099, 200)])
def test_small_hidden_sizes_ragged_vs_oracle(ntm, H, B, T):
    """Ragged batches (B not a multiple of the 64/H streams of a wavefront, T around the 64-sample tile)."""
    rng = np.random.default_rng(100 * H + B + T)
    torch.manual_seed(H)
    m = ntm.RNN(1, H, 1).to("cuda").eval()
    w = oracle.Weights.from_state_dict({k: v.cpu().numpy() for k, v in m.state_dict().items()})
    x = rng.uniform(-0.7, 0.7, (B, T)).astype(np.float32)
    h0 = rng.uniform(-0.9, 0.9, (B, H)).astype(np.float32)
    m.hidden = dev(h0).view(1, B, H)
    y = m(dev(x).unsqueeze(1))
    yo, ho = oracle.gru_forward(w, x, h0, threads=8)
    assert np.abs(y[:, 0].cpu().numpy() - yo).max() < TOL and np.abs(m.hidden[0].cpu().numpy() - ho).max() < TOL


def test_g16_diffdel_hidden_16_and_skip(ntm):
    g = load("g16_hidden_sizes.npz")
    md = int(g["dd_max_delay"])
    m = _load_g16(ntm, g, "dd_sd_", 16, ntm.DiffDelRNN, max_delay=md)
    y, pre = m.predict(dev(g["dd_x"]), dev(g["dd_d"]))
    assert np.abs(pre.cpu().numpy() - g["dd_pre_d"]).max() < TOL and np.abs(y.cpu().numpy() - g["dd_y"]).max() < TOL
    assert np.abs(m.diffdel.buffer.cpu().numpy() - g["dd_buffer"]).max() < TOL
    assert np.abs(m.hidden.cpu().numpy() - g["dd_hidden"]).max() < TOL
    m.skip = True                                                 # code/model.py:409-415
    ys, pres = m.predict(dev(g["dd_x"]), dev(g["dd_d"]))
    assert np.abs(pres.cpu().numpy() - g["dd_pre_d_skip"]).max() < TOL and np.abs(ys.cpu().numpy() - g["dd_y_skip"]).max() < TOL
    r = _load_g16(ntm, g, "sd_16_", 16, ntm.RNN)
    r.skip = True                                                 # code/model.py:79-84
    assert np.abs(r.predict(dev(g["x_16"][:1])).cpu().numpy() - g["y_16_skip"]).max() < TOL


@pytest.mark.parametrize("variant", ["mfma2", "lat"])
def test_skip_connection_hs64(ntm, variant):
    """skip=True with the shipped HS[64] weights: y = GRU+head output + x (the reference's scripts always pass
    skip=False, code/test-model.py:126, but the constructor argument is part of the protocol), also through
    forward_into and the chunked predict."""
    rng = np.random.default_rng(64)
    x = rng.uniform(-0.5, 0.5, (40, 1500)).astype(np.float32)
    m = ntm.RNN(1, 64, 1, skip=True)
    m.load_state_dict(ntm.weights.load_state_dict(W_G))
    m = m.to("cuda").eval()
    m.kernel_variant = variant
    y = m.predict(dev(x).unsqueeze(1))
    yo, _ = oracle.gru_predict(oracle_weights(W_G), x, threads=4)
    assert np.abs(y[:, 0].cpu().numpy() - (yo + x)).max() < TOL
    assert torch.equal(m.predict(dev(x).unsqueeze(1), segment_length=2048), y)
    m.initialize_hidden(); m.warm_start(); m.hidden = m.hidden.expand(1, 40, 64).contiguous()
    out = torch.empty(40, 1500, device="cuda")
    m.forward_into(dev(x), out)
    assert torch.equal(out, y[:, 0])


# ----------------------------------------------------------------------------- K2: one pass, deferred range check
def test_delay_line_sticky_flag_and_deferred_check(ntm):
    """The range check rides in the pass that reads d.  Deferred mode: a violating chunk leaves the buffer at the
    state before it, every later chunk is a no-op, and raise_if_violated() reports it once -- no host sync per chunk.
    Default mode raises AssertionError at the violating call like code/model.py:284.  NaN counts as a violation
    (`max_delay >= max(dt)` is False for NaN)."""
    rng = np.random.default_rng(21)
    B, D, C = 6, 300, 256
    x = rng.standard_normal((B, 5 * C)).astype(np.float32)
    d = rng.uniform(0, D, (B, 5 * C)).astype(np.float32)
    dl = ntm.TimeVaryingDelayLine(max_delay=D)
    dl.init_buffer(B, D)
    dl.defer_check = True
    xs, ds = dev(x).unsqueeze(1), dev(d).unsqueeze(1)
    y0 = dl(xs[:, :, :C], ds[:, :, :C])
    y1 = dl(xs[:, :, C:2 * C], ds[:, :, C:2 * C])
    good = dl.buffer.clone()
    bad = ds[:, :, 2 * C:3 * C].clone()
    bad[4, 0, 77] = D + 0.25
    dl(xs[:, :, 2 * C:3 * C], bad)                                # violates: state frozen
    dl(xs[:, :, 3 * C:4 * C], ds[:, :, 3 * C:4 * C])              # later good chunk: still frozen (sticky flag)
    assert torch.equal(dl.buffer, good)
    with pytest.raises(AssertionError):
        dl.raise_if_violated()
    dl.raise_if_violated()                                        # cleared by the raise
    y2 = dl(xs[:, :, 2 * C:3 * C], ds[:, :, 2 * C:3 * C])         # resume from the frozen state: as if nothing happened
    yo, bo = oracle.delay_forward(x[:, :3 * C], d[:, :3 * C], np.zeros((B, D), np.float32))
    assert np.array_equal(torch.cat([y0, y1, y2], 2)[:, 0].cpu().numpy(), yo)
    assert np.array_equal(dl.buffer[:, 0].cpu().numpy(), bo)
    dl.raise_if_violated()
    # default mode + NaN
    dl.defer_check = False
    before = dl.buffer.clone()
    nan = ds[:, :, :C].clone()
    nan[0, 0, 3] = float("nan")
    with pytest.raises(AssertionError):
        dl(xs[:, :, :C], nan)
    assert torch.equal(dl.buffer, before)
    with pytest.raises(AssertionError):                           # also in warm-up mode (the assert precedes the branch)
        dl(xs[:, :, :C], bad, warmup=True)
    assert torch.equal(dl.buffer, before)


def test_diffdel_predict_checks_range_once_at_the_end(ntm):
    m = ntm.harness.build_model(W_D, max_delay_seconds=0.005)     # D = 276 + 1
    T = 5000
    x = torch.zeros(3, 1, T, device="cuda")
    d = torch.full((3, 1, T), 100.0, device="cuda")
    m.predict(x, d, segment_length=2048)                          # fine
    d[1, 0, 4100] = m.diffdel.max_delay + 1.0                     # violation in the LAST chunk
    with pytest.raises(AssertionError):
        m.predict(x, d, segment_length=2048)
    with pytest.raises(AssertionError):
        m.predict(x, d)
    y, _ = m.predict(x, d.clamp(max=float(m.diffdel.max_delay)))  # and the model is usable afterwards
    assert torch.isfinite(y).all()


@pytest.mark.parametrize("T,D", [(8192, 1847), (4096, 8900), (1000, 37), (2048 + 5, 300)])
def test_delay_fast_path_and_general_path_agree_with_oracle(ntm, T, D):
    """Slowly varying delays take the kernel's windowed fast path (one integer part per 8-sample run), fast-changing /
    boundary delays the general path; both must be bit-identical to the oracle, aligned and unaligned T."""
    rng = np.random.default_rng(T + D)
    B = 5
    n = np.arange(T)
    x = rng.standard_normal((B, T)).astype(np.float32)
    d = np.empty((B, T), np.float32)
    d[0] = 0.6 * D + 0.3 * D * np.sin(2 * np.pi * n / 3000.0)           # slow wow: fast path nearly everywhere
    d[1] = rng.uniform(0, D, T)                                          # white: general path
    d[2] = np.floor(0.5 * D) + (n % 2) * 0.999                           # integer part constant, fraction toggles
    d[3] = np.clip(n.astype(np.float64) * D / T, 0, D)                   # ramp through every integer part, ends at D
    d[3, -1] = D
    d[4] = np.where(n % 97 == 0, -0.5, 0.25 * D)                         # negative delays sprinkled in
    buf0 = rng.standard_normal((B, D)).astype(np.float32)
    dl = ntm.TimeVaryingDelayLine(max_delay=D)
    dl.init_buffer(B, D)
    dl.buffer = dev(buf0).view(B, 1, D)
    y = dl(dev(x).unsqueeze(1), dev(d).unsqueeze(1))
    yo, bo = oracle.delay_forward(x, d, buf0)
    assert np.array_equal(y[:, 0].cpu().numpy(), yo) and np.array_equal(dl.buffer[:, 0].cpu().numpy(), bo)


# ----------------------------------------------------------------------------- bench.py launches its own ranks
def test_bench_spawns_its_own_ranks_on_one_gpu_over_gloo():
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE: the parent spawns two ranks (NTM_DIST_BACKEND=gloo
    lets them share this box's single GPU), relays rank 0's JSON line and exits 0.  Whole-job value = both ranks'
    segments over the max-over-ranks time; the per-step loss scalars are reduced over both ranks."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["NTM_DIST_BACKEND"] = "gloo"
    for extra, total in ((["--batch", "512"], 1024), (["--scaling", "strong", "--total-batch", "768"], 768)):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                            "--samples", "4096", "--no-cpu-baseline", "--no-extra"] + extra,
                           env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        line, out = bench_record(r.stdout)                  # one compact line (< 4 KB) + the detail file it names
        assert line["n_gpus"] == 2 and line["backend"] == "gloo" and line["rccl_ranks"] == 0 and "roofline" in line
        assert out["n_gpus"] == 2 and out["ranks"] == 2 and out["backend"] == "gloo" and out["rccl_ranks"] == 0
        assert [d["rank"] for d in out["rank_devices"]] == [0, 1]
        assert out["config"]["segments_total"] == total and out["checks"]["segments"] == total
        assert out["checks"]["job_esr"] > 0 and out["checks"]["last_output_equals_first_pass_bitwise"] is True
        assert out["scaling"] == ("strong" if "--scaling" in extra else "weak")
        assert abs(out["value"] - total * 4096 * 2 / (out["ms_per_step"] * 2e-3)) < 1e-6 * out["value"]


def test_bench_one_rank_process_group_runs_the_rccl_calls():
    """The collectives of the N-rank bench path on the REAL backend: with NTM_DIST_FORCE_INIT=1 a single rank still
    builds the process group (backend nccl = RCCL), all-gathers the rank/device table, all-reduces the K x 4 fp64 loss
    scalars and the elapsed-time MAX, passes the barriers and destroys the group -- everything an 8-GPU launch does
    except moving bytes over xGMI.  Same line as a plain one-process run otherwise."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k != "NTM_DIST_BACKEND"}
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), NTM_DIST_FORCE_INIT="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--batch", "512", "--samples", "4096", "--no-cpu-baseline", "--no-extra"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line, out = bench_record(r.stdout)
    # the compact line of the nccl code path: the first 8-GPU run must not be the first time it prints one
    assert line["backend"].startswith("rccl") and line["rccl_ranks"] == 1 and line["ranks"] == 1 and line["n_gpus"] == 1
    assert line["roofline"]["frac"] > 0 and line["checks"]["deterministic"] is True and line["checks"]["segments"] == 512
    assert out["backend"].startswith("rccl") and out["rccl_ranks"] == 1 and out["ranks"] == 1
    assert out["rank_devices"][0]["rank"] == 0 and out["rank_devices"][0]["device"] == 0
    assert out["checks"]["segments"] == 512 and out["checks"]["job_esr"] > 0 and out["checks"]["last_output_equals_first_pass_bitwise"] is True
    assert out["checks"]["every_timed_step_same_loss"] is True


def test_tcn_refuses_dilations_outside_its_32_bit_sample_arithmetic(ntm):
    """The block kernels index samples with 32-bit arithmetic: dilations outside [1, 2^20] are refused with a message
    (NTM_EINVAL through the C ABI), nothing is launched."""
    x = torch.zeros(1, 1, 64, device="cuda")
    for dil in ((1, 0), (1, (1 << 20) + 1)):
        m = ntm.TCN(dilations=dil).to("cuda")
        with pytest.raises(ntm._lib.NtmError, match="dilations"):
            m(x)


# ----------------------------------------------------------------------------- feeder, demodulated items (golden g15 c)
def test_g15_feeder_demodulated_items_equal_reference(ntm, tmp_path):
    """VADataset(demodulate=True).__getitem__ (code/dataset.py:395-408: demodulate the 2-channel target with the
    segment's pulse indices, cut mean_delay from input / target / trajectory, restrict the pulses) reproduced by the
    feeder with ntm_demodulate on the device: bit-identical audio, identical pulse indices and meta strings."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_feeder import rebuild_g15_tree
    g = load("g15_vadataset.npz")
    ds = rebuild_g15_tree(g, str(tmp_path))
    L = int(g["c_length"])
    f = ntm.feeder.SegmentFeeder(ds, subset="test", length=L, sync=float(g["c_sync"]), demodulate=True)
    assert len(f) == int(g["c_n"]) and [[e["idx"], e["offset"]] for e in f.examples] == g["c_examples"].tolist()
    checked = 0
    for i in range(len(f)):
        x, t, meta = f[i]
        nin, ntg, ntr, sin_, stg, str_ = g[f"c_{i}_shape_sums"]
        assert (x.shape[-1], t.shape[-1], len(meta["delay_trajectory"])) == (nin, ntg, ntr)
        assert [meta["input_name"], meta["target_name"]] == list(g[f"c_{i}_names"])
        assert np.array_equal(np.asarray(meta["input_peaks"]), g[f"c_{i}_input_peaks"])
        assert np.array_equal(np.asarray(meta["output_peaks"]), g[f"c_{i}_output_peaks"])
        assert abs(float(t.double().sum()) - stg) < 1e-6 * max(1.0, abs(stg))
        if f"c_{i}_target" in g.files:
            assert np.array_equal(x.numpy(), g[f"c_{i}_input"])
            assert np.array_equal(t.numpy(), g[f"c_{i}_target"].astype(np.float32))
            assert np.abs(meta["delay_trajectory"].numpy() - g[f"c_{i}_traj"]).max() < 1e-8
            checked += 1
    assert checked == 3


# ----------------------------------------------------------------------------- sharded CLI, fewer segments than ranks
def test_cli_two_ranks_one_segment_does_not_hang(tmp_path):
    """tools/test_model.py over 2 ranks (gloo, sharing this box's GPU) on a dataset with ONE segment: rank 1's shard is
    empty.  Every rank must issue the same collectives whatever its shard holds (round 1 decided the MultiSTFT key per
    rank: the empty rank would have issued 2 all-reduces against 3 and RCCL would have blocked).  Both ranks finish and
    rank 0 reports the single segment's losses, equal to a one-process run."""
    import socket
    from scipy.io import wavfile
    d = tmp_path / "One" / "Test"
    d.mkdir(parents=True)
    rng = np.random.default_rng(5)
    x = rng.uniform(-0.4, 0.4, 6000).astype(np.float32)
    wavfile.write(str(d / "input_1_.wav"), 44100, x)
    wavfile.write(str(d / "target_1_.wav"), 44100, (0.5 * x).astype(np.float32))
    args = [sys.executable, os.path.join(ROOT, "tools", "test_model.py"), "--DATASET_DIR", str(tmp_path / "One"), "--SUBSET", "Test", "--NO_SHUFFLE", "--WEIGHTS", W_G,
            "--COMPUTE_LOSS"]
    base = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    one = subprocess.run(args + ["--TEMP_PATH", str(tmp_path / "t1")], env=base, capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert one.returncode == 0, one.stderr[-2000:]
    args += ["--TEMP_PATH", str(tmp_path / "t2")]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(base, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   NTM_DIST_BACKEND="gloo")
        procs.append(subprocess.Popen(args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=str(tmp_path)))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    stats = lambda text: [ln for ln in text.splitlines() if ":" in ln and ln.split(":")[0].strip() in ("Segments", "ESR", "DCPreESR", "MultiSTFT")]  # noqa: E731
    assert stats(outs[0][0]) == stats(one.stdout) and len(stats(one.stdout)) == 4 and "Segments:   1" in one.stdout
    assert "Stats" not in outs[1][0]


# ----------------------------------------------------------------------------- the rest of Tape.__call__
def test_tape_resamplers_playback_filter_and_whole_chain(ntm):
    """ntm_amd.Tape = the reference's Tape.__call__ (code/tape.py:389-464) on the device.  The sinc resamplers and the
    lfilter playback loss follow torchaudio's published algorithms (torchaudio is absent: parity unpinned) and are
    checked against the oracle's independent restatement; the magnetisation against the oracle pinned by golden g9;
    the chain is stateful over two calls (bias phase, magnetisation, the downsampler's left context)."""
    rng = np.random.default_rng(17)
    B, N1, N2 = 3, 1500, 700
    V = 0.5 * rng.standard_normal((B, N1 + N2))
    tp = ntm.Tape(batch_size=B, startup_enable=False, playback_loss_enable=True)
    # stage by stage
    x = dev(V[:, :400])
    up = tp.oversample(x)
    assert up.shape == (B, 6400) and np.abs(up.cpu().numpy() - oracle.sinc_resample(V[:, :400], 48000, 768000)).max() < 1e-12
    tp.M_OS, tp.M = up[:, :1600].clone(), torch.zeros(B, 100, dtype=torch.float64, device="cuda")
    dn = tp.downsample(up[:, 1600:])
    want = oracle.sinc_resample(up.cpu().numpy(), 768000, 48000)[:, 100:]
    assert dn.shape == (B, 300) and np.abs(dn.cpu().numpy() - want).max() < 1e-12
    m = dev(rng.standard_normal((B, 500)) * 2e5)
    g_play = tp.PLAY_N * tp.PLAY_W * tp.PLAY_E * tp.TAPE_V * tp.PLAY_MU0 * tp.PLAY_G
    assert np.abs(tp.H_play(m).cpu().numpy() - oracle.fir_clamp(g_play * m.cpu().numpy(), tp.b.cpu().numpy())).max() < 1e-12
    # the whole chain, two stateful calls, against the oracle's composition of the same stages
    tp = ntm.Tape(batch_size=B, startup_enable=False, playback_loss_enable=False)
    ref = ntm.TapeMagnetization(batch_size=B)            # host-side bias waveform generator (pinned by golden g14)
    outs, state, m_os_prev, m_prev = [], None, np.zeros((B, 0)), np.zeros((B, 0))
    for sl in (slice(0, N1), slice(N1, N1 + N2)):
        out = tp(dev(V[:, sl]))
        I_os = oracle.sinc_resample(tp.signal_amplitude * V[:, sl], 48000, 768000)
        H = (10.0 * (I_os + ref.bias_signal(I_os.shape[1])[None, :])) / 6e-6
        M_os, state = oracle.tape_hmag(H, state, tp.Ts_OS)
        M = oracle.sinc_resample(np.concatenate([m_os_prev, M_os], 1), 768000, 48000)[:, m_prev.shape[1]:]
        m_os_prev, m_prev = M_os, M
        want = tp.POST_GAIN * (g_play * M)
        assert out.shape == (B, sl.stop - sl.start)
        assert np.abs(out.cpu().numpy() - want).max() < 1e-6 * tp.POST_GAIN * g_play * tp.TAPE_Ms
        outs.append(out)
    # start-up (10 ms of silence through the chain, code/tape.py:376-386) and a batch smaller than batch_size
    tp2 = ntm.Tape(batch_size=4)
    y = tp2(dev(V[:2, :300]))
    assert y.shape == (2, 300) and torch.isfinite(y).all() and tp2.FLAG_STARTUP is False
