"""GPU parity, round 3: the reference's own validate() methods (golden g18), the cached warm-start state, the
`ntm_diffdel_gru_forward` entry point called through raw ctypes and from the torch-free C++ caller, eight ranks sharing
this box's GPU over gloo, and the `other_workloads` of the default bench line.  Tolerance for the GRU path: 1e-5 abs
fp32 (BASELINE.json north_star); the delay line is bit-exact on the pre-delay signal it is given."""
import ctypes
import json
import os
import subprocess
import sys
import types

import numpy as np
import pytest
import torch

import oracle
from helpers import ROOT, bench_record, load, oracle_weights, validate_batches

pytestmark = pytest.mark.gpu

TOL = 1e-5
W_G = "GRU-HS[64]-L[DCPreESR]-DS[ReelToReel_Dataset_MiniPulse100_CHOWTAPE]_BEST"
W_G_ESR = "GRU-HS[64]-L[ESR]-DS[ReelToReel_Dataset_MiniPulse100_CHOWTAPE]_BEST"
W_D = "DiffDelGRU-HS[64]-L[DCPreESR]-DS[ReelToReel_Dataset_MiniPulse100_CHOWTAPE_WOWFLUTTER]_BEST"


@pytest.fixture(scope="module")
def ntm():
    import ntm_amd
    assert torch.cuda.is_available(), "these tests need the MI355X"
    ntm_amd._lib.lib()       # raises if libntm.so is missing: no silent fallback
    return ntm_amd


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def p(t):
    return ctypes.c_void_p(t.data_ptr())


# ----------------------------------------------------------------------------- validate() (golden g18)
class _Loader:
    """What RNN.validate / DiffDelRNN.validate use of a dataloader (tools/make_goldens_validate.py builds the same)."""

    def __init__(self, batches, with_meta, max_delay_s, fs):
        self.batches, self.with_meta = batches, with_meta
        self.dataset = types.SimpleNamespace(delay_analyzer=types.SimpleNamespace(max_delay=max_delay_s), fs=fs)

    def __len__(self):
        return len(self.batches)

    def __iter__(self):
        for x, t, d in self.batches:
            yield torch.from_numpy(x), torch.from_numpy(t), ({"delay_trajectory": torch.from_numpy(d)} if self.with_meta else {})


def _esr(pred, target):
    return ((target - pred) ** 2).mean() / ((target ** 2).mean() + 1e-5)


def test_g18_validate_matches_the_reference(ntm):
    """`val_loss, examples = model.validate(dataloader_val, loss_fcn)` (code/train.py:242) against what the reference's
    own methods returned on the same batches (code/model.py:163-216, :513-616): loss, the examples' predictions, the
    carried state; host-CPU batches go to the model's device as in the reference."""
    g = load("g18_validate.npz")
    fs = int(g["fs"])
    batches = validate_batches(int(g["seed"]), int(g["n_batches"]), int(g["B"]), int(g["T"]), fs)
    m = ntm.harness.build_model(W_G)
    val, ex = m.validate(_Loader(batches, False, float(g["max_delay_s"]), fs), _esr)
    assert abs(val - float(g["rnn_val_loss"])) < 1e-5 and len(ex) == len(batches)
    for k, e in enumerate(ex):
        assert set(e) == {"input", "target", "prediction"}
        assert np.abs(e["prediction"].cpu().numpy() - g["rnn_pred"][k]).max() < TOL
        assert np.array_equal(e["input"].cpu().numpy(), batches[k][0][0, 0, 1024:])
        assert np.array_equal(e["target"].cpu().numpy(), batches[k][1][0, 0, 1024:])
    assert np.abs(m.hidden.cpu().numpy() - g["rnn_hidden"]).max() < TOL
    val2, ex2 = m.validate(_Loader(batches, False, float(g["max_delay_s"]), fs), _esr, store_examples=False)
    assert val2 == val and ex2 == []

    md = ntm.DiffDelRNN(1, 64, 1, max_delay=int(g["model_max_delay"]))
    md.load_state_dict(ntm.weights.load_state_dict(W_D))
    md = md.to("cuda").eval()
    val, ex = md.validate(_Loader(batches, True, float(g["max_delay_s"]), fs), _esr)
    assert abs(val - float(g["dd_val_loss"])) < 1e-5
    for k, e in enumerate(ex):
        assert set(e) == {"input", "target", "prediction", "prediction_pre_d"}
        assert np.abs(e["prediction_pre_d"].cpu().numpy() - g["dd_pre_d"][k]).max() < TOL
        assert np.abs(e["prediction"].cpu().numpy() - g["dd_pred"][k]).max() < TOL
    assert np.abs(md.hidden.cpu().numpy() - g["dd_hidden"]).max() < TOL
    assert np.abs(md.diffdel.buffer.cpu().numpy() - g["dd_buffer"]).max() < TOL
    # detach_hidden / detach_buffer (code/model.py:54-56, :322-324, :377-380): new tensors, same values
    h0, b0 = md.hidden, md.diffdel.buffer
    md.detach_hidden()
    assert md.hidden is not h0 and torch.equal(md.hidden, h0) and md.hidden.data_ptr() != h0.data_ptr()
    assert md.diffdel.buffer is not b0 and torch.equal(md.diffdel.buffer, b0)
    h0 = m.hidden
    m.detach_hidden()
    assert m.hidden is not h0 and torch.equal(m.hidden, h0)


# ----------------------------------------------------------------------------- cached warm-start state
def test_warm_start_cache_is_bit_identical_and_follows_the_parameters(ntm):
    """warm_start() from a fresh state is computed once per parameter version: a second predict() launches nothing for
    it and returns the same bits as a model that recomputes it (warm_cache = False); load_state_dict, a kernel-variant
    change and a different delay-line length each invalidate; goldens g1 / g3 still hold."""
    g1, g3 = load("g1_predict_16x8192.npz"), load("g3_warm_start.npz")
    x = dev(g1["x"][:3].reshape(3, 1, -1))
    cached, plain = ntm.harness.build_model(W_G), ntm.harness.build_model(W_G)
    plain.warm_cache = False
    ya = cached.predict(x)
    assert cached._warm is not None and plain._warm is None
    key = cached._warm[0]
    yb, yc = cached.predict(x), plain.predict(x)
    assert plain._warm is None and cached._warm[0] == key
    assert torch.equal(ya, yb) and torch.equal(ya, yc)
    cached.initialize_hidden(); cached.warm_start()
    assert np.abs(cached.hidden.cpu().numpy() - g3["wg_hidden"]).max() < 2e-6
    # the cached tensor is never handed out itself: mutating the model's state does not poison the cache
    cached.hidden.zero_()
    cached.initialize_hidden(); cached.warm_start()
    assert np.abs(cached.hidden.cpu().numpy() - g3["wg_hidden"]).max() < 2e-6
    # warm_start() continuing from an existing state (code/model.py:58-65 continues from self.hidden) is computed, not served
    cached.hidden = torch.full((1, 1, 64), 0.5, device="cuda")
    cached.warm_start()
    plain.hidden = torch.full((1, 1, 64), 0.5, device="cuda")
    plain.warm_start()
    assert torch.equal(cached.hidden, plain.hidden) and cached._warm[0] == key
    # new parameters -> recomputed
    cached.load_state_dict(ntm.weights.load_state_dict(W_G_ESR))
    cached.initialize_hidden(); cached.warm_start()
    assert cached._warm[0] != key and np.abs(cached.hidden.cpu().numpy() - g3["wg_esr_hidden"]).max() < 2e-6
    other = ntm.harness.build_model(W_G_ESR)
    other.warm_cache = False
    assert torch.equal(cached.predict(x), other.predict(x))
    # in-place parameter update (what an optimiser step does) -> recomputed
    k2 = cached._warm[0]
    with torch.no_grad():
        cached.output.bias.add_(0.25)
    y_shift = cached.predict(x)
    assert cached._warm[0] != k2 and torch.allclose(y_shift, other.predict(x) + 0.25, atol=1e-6)
    # kernel variant is part of the key (the variants differ in the last bits)
    k3 = cached._warm[0]
    cached.kernel_variant = "mfma2"
    cached.predict(x)
    assert cached._warm[0] != k3

    # DiffDelGRU: hidden AND delay buffer are cached, keyed by the delay-line length too
    md = ntm.DiffDelRNN(1, 64, 1, max_delay=300)
    md.load_state_dict(ntm.weights.load_state_dict(W_D))
    md = md.to("cuda").eval()
    assert md.warm_cache is False                 # the class default: recompute, like the reference
    md.warm_cache = True
    rng = np.random.default_rng(3)
    xd = dev(rng.uniform(-0.5, 0.5, (2, 1, 3000)).astype(np.float32))
    dd = dev((150 + 100 * np.sin(np.arange(3000) / 200.0))[None, None, :].repeat(2, 0).astype(np.float32))
    y1, p1 = md.predict(xd, dd)
    kd = md._warm[0]
    y2, p2 = md.predict(xd, dd)
    assert md._warm[0] == kd and torch.equal(y1, y2) and torch.equal(p1, p2)
    md.initialize_hidden(1, 300); md.warm_start()
    assert np.abs(md.hidden.cpu().numpy() - g3["wd_hidden"]).max() < 2e-6
    assert np.abs(md.diffdel.buffer.cpu().numpy() - g3["wd_buffer"]).max() < 2e-6
    md.warm_cache = False
    y3, p3 = md.predict(xd, dd)
    assert torch.equal(y1, y3) and torch.equal(p1, p3)
    md.warm_cache = True
    md.initialize_hidden(1, 200); md.warm_start()                     # another delay-line length: another state
    assert md._warm[0] != kd and md.diffdel.buffer.shape[-1] == 201
    # a buffer that is not fresh (batch of 2, or already used) is never served from the cache
    md.initialize_hidden(2, 300)
    with pytest.raises(RuntimeError):
        md.warm_start()                                               # B = 1 input against a batch-2 buffer, as in the reference


# ----------------------------------------------------------------------------- ntm_diffdel_gru_forward through raw ctypes
_DD_MODE = [None]      # None: ntm_diffdel_gru_forward; else the mode argument of ntm_diffdel_gru_forward_ex


def _dd_call(L, w, x, d, y, pre, h, buf, D, flag, warmup=0):
    B, T = x.shape
    args = [p(w["GRU.weight_ih_l0"]), p(w["GRU.weight_hh_l0"]), p(w["GRU.bias_ih_l0"]), p(w["GRU.bias_hh_l0"]),
            p(w["output.weight"]), 64, p(x), p(d), p(y), p(pre), B, T, p(h), p(buf), D, warmup, p(flag)]
    if _DD_MODE[0] is None:
        return L.ntm_diffdel_gru_forward(*args, None)
    return L.ntm_diffdel_gru_forward_ex(*args, _DD_MODE[0], None)


@pytest.mark.parametrize("mode", [None, 1, 2])
def test_diffdel_entry_point_raw_ctypes(ntm, mode):
    _DD_MODE[0] = mode
    try:
        _diffdel_entry_point_raw_ctypes(ntm)
    finally:
        _DD_MODE[0] = None


def _diffdel_entry_point_raw_ctypes(ntm):
    """include/ntm.h `ntm_diffdel_gru_forward` (replaces DiffDelRNN.forward, code/model.py:393-424) called directly:
    golden g5 (the reference's predict = warm-up call + forward), the oracle on a ragged batch, chunked == one-shot,
    warm-up mode, the refusal of pre_d == y, and a delay beyond D (sticky flag, dl_state untouched, later calls no-ops
    for the delay line until the flag is cleared)."""
    L = ntm._lib.lib()
    w = {k: v.cuda() for k, v in ntm.weights.load_state_dict(W_D).items()}
    g = load("g5_diffdel_predict.npz")
    D = int(g["D_effective"])
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    # g5 = predict: warm-up (1024 zeros, zero delay, B = 1) then the sequence, state carried through h / buf
    h, buf = torch.zeros(1, 64, device="cuda"), torch.zeros(1, D, device="cuda")
    z = torch.zeros(1, 1024, device="cuda")
    yz, pz = torch.empty_like(z), torch.empty_like(z)
    assert _dd_call(L, w, z, z, yz, pz, h, buf, D, flag) == 0
    x, d = dev(g["x"][0]), dev(g["d"][0])
    y, pre = torch.empty_like(x), torch.empty_like(x)
    assert _dd_call(L, w, x, d, y, pre, h, buf, D, flag) == 0
    torch.cuda.synchronize()
    assert int(flag.item()) == 0
    assert np.abs(pre.cpu().numpy() - g["pre_d"][0]).max() < TOL and np.abs(y.cpu().numpy() - g["y"][0]).max() < TOL
    assert np.abs(h.cpu().numpy() - g["hidden"][0]).max() < TOL and np.abs(buf.cpu().numpy() - g["buffer"][0]).max() < TOL
    # the same pre_d through the oracle's delay line: y bit for bit (the delay line is exact)
    yo, bo = oracle.delay_forward(pre.cpu().numpy(), g["d"][0], pz[:, -D:].cpu().numpy() if D <= 1024 else
                                  np.concatenate([np.zeros((1, D - 1024), np.float32), pz.cpu().numpy()], 1))
    assert np.array_equal(y.cpu().numpy(), yo) and np.array_equal(buf.cpu().numpy(), bo)

    # ragged batch against the oracle, one-shot and in three chunks (T not a multiple of anything, T < D in a chunk)
    rng = np.random.default_rng(11)
    B, T, D2 = 37, 2500, 301
    xs = rng.uniform(-0.5, 0.5, (B, T)).astype(np.float32)
    ds = np.clip(150 + 140 * np.sin(np.arange(T)[None, :] / rng.uniform(50, 400, (B, 1))) + rng.uniform(-5, 5, (B, 1)), 0, D2).astype(np.float32)
    h0 = rng.uniform(-0.3, 0.3, (B, 64)).astype(np.float32)
    b0 = rng.uniform(-0.3, 0.3, (B, D2)).astype(np.float32)
    wo = oracle_weights(W_D)
    yo, po, ho, bo = oracle.diffdel_forward(wo, xs, ds, h0, b0)
    h, buf = dev(h0), dev(b0)
    x, d = dev(xs), dev(ds)
    y, pre = torch.empty_like(x), torch.empty_like(x)
    assert _dd_call(L, w, x, d, y, pre, h, buf, D2, flag) == 0
    assert np.abs(pre.cpu().numpy() - po).max() < TOL and np.abs(y.cpu().numpy() - yo).max() < TOL
    assert np.abs(h.cpu().numpy() - ho).max() < TOL and np.abs(buf.cpu().numpy() - bo).max() < TOL
    h2, buf2 = dev(h0), dev(b0)
    ys, ps = [], []
    for c0, c1 in ((0, 1000), (1000, 1200), (1200, T)):                 # the middle chunk is shorter than the delay line
        xc, dc = x[:, c0:c1].contiguous(), d[:, c0:c1].contiguous()
        yc, pc = torch.empty_like(xc), torch.empty_like(xc)
        assert _dd_call(L, w, xc, dc, yc, pc, h2, buf2, D2, flag) == 0
        ys.append(yc); ps.append(pc)
    assert torch.equal(torch.cat(ys, 1), y) and torch.equal(torch.cat(ps, 1), pre)
    assert torch.equal(h2, h) and torch.equal(buf2, buf)
    # warm-up mode: y = pre_d, the buffer takes the tail of pre_d (code/model.py:288-292)
    h3, buf3 = dev(h0), dev(b0)
    yw, pw = torch.empty_like(x), torch.empty_like(x)
    assert _dd_call(L, w, x, d, yw, pw, h3, buf3, D2, flag, warmup=1) == 0
    assert torch.equal(yw, pw) and torch.equal(pw, pre) and torch.equal(buf3, pre[:, -D2:])
    torch.cuda.synchronize()
    assert int(flag.item()) == 0

    # refused: pre_d aliasing y, or missing
    assert _dd_call(L, w, x, d, y, y, h, buf, D2, flag) == -1 and b"distinct" in L.ntm_last_error()
    assert L.ntm_diffdel_gru_forward(p(w["GRU.weight_ih_l0"]), p(w["GRU.weight_hh_l0"]), p(w["GRU.bias_ih_l0"]), p(w["GRU.bias_hh_l0"]),
                                     p(w["output.weight"]), 64, p(x), p(d), p(y), None, B, T, p(h), p(buf), D2, 0, p(flag), None) == -1
    # a hidden size outside [1, NTM_MAX_HIDDEN]: refused through this entry as well
    assert L.ntm_diffdel_gru_forward(p(w["GRU.weight_ih_l0"]), p(w["GRU.weight_hh_l0"]), p(w["GRU.bias_ih_l0"]), p(w["GRU.bias_hh_l0"]),
                                     p(w["output.weight"]), 2000, p(x), p(d), p(y), p(pre), B, T, p(h), p(buf), D2, 0, p(flag), None) == -1
    assert b"[1, 1024]" in L.ntm_last_error()

    # a delay beyond D in ONE stream: the flag goes up, the delay state of EVERY stream stays as it was (the reference
    # asserts before it touches its buffer, code/model.py:284), the GRU state moves on (self.hidden is assigned at :412)
    bad = ds.copy()
    bad[5, 1234] = D2 + 0.25
    hb, bufb = dev(h0), dev(b0)
    assert _dd_call(L, w, x, dev(bad), y, pre, hb, bufb, D2, flag) == 0
    torch.cuda.synchronize()
    assert int(flag.item()) == 1 and torch.equal(bufb, dev(b0)) and torch.equal(hb, h)
    # sticky: a later good call leaves the delay state frozen until the caller clears the flag
    assert _dd_call(L, w, x, d, y, pre, hb, bufb, D2, flag) == 0
    torch.cuda.synchronize()
    assert int(flag.item()) == 1 and torch.equal(bufb, dev(b0))
    flag.zero_()
    hb = dev(h0)
    assert _dd_call(L, w, x, d, y, pre, hb, bufb, D2, flag) == 0
    torch.cuda.synchronize()
    assert int(flag.item()) == 0 and torch.equal(bufb, buf) and np.abs(y.cpu().numpy() - yo).max() < TOL
    # NaN counts as a violation too
    bad = ds.copy()
    bad[0, 0] = np.nan
    assert _dd_call(L, w, x, dev(bad), y, pre, dev(h0), dev(b0), D2, flag) == 0
    torch.cuda.synchronize()
    assert int(flag.item()) == 1


def test_c_abi_whole_hot_path_from_a_plain_cpp_process(ntm, tmp_path):
    """tools/cabi/cabi_demo `diffdel` mode (C++, raw HIP allocations, no torch, no Python in the process):
    ntm_diffdel_gru_forward in two chunks with carried state, the refusal, the range violation, then ntm_esr_sums --
    against the oracle and bit for bit against the Python layer."""
    exe = os.path.join(ROOT, "tools", "cabi", "cabi_demo.bin")
    if not os.path.exists(exe):
        subprocess.run(["make", "-s", "-C", os.path.dirname(exe)], check=True)
    rng = np.random.default_rng(21)
    B, T, D, split = 21, 1500, 301, 600
    x = rng.uniform(-0.5, 0.5, (B, T)).astype(np.float32)
    d = np.clip(120 + 100 * np.sin(np.arange(T)[None, :] / rng.uniform(40, 300, (B, 1))), 0, D).astype(np.float32)
    x.tofile(str(tmp_path / "x.f32")); d.tofile(str(tmp_path / "d.f32"))
    wfile = os.path.join(ROOT, "neural-tape-modeling_amd", "weights", "w2.bin")
    r = subprocess.run([exe, "diffdel", wfile, str(tmp_path / "x.f32"), str(tmp_path / "d.f32"), str(B), str(T), str(D), str(split),
                        str(tmp_path / "out")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr + r.stdout
    assert "flag_after_good_calls=0 flag_after_violation=1 buffer_frozen=1" in r.stdout and "distinct buffer" in r.stdout
    rd = lambda ext, dt, shape: np.fromfile(str(tmp_path / ("out." + ext)), dt).reshape(shape)     # noqa: E731
    y, pre, h, buf = rd("y", np.float32, (B, T)), rd("pre", np.float32, (B, T)), rd("h", np.float32, (B, 64)), rd("buf", np.float32, (B, D))
    esr = rd("esr", np.float64, (B, 2))
    yo, po, ho, bo = oracle.diffdel_forward(oracle_weights(W_D), x, d, None, np.zeros((B, D), np.float32))
    assert np.abs(pre - po).max() < TOL and np.abs(y - yo).max() < TOL and np.abs(h - ho).max() < TOL and np.abs(buf - bo).max() < TOL
    so = oracle.esr_sums(y[:, split:], pre[:, split:])
    assert np.abs(esr / so - 1).max() < 1e-9
    md = ntm.DiffDelRNN(1, 64, 1, max_delay=D - 1)
    md.load_state_dict(ntm.weights.load_state_dict(W_D))
    md = md.to("cuda").eval()
    md.initialize_hidden(B, D - 1)
    y1, p1 = md(dev(x[:, :split]).unsqueeze(1), dev(d[:, :split]).unsqueeze(1))
    y2, p2 = md(dev(x[:, split:]).unsqueeze(1), dev(d[:, split:]).unsqueeze(1))
    assert np.array_equal(torch.cat([y1, y2], 2)[:, 0].cpu().numpy(), y) and np.array_equal(torch.cat([p1, p2], 2)[:, 0].cpu().numpy(), pre)


# ----------------------------------------------------------------------------- eight ranks on the one GPU
def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


def test_bench_eight_ranks_share_the_gpu_over_gloo():
    """`python bench.py --gpus 8` (how the driver's 8-GPU node will start it, no launcher) with NTM_DIST_BACKEND=gloo so
    that the eight ranks share this box's GPU: weak (8 x 64 segments), strong with an uneven split (500 = 4 x 63 + 4 x 62),
    and a rank that dies after warm-up brings the whole job down, non-zero, well inside the timeout.
    Reference counterpart: none (scripts/sbatch-train-exp1a.sh:7 runs replicas only)."""
    import time
    common = ["--gpus", "8", "--steps", "2", "--warmup", "1", "--samples", "4096", "--no-cpu-baseline", "--no-extra"]
    env = _clean_env(NTM_DIST_BACKEND="gloo")
    for extra, total in ((["--batch", "64"], 512), (["--scaling", "strong", "--total-batch", "500"], 500)):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common + extra, env=env, capture_output=True,
                           text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        line, out = bench_record(r.stdout)
        assert line["n_gpus"] == 8 and line["roofline"]["aggregate"]["n_gpus"] == 8 and "legs" not in line
        assert out["n_gpus"] == 8 and out["ranks"] == 8 and out["backend"] == "gloo" and out["rccl_ranks"] == 0
        assert [d["rank"] for d in out["rank_devices"]] == list(range(8)) and all(d["device"] == 0 for d in out["rank_devices"])
        assert out["config"]["segments_total"] == total and out["checks"]["segments"] == total
        assert out["config"]["segments_rank0"] == (64 if total == 512 else 63)
        assert out["checks"]["job_esr"] > 0 and out["checks"]["every_timed_step_same_loss"] is True
        assert out["checks"]["last_output_equals_first_pass_bitwise"] is True and out["checks"]["no_warm_cache_steps_same_loss"] is True
        assert out["ms_per_step_no_warm_cache"] > 0
        assert abs(out["value"] - total * 4096 * 2 / (out["ms_per_step"] * 2e-3)) < 1e-6 * out["value"]
        assert "other_workloads" not in out
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common + ["--batch", "64", "--fail-rank", "5"], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode != 0 and "rank 5 exited with status 3" in r.stderr, r.stderr[-3000:]
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert time.monotonic() - t0 < 300


# ----------------------------------------------------------------------------- other_workloads of the default line
def test_bench_line_carries_the_other_workloads():
    """The default `bench.py --gpus 1` attaches configs[2], configs[3] and the GRU workload at other per-GPU batch sizes
    as `other_workloads` (here at small shapes, `--other on`): each with kernel, kernel_ms, roofline.frac, determinism
    and scattered streams against the oracle; the headline fields are those of a run without them."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "1040",
                        "--samples", "4096", "--no-extra", "--other", "on", "--other-steps", "2", "--other-gru-batches", "2048,4112", "--other-tcn-batches", "2048", "--other-diffdel-batches", "",
                        "--cli-segments", "0"],              # (other_workloads.cli: tests/test_gpu_round5.py)
                       env=_clean_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line, out = bench_record(r.stdout)
    ow = out["other_workloads"]
    assert set(line["legs"]) == set(ow) - {"note"} and all(0 < v["frac"] < 1 and v["value"] > 0 for v in line["legs"].values())
    assert set(ow) == {"note", "diffdel", "tcn", "gru_B2048", "gru_B4112", "tcn_B2048"} and "not a scaling curve" in ow["note"]
    for k, v in ow.items():
        if k == "note":
            continue
        assert v["kernel"] and v["kernel_ms"] > 0 and 0 < v["roofline"]["frac"] < 1.0 and v["roofline"]["kernel_ms"] == v["kernel_ms"]
        assert v["checks"]["deterministic"] is True and len(v["checks"]["streams_checked"]) == 4
        assert v["checks"]["vs_oracle_max_abs"] < TOL, (k, v["checks"])
        assert v["kernel_ms"] <= v["device_ms_per_step"] <= v["ms_per_step"] * 1.001
    assert ow["diffdel"]["bytes_per_sample"] == 16 and "1040 segments x 4096" in ow["diffdel"]["workload"]
    assert "2048 segments x 4096" in ow["gru_B2048"]["workload"]
    assert out["metric"].startswith("audio samples/sec") and out["cpu_baseline"]["value"] > 0 and out["checks"]["streams_vs_oracle"]["max_abs"] < TOL
    # round 4: the loss leg of the timed step means something -- a target that is not the output, sums checked against the oracle
    e = out["checks"]["esr_sums_vs_oracle"]
    assert out["checks"]["job_esr"] > 1e-4 and e["max_rel"] < 1e-9 and e["esr_max_rel_diff"] < 1e-3 and min(e["esr_device"]) > 0
    assert out["ms_per_step_no_warm_cache"] > 0 and ow["tcn"]["roofline"]["bytes_per_sample_if_fused"] == 8


# ----------------------------------------------------------------------------- the fused DiffDelRNN step
def _ddr(ntm, max_delay, mode):
    m = ntm.DiffDelRNN(1, 64, 1, max_delay=max_delay)
    m.load_state_dict(ntm.weights.load_state_dict(W_D))
    m = m.to("cuda").eval()
    m.delay_mode = mode
    return m


def _trajectories(rng, B, T, D):
    """Delay trajectories that reach every branch of the fused pass: slow wow (one 8-sample window per thread), white
    (general form), constant integer parts, a ramp through every integer part up to D itself, negative delays, delays so
    small that the taps are the newest samples (read back right behind the store), steps of exactly 3 and of 4 in the
    integer part inside one 4-sample group."""
    n = np.arange(T)
    d = np.empty((B, T), np.float32)
    for b in range(B):
        k = b % 8
        if k == 0:
            d[b] = 0.6 * D + 0.3 * D * np.sin(2 * np.pi * n / rng.uniform(500, 3000) + rng.uniform(0, 6))
        elif k == 1:
            d[b] = rng.uniform(0, D, T)
        elif k == 2:
            d[b] = np.floor(0.5 * D) + (n % 2) * 0.999
        elif k == 3:
            d[b] = np.clip(n.astype(np.float64) * D / max(T - 1, 1), 0, D)
            d[b, -1] = D
        elif k == 4:
            d[b] = np.where(n % 97 == 0, -0.5, 0.25 * D)
        elif k == 5:
            d[b] = np.abs(1.5 * np.sin(n / 7.0)) + (0 if D < 3 else 0.25 * ((n // 64) % 3))      # taps = the newest samples
        elif k == 6:
            d[b] = np.minimum(D, 5 + 3.0 * (n % 4 == 3) + 0.5)                                   # span 3 inside a group
        else:
            d[b] = np.minimum(D, 5 + 4.0 * (n % 4 == 2) + 0.25)                                  # span 4: general form
    return np.clip(d, -0.9, D).astype(np.float32)


@pytest.mark.parametrize("B,T,D", [(1, 1, 37), (5, 7, 37), (16, 63, 5), (17, 64, 37), (33, 65, 301), (8, 127, 37), (40, 130, 64),
                                   (19, 200, 301), (24, 1000, 37), (37, 2500, 301), (9, 4099, 1847), (64, 4096, 1847),
                                   (300, 1500, 128), (16, 3000, 1), (16, 997, 3)])
def test_fused_diffdel_step_equals_the_two_pass_step_bit_for_bit(ntm, B, T, D):
    """The delay line fused into the GRU kernel (ntm_diffdel_gru_forward, NTM_DIFFDEL_FUSED) against the GRU launch +
    streaming delay pass (NTM_DIFFDEL_TWO_PASS) on the same inputs and carried state: pre_d, y, hidden state and
    delay buffer bit for bit -- ragged batches and lengths (T % 4 != 0 takes the unaligned stores), T shorter than a
    tile / than the delay line, history taps out of a random carried buffer, every trajectory class; then chunked ==
    one-shot for the fused form, and y against the oracle's delay line on the GPU's own pre_d."""
    rng = np.random.default_rng(B * 100003 + T * 17 + D)
    x = rng.uniform(-0.5, 0.5, (B, T)).astype(np.float32)
    d = _trajectories(rng, B, T, D)
    h0 = rng.uniform(-0.3, 0.3, (B, 64)).astype(np.float32)
    b0 = rng.uniform(-0.3, 0.3, (B, D)).astype(np.float32)
    res = {}
    for mode in ("two_pass", "fused"):
        m = _ddr(ntm, D - 1, mode)
        if mode == "two_pass":
            m.kernel_variant = "mfma2"      # the same GRU kernel as the fused step ("auto" would pick the low-latency one here)
        m.initialize_hidden(B, D - 1)
        m.hidden, m.diffdel.buffer = dev(h0).view(1, B, 64), dev(b0).view(B, 1, D)
        y, pre = m(dev(x).unsqueeze(1), dev(d).unsqueeze(1))
        res[mode] = (y[:, 0].clone(), pre[:, 0].clone(), m.hidden.clone(), m.diffdel.buffer.clone())
    for a, b, what in zip(res["two_pass"], res["fused"], ("y", "pre_d", "hidden", "buffer")):
        assert torch.equal(a, b), f"{what}: max |diff| {float((a - b).abs().max())}"
    y, pre, h, buf = (t.cpu().numpy() for t in res["fused"])
    yo, bo = oracle.delay_forward(pre, d, b0)
    assert np.array_equal(y, yo) and np.array_equal(buf[:, 0], bo)
    if T >= 3:
        m = _ddr(ntm, D - 1, "fused")
        m.initialize_hidden(B, D - 1)
        m.hidden, m.diffdel.buffer = dev(h0).view(1, B, 64), dev(b0).view(B, 1, D)
        cuts = sorted({0, T // 3, min(T, T // 3 + 70), T})
        ys = [m(dev(x[:, c0:c1]).unsqueeze(1), dev(d[:, c0:c1]).unsqueeze(1))[0] for c0, c1 in zip(cuts[:-1], cuts[1:]) if c1 > c0]
        assert torch.equal(torch.cat(ys, 2)[:, 0], res["fused"][0])
        assert torch.equal(m.hidden, res["fused"][2]) and torch.equal(m.diffdel.buffer, res["fused"][3])


def test_fused_diffdel_step_warmup_violation_and_predict(ntm):
    """Fused form: warm-up mode (y = pre_d, buffer filled), a delay beyond D (AssertionError, delay state untouched,
    hidden moved on -- as the two-pass form and the reference), the deferred check of predict(), golden g5 and g8."""
    rng = np.random.default_rng(77)
    B, T, D = 21, 700, 301
    x = dev(rng.uniform(-0.5, 0.5, (B, 1, T)).astype(np.float32))
    d_np = _trajectories(rng, B, T, D)
    out = {}
    for mode in ("two_pass", "fused"):
        m = _ddr(ntm, D - 1, mode)
        if mode == "two_pass":
            m.kernel_variant = "mfma2"
        m.initialize_hidden(B, D - 1)
        yw, pw = m(x, dev(d_np).unsqueeze(1), warmup=True)
        assert torch.equal(yw, pw)
        y2, p2 = m(x, dev(d_np).unsqueeze(1))
        bad = d_np.copy()
        bad[3, 500] = D + 0.5
        buf_before, h_before = m.diffdel.buffer.clone(), m.hidden.clone()
        with pytest.raises(AssertionError):
            m(x, dev(bad).unsqueeze(1))
        assert torch.equal(m.diffdel.buffer, buf_before) and not torch.equal(m.hidden, h_before)
        y3, _ = m(x, dev(d_np).unsqueeze(1))                            # usable again after the raise
        out[mode] = (yw, y2, y3, m.hidden.clone(), m.diffdel.buffer.clone())
    for a, b in zip(out["two_pass"], out["fused"]):
        assert torch.equal(a, b)
    g = load("g5_diffdel_predict.npz")
    m = _ddr(ntm, int(g["max_delay"]), "fused")
    y, pre = m.predict(dev(g["x"]), dev(g["d"]))
    assert np.abs(pre.cpu().numpy() - g["pre_d"]).max() < TOL and np.abs(y.cpu().numpy() - g["y"]).max() < TOL
    assert np.abs(m.diffdel.buffer.cpu().numpy() - g["buffer"]).max() < TOL and np.abs(m.hidden.cpu().numpy() - g["hidden"]).max() < TOL
    bad = g["d"].copy()
    bad[0, 0, 4000] = int(g["D_effective"]) + 1.0
    with pytest.raises(AssertionError):
        m.predict(dev(g["x"]), dev(bad))
    g8 = load("g8_diffdel_batched.npz")
    m = _ddr(ntm, int(g8["max_delay"]), "fused")
    xb, db = dev(g8["x"]), dev(g8["d"])
    init, chunk = int(g8["init_len"]), int(g8["chunk"])
    m.initialize_hidden(xb.shape[0], m.max_delay)
    ys, ps = [], []
    yy, pp = m(xb[:, :, :init], db[:, :, :init], warmup=True)
    ys.append(yy); ps.append(pp)
    for c0 in range(init, xb.shape[2], chunk):
        yy, pp = m(xb[:, :, c0:c0 + chunk], db[:, :, c0:c0 + chunk])
        ys.append(yy); ps.append(pp)
    assert np.abs(torch.cat(ys, 2).cpu().numpy() - g8["y"]).max() < TOL and np.abs(torch.cat(ps, 2).cpu().numpy() - g8["pre_d"]).max() < TOL
    assert np.abs(m.diffdel.buffer.cpu().numpy() - g8["buffer"]).max() < TOL


FULL = pytest.mark.skipif(os.environ.get("NTM_SKIP_FULL") == "1", reason="full 4096x65536 pass skipped on request")


@FULL
def test_fused_diffdel_full_size_equals_two_pass_and_oracle(ntm):
    """BASELINE configs[2] at full size (4096 distinct streams x 65 536 samples, D = 1847, the bench's trajectories): the
    fused launch (what `auto` picks here) against the two-pass step over the WHOLE batch bit for bit, scattered streams
    against the oracle, and a batch with a remainder (4096 + 40 streams: fused kernel + low-latency kernel + streaming pass
    + one buffer update)."""
    sys.path.insert(0, ROOT)
    import bench
    B, T = 4096, 65536
    devc = torch.device("cuda", 0)
    x = bench.synth_input(B, T, devc, seed=1234)
    ma = ntm.harness.build_model(W_D, max_delay_seconds=0.0335)
    d = bench.delay_trajectories(B, T, devc, ma.max_delay)
    assert ma.delay_mode == "auto"
    ya, pa = ma.predict(x, d)
    ha, ba = ma.hidden.clone(), ma.diffdel.buffer.clone()
    mt = ntm.harness.build_model(W_D, max_delay_seconds=0.0335)
    mt.delay_mode = "two_pass"
    yt, pt = mt.predict(x, d)
    assert torch.equal(pa, pt) and torch.equal(ya, yt) and torch.equal(ha, mt.hidden) and torch.equal(ba, mt.diffdel.buffer)
    del yt, pt
    rows = [0, 15, 16, 2047, 2048, 4095]
    yo, po, ho, bo = oracle.diffdel_predict(oracle_weights(W_D), x[rows, 0].cpu().numpy(), d[rows, 0].cpu().numpy(), ma.max_delay, threads=8)
    assert np.abs(pa[rows, 0].cpu().numpy() - po).max() < TOL and np.abs(ya[rows, 0].cpu().numpy() - yo).max() < TOL
    assert np.abs(ba[rows, 0].cpu().numpy() - bo).max() < TOL
    del ya, pa
    # remainder: B = one device round of the matrix-pipe kernel + 40 streams
    Br, Tr = 4096 + 40, 4096
    xr, dr = x[:, :, :Tr].contiguous(), d[:, :, :Tr].contiguous()
    xr = torch.cat([xr, xr[:40] * 0.5], 0)
    dr = torch.cat([dr, dr[100:140]], 0)
    y1, p1 = ma.predict(xr, dr)
    y2, p2 = mt.predict(xr, dr)
    assert y1.shape[0] == Br and torch.equal(y1, y2) and torch.equal(p1, p2) and torch.equal(ma.diffdel.buffer, mt.diffdel.buffer)


@pytest.mark.parametrize("B,T,D", [(5136, 1000, 301), (8192 + 16, 515, 64)])
def test_fused_diffdel_step_many_groups_build(ntm, B, T, D):
    """More stream groups than CUs: the fused step runs the small-LDS build of the kernel (YPN = 4, two workgroups per CU)
    -- `auto` picks it for these batches (B = one device round + more than 1024 streams).  Against the two-pass step with
    the same GRU kernel: bit for bit; scattered streams against the oracle."""
    rng = np.random.default_rng(B + T)
    x = torch.from_numpy(rng.uniform(-0.5, 0.5, (B, 1, T)).astype(np.float32)).cuda()
    n = np.arange(T)
    d_np = np.clip(0.5 * D + 0.45 * D * np.sin(2 * np.pi * n[None, :] / rng.uniform(50, 900, (B, 1)) + rng.uniform(0, 6, (B, 1))), 0, D).astype(np.float32)
    d_np[:, :8] = rng.uniform(-0.5, D, (B, 8))
    d = torch.from_numpy(d_np).cuda().unsqueeze(1)
    h0 = rng.uniform(-0.3, 0.3, (B, 64)).astype(np.float32)
    b0 = rng.uniform(-0.3, 0.3, (B, D)).astype(np.float32)
    res = {}
    for mode in ("two_pass", "auto"):
        m = _ddr(ntm, D - 1, mode)
        if mode == "two_pass":
            m.kernel_variant = "mfma2"
        m.initialize_hidden(B, D - 1)           # same carried state for both (predict()'s warm-up would come from two
        m.hidden, m.diffdel.buffer = dev(h0).view(1, B, 64), dev(b0).view(B, 1, D)      # different kernels at B = 1)
        y, pre = m(x, d)
        res[mode] = (y, pre, m.hidden.clone(), m.diffdel.buffer.clone())
    # `auto` gives a remainder of at most 1024 streams behind whole device rounds to the low-latency kernel + the streaming
    # pass (B = 8208: 8192 fused + 16): those streams agree with the matrix-pipe kernel to rounding, the others bit for bit
    nf = B if B % 4096 > 1024 or B % 4096 == 0 else (B // 4096) * 4096
    for a, b, what in zip(res["two_pass"], res["auto"], ("y", "pre_d", "hidden", "buffer")):
        a, b = (a[0], b[0]) if what == "hidden" else (a, b)
        assert torch.equal(a[:nf], b[:nf]), what
        assert nf == B or float((a[nf:] - b[nf:]).abs().max()) < TOL, what
    rows = [0, 15, 16, 4095, 4096, 4111, B - 1]
    yo, po, ho, bo = oracle.diffdel_forward(oracle_weights(W_D), x[rows, 0].cpu().numpy(), d_np[rows], h0[rows], b0[rows], threads=4)
    assert np.abs(res["auto"][1][rows, 0].cpu().numpy() - po).max() < TOL and np.abs(res["auto"][0][rows, 0].cpu().numpy() - yo).max() < TOL


# ----------------------------------------------------------------------------- forward + ESR sums in one launch
@pytest.mark.parametrize("B,T,skip", [(1040, 64, 0), (1040, 100, 4), (2050, 4096, 1024), (4096 + 40, 1500, 1024), (1100, 333, 332),
                                     (5136, 1500, 1024), (8192, 700, 64),        # more groups than CUs: the small-LDS build
                                     (5000, 700, 5), (16, 900, 64), (1, 70, 0), (1040, 130, 128)])
def test_forward_esr_equals_forward_plus_esr_pass(ntm, B, T, skip):
    """RNN.forward_esr / ntm_gru_forward_esr (the loss leg accumulated in the recurrent launch's output flush) against
    forward() followed by the streaming ESR pass: y and the carried state bit for bit, the per-stream sums to fp64
    summation order, and against the oracle.  Ragged batches and lengths, skip at / beyond tile borders, a skip that is
    not a multiple of 4 and small batches (both: forward launch + streaming pass inside the entry point), a remainder
    behind a whole device round (fused kernel + low-latency kernel)."""
    rng = np.random.default_rng(B + 7 * T + skip)
    x = rng.uniform(-0.5, 0.5, (B, 1, T)).astype(np.float32)
    t = (0.3 * np.tanh(2 * x) + 0.02 * rng.standard_normal(x.shape)).astype(np.float32)
    h0 = rng.uniform(-0.3, 0.3, (1, B, 64)).astype(np.float32)
    m = ntm.harness.build_model(W_G)
    m.hidden = dev(h0)
    y1, s1 = m.forward_esr(dev(x), dev(t), skip)
    h1 = m.hidden.clone()
    m.hidden = dev(h0)
    y2 = m.forward(dev(x))
    s2 = ntm.model.esr_sums(y2, dev(t), skip)
    assert torch.equal(y1, y2) and torch.equal(h1, m.hidden)
    assert s1.shape == (B, 2) and s1.dtype == torch.float64
    a, b = s1.cpu().numpy(), s2.cpu().numpy()
    assert np.abs(a - b).max() <= 1e-12 * max(1.0, np.abs(b).max())
    rows = sorted({0, B // 2, B - 1})
    so = oracle.esr_sums(y1[rows, 0].cpu().numpy(), t[rows, 0], skip)
    assert np.abs(a[rows] - so).max() <= 1e-9 * max(1.0, np.abs(so).max())
    if skip == T:
        assert not a.any()


def test_forward_esr_raw_entry_point_and_predict_esr(ntm):
    """ntm_gru_forward_esr through raw ctypes: a strided x view, refusals (target aliasing y, negative skip, null
    pointers), and predict_esr = predict + esr_sums on golden g1's programme material; a model with an explicitly chosen
    kernel variant or a skip connection takes the two calls and gives the same numbers."""
    L = ntm._lib.lib()
    g = load("g1_predict_16x8192.npz")
    x = dev(np.repeat(g["x"].reshape(16, 1, -1), 70, axis=0))            # 1120 streams: the matrix-pipe kernel
    tgt = dev(np.repeat(g["y"].reshape(16, 1, -1), 70, axis=0))
    m = ntm.harness.build_model(W_G)
    y, s = m.predict_esr(x, tgt, skip=1024)
    y_ref = m.predict(x)
    assert torch.equal(y, y_ref)
    s_ref = ntm.model.esr_sums(y_ref, tgt, 1024)
    assert np.abs(s.cpu().numpy() - s_ref.cpu().numpy()).max() <= 1e-12 * float(s_ref.max())
    n = x.shape[-1] - 1024
    esr = ((s[:, 0] / n) / (s[:, 1] / n + ntm.model.ESR_EPS)).cpu().numpy()
    assert esr.max() < 1e-9                                              # the target IS the reference's own output
    for variant, skipc in (("mfma2", False), ("auto", True)):
        m2 = ntm.harness.build_model(W_G)
        m2.kernel_variant, m2.skip = variant, skipc
        y2, s2 = m2.predict_esr(x, tgt, skip=1024)
        y2r = m2.predict(x)
        assert torch.equal(y2, y2r)
        assert np.abs(s2.cpu().numpy() - ntm.model.esr_sums(y2r, tgt, 1024).cpu().numpy()).max() <= 1e-12 * float(s2.max())
    # raw entry: row-strided x
    w = {k: v.cuda() for k, v in ntm.weights.load_state_dict(W_G).items()}
    B, T = 1040, 256
    big = torch.rand(B, 300, device="cuda") - 0.5
    xv = big[:, 20:20 + T]
    t2 = torch.rand(B, T, device="cuda") - 0.5
    yv, out = torch.empty(B, T, device="cuda"), torch.empty(B, 2, device="cuda", dtype=torch.float64)
    h = torch.zeros(B, 64, device="cuda")
    args = lambda tg, sk, o: [p(w["GRU.weight_ih_l0"]), p(w["GRU.weight_hh_l0"]), p(w["GRU.bias_ih_l0"]), p(w["GRU.bias_hh_l0"]),   # noqa: E731
                              p(w["output.weight"]), p(w["output.bias"]), 64, p(xv), p(yv), B, T, 300, T, p(h), tg, sk, o, None]
    assert L.ntm_gru_forward_esr(*args(p(t2), 0, p(out))) == 0
    mm = ntm.harness.build_model(W_G)
    mm.initialize_hidden()
    yr = mm(xv.contiguous().unsqueeze(1))
    assert torch.equal(yv, yr[:, 0]) and torch.equal(h.unsqueeze(0), mm.hidden)
    assert np.abs(out.cpu().numpy() - ntm.model.esr_sums(yr, t2.unsqueeze(1), 0).cpu().numpy()).max() < 1e-10
    assert L.ntm_gru_forward_esr(*args(p(yv), 0, p(out))) == -1 and b"alias" in L.ntm_last_error()
    assert L.ntm_gru_forward_esr(*args(p(t2), -4, p(out))) == -1
    assert L.ntm_gru_forward_esr(*args(None, 0, p(out))) == -1 and L.ntm_gru_forward_esr(*args(p(t2), 0, None)) == -1


@pytest.mark.parametrize("B,T,D,skip", [(1040, 64, 37, 0), (1100, 333, 64, 128), (2050, 2000, 301, 1024), (5136, 900, 128, 64),
                                       (4096 + 40, 700, 301, 256), (1040, 130, 37, 6), (24, 500, 64, 64), (1, 70, 5, 0)])
def test_diffdel_forward_esr_equals_forward_plus_esr_pass(ntm, B, T, D, skip):
    """DiffDelRNN.forward_esr / ntm_diffdel_gru_forward_esr (the loss leg accumulated in the fused delay stage, on the DELAYED
    output) against forward() + the streaming ESR pass: y, pre_d and the carried state bit for bit, the sums to fp64 summation
    order and against the oracle; trajectories with history taps, general-form lanes and ragged tails; a skip that is not a
    multiple of 4, small batches, the many-groups build and a remainder behind a device round."""
    rng = np.random.default_rng(3 * B + 5 * T + D + skip)
    x = rng.uniform(-0.5, 0.5, (B, 1, T)).astype(np.float32)
    d = _trajectories(rng, B, T, D)
    t = (0.3 * np.tanh(2 * x) + 0.02 * rng.standard_normal(x.shape)).astype(np.float32)
    h0 = rng.uniform(-0.3, 0.3, (1, B, 64)).astype(np.float32)
    b0 = rng.uniform(-0.3, 0.3, (B, 1, D)).astype(np.float32)
    out = {}
    for fused in (True, False):
        m = _ddr(ntm, D - 1, "auto")
        m.initialize_hidden(B, D - 1)
        m.hidden, m.diffdel.buffer = dev(h0), dev(b0)
        if fused:
            y, pre, s = m.forward_esr(dev(x), dev(d).unsqueeze(1), dev(t), skip)
        else:
            y, pre = m.forward(dev(x), dev(d).unsqueeze(1))
            s = ntm.model.esr_sums(y, dev(t), skip)
        out[fused] = (y, pre, m.hidden.clone(), m.diffdel.buffer.clone(), s)
    for a, b in zip(out[True][:4], out[False][:4]):
        assert torch.equal(a, b)
    sa, sb = out[True][4].cpu().numpy(), out[False][4].cpu().numpy()
    assert sa.shape == (B, 2) and np.abs(sa - sb).max() <= 1e-12 * max(1.0, np.abs(sb).max())
    rows = sorted({0, B // 2, B - 1})
    so = oracle.esr_sums(out[True][0][rows, 0].cpu().numpy(), t[rows, 0], skip)
    assert np.abs(sa[rows] - so).max() <= 1e-9 * max(1.0, np.abs(so).max())


def test_diffdel_predict_esr_and_compute_loss(ntm):
    """predict_esr on golden g5 repeated over a batch (the reference's own DiffDelGRU output as target: ESR ~ 0), the deferred
    range check still raises, and harness.compute_loss (code/test-model.py:332-398) gives the ESR of predict() + esr_sums."""
    g = load("g5_diffdel_predict.npz")
    B = 1056
    x, d, tgt = dev(np.repeat(g["x"], B, 0)), dev(np.repeat(g["d"], B, 0)), dev(np.repeat(g["y"], B, 0))
    m = _ddr(ntm, int(g["max_delay"]), "auto")
    y, pre, s = m.predict_esr(x, d, tgt, skip=1024)
    y2, pre2 = m.predict(x, d)
    assert torch.equal(y, y2) and torch.equal(pre, pre2)
    n = x.shape[-1] - 1024
    esr = ((s[:, 0] / n) / (s[:, 1] / n + ntm.model.ESR_EPS)).cpu().numpy()
    assert esr.max() < 1e-8
    bad = d.clone()
    bad[7, 0, 3000] = int(g["D_effective"]) + 2.0
    with pytest.raises(AssertionError):
        m.predict_esr(x, bad, tgt, skip=1024)
    res, out = ntm.harness.compute_loss(m, x, tgt, d_traj=d, INIT_LEN=1024)
    s_ref = ntm.model.esr_sums(y2, tgt, 1024)
    want = float(((s_ref[:, 0] / n) / (s_ref[:, 1] / n + ntm.model.ESR_EPS)).mean())
    assert abs(res["ESR"] - want) <= 1e-12 + 1e-9 * want and torch.equal(out, y2)
