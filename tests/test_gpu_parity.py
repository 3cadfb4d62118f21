"""GPU parity: the HIP path (through the C ABI / host mirror) against the CPU oracle and the golden
vectors generated from the reference.  Tolerance for the GRU path: 1e-5 abs fp32 (BASELINE.json
north_star); the delay line is bit-exact; ESR sums 1e-9 relative (fp64 accumulation order).
"""
import ctypes
import os

import numpy as np
import pytest
import torch

import oracle
from helpers import load, oracle_weights

pytestmark = pytest.mark.gpu

TOL = 1e-5
W_G = "GRU-HS[64]-L[DCPreESR]-DS[ReelToReel_Dataset_MiniPulse100_CHOWTAPE]_BEST"
W_D = "DiffDelGRU-HS[64]-L[DCPreESR]-DS[ReelToReel_Dataset_MiniPulse100_CHOWTAPE_WOWFLUTTER]_BEST"
VARIANTS = ["mfma2", "mfma", "valu", "f16x3", "lat", "bf16x3"]


@pytest.fixture(scope="module")
def ntm():
    import ntm_amd
    assert torch.cuda.is_available(), "these tests need the MI355X"
    ntm_amd._lib.lib()       # raises if libntm.so is missing: no silent fallback
    return ntm_amd


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def make_rnn(ntm, name=W_G, variant="auto"):
    m = ntm.RNN(1, ntm.parse_hidden_size(name), 1, skip=False)
    m.load_state_dict(ntm.weights.load_state_dict(name))
    m = m.to("cuda").eval()
    m.kernel_variant = variant
    return m


def make_ddr(ntm, max_delay, name=W_D, variant="auto"):
    m = ntm.DiffDelRNN(1, ntm.parse_hidden_size(name), 1, skip=False, max_delay=max_delay)
    m.load_state_dict(ntm.weights.load_state_dict(name))
    m = m.to("cuda").eval()
    m.kernel_variant = variant
    return m


def test_lane_group_transpose(ntm):
    """The in-register 4x4 transpose of the MFMA2 kernel (v_permlane32_swap / v_permlane16_swap):
    out[i] at lane group k == in[k] at lane group i, per wave, same lane-in-group."""
    L = ntm._lib.lab()                                                    # diagnostics live in libntm_lab.so
    a = np.arange(256 * 4, dtype=np.float32).reshape(4, 4, 16, 4)        # [wave][group][lane][reg]
    x = dev(a.reshape(256, 4))
    out = torch.empty_like(x)
    assert L.ntm_debug_transpose4(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(out.data_ptr()), None) == 0
    torch.cuda.synchronize()
    got = out.cpu().numpy().reshape(4, 4, 16, 4)
    want = a.transpose(0, 3, 2, 1)                                        # swap group <-> reg
    assert np.array_equal(got, want)


# ----------------------------------------------------------------------------- GRU vs goldens
@pytest.mark.parametrize("variant", VARIANTS)
def test_g1_predict_real_material_batched(ntm, variant):
    """cfg 1: 16 x 8192 real programme material; batched predict == the reference's B=1 predicts."""
    g = load("g1_predict_16x8192.npz")
    m = make_rnn(ntm, str(g["weights"]), variant)
    y = m.predict(dev(g["x"]).unsqueeze(1)).cpu().numpy()[:, 0, :]
    assert np.abs(y - g["y"]).max() < TOL
    # and one stream alone (B=1, exactly the reference call)
    y1 = m.predict(dev(g["x"][3:4]).unsqueeze(1)).cpu().numpy()[:, 0, :]
    assert np.abs(y1 - g["y"][3:4]).max() < TOL
    # the reference's 2048-sample chunk loop gives the same numbers as one persistent launch
    y2 = m.predict(dev(g["x"]).unsqueeze(1), segment_length=2048).cpu().numpy()[:, 0, :]
    assert np.array_equal(y2, y)


@pytest.mark.parametrize("variant", VARIANTS)
def test_g2_forward_state_carry_and_f64(ntm, variant):
    g = load("g2_forward_carry.npz")
    m = make_rnn(ntm, str(g["weights"]), variant)
    x = dev(g["x"])
    m.initialize_hidden()
    y0 = m(x[:, :, :1500])
    y1 = m(x[:, :, 1500:])
    y = torch.cat([y0, y1], 2).cpu().numpy()
    assert y.shape == g["y"].shape
    assert np.abs(y - g["y"]).max() < TOL
    assert tuple(m.hidden.shape) == (1, 4, 64)
    assert np.abs(m.hidden.cpu().numpy() - g["hidden"]).max() < TOL
    m.initialize_hidden()
    y64 = m(dev(g["x"][:2, :, :256].astype(np.float64)))
    assert y64.dtype == torch.float32
    assert np.abs(y64.cpu().numpy() - g["y64"]).max() < TOL


@pytest.mark.parametrize("variant", VARIANTS)
def test_g3_warm_start(ntm, variant):
    g = load("g3_warm_start.npz")
    m = make_rnn(ntm, W_G, variant)
    m.initialize_hidden()
    m.warm_start()
    assert np.abs(m.hidden.cpu().numpy() - g["wg_hidden"]).max() < TOL
    d = make_ddr(ntm, 300, W_D, variant)
    d.initialize_hidden(1, d.max_delay)
    d.warm_start()
    assert np.abs(d.hidden.cpu().numpy() - g["wd_hidden"]).max() < TOL
    assert np.abs(d.diffdel.buffer.cpu().numpy() - g["wd_buffer"]).max() < TOL


@pytest.mark.parametrize("variant", VARIANTS)
def test_g6_long_sequence(ntm, variant):
    g = load("g6_long_65536.npz")
    m = make_rnn(ntm, str(g["weights"]), variant)
    y = m.predict(dev(g["x"])).cpu().numpy()
    err = np.abs(y - g["y"])
    assert err.max() < TOL
    assert err[0, 0, -8192:].max() < 2 * max(err[0, 0, :8192].max(), 1e-6)     # no drift


# ----------------------------------------------------------------------------- GRU vs oracle, ragged shapes
@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("B,T", [(1, 1), (1, 63), (2, 64), (3, 65), (17, 129), (37, 700), (16, 2048), (33, 1)])
def test_ragged_shapes_vs_oracle(ntm, variant, B, T):
    w = oracle_weights(W_G)
    rng = np.random.default_rng(B * 1000 + T)
    x = rng.uniform(-0.6, 0.6, (B, T)).astype(np.float32)
    h0 = rng.uniform(-0.5, 0.5, (B, 64)).astype(np.float32)
    yo, ho = oracle.gru_forward(w, x, h0)
    m = make_rnn(ntm, W_G, variant)
    m.hidden = dev(h0).view(1, B, 64)
    y = m(dev(x).unsqueeze(1)).cpu().numpy()[:, 0, :]
    assert np.abs(y - yo).max() < TOL
    assert np.abs(m.hidden.cpu().numpy()[0] - ho).max() < TOL


@pytest.mark.parametrize("variant", ["mfma2", "f16x3", "bf16x3"])
def test_many_groups_variant(ntm, variant):
    """B >= 8192 selects the small-LDS build of the MFMA2 kernel (several workgroups per CU, head partial
    pre-reduced with permlane swaps): check it against the oracle and against the one-per-CU build."""
    B, T = 8192 + 16, 192
    rng = np.random.default_rng(4)
    base = rng.uniform(-0.6, 0.6, (8, T)).astype(np.float32)
    x = np.tile(base, (B // 8, 1))
    m = make_rnn(ntm, W_G, variant)
    y = m.predict(dev(x).unsqueeze(1))[:, 0]
    yo, _ = oracle.gru_predict(oracle_weights(W_G), base)
    assert np.abs(y[:8].cpu().numpy() - yo).max() < TOL
    yv = y.view(B // 8, 8, T)
    assert torch.equal(yv, yv[:1].expand_as(yv))
    y_small = m.predict(dev(x[:64]).unsqueeze(1))[:, 0]          # B = 64: the 16-plane build
    assert (y_small - y[:64]).abs().max().item() < 2e-6          # (plane summation order differs)


def test_variants_agree_and_raw_abi_strides(ntm):
    """Call the C ABI directly: row strides > T, NULL h_state; the product kernels through libntm.so and the
    laboratory kernels (independent implementations of the same arithmetic) through libntm_lab.so."""
    L, LAB = ntm._lib.lib(), ntm._lib.lab()
    sd = {k: v.cuda() for k, v in ntm.weights.load_state_dict(W_G).items()}
    B, T, XS, YS = 19, 333, 400, 352
    rng = np.random.default_rng(7)
    xh = rng.uniform(-0.5, 0.5, (B, XS)).astype(np.float32)
    x = dev(xh)
    outs = []
    for name in ("mfma2", "lat", "f16x3", "mfma", "valu", "bf16x3"):
        y = torch.full((B, YS), 7.0, device="cuda")
        fn = LAB.ntm_lab_gru_forward if name in ntm._lib.LAB_VARIANTS else L.ntm_gru_forward_ex
        rc = fn(*[ctypes.c_void_p(sd[k].data_ptr()) for k in
                  ["GRU.weight_ih_l0", "GRU.weight_hh_l0", "GRU.bias_ih_l0", "GRU.bias_hh_l0",
                   "output.weight", "output.bias"]],
                64, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()), B, T, XS, YS,
                None, ntm._lib.VARIANTS[name], None)
        assert rc == 0, (L.ntm_last_error(), LAB.ntm_lab_last_error())
        torch.cuda.synchronize()
        yh = y.cpu().numpy()
        assert np.all(yh[:, T:] == 7.0)                       # nothing written past T
        outs.append(yh[:, :T])
    yo, _ = oracle.gru_forward(oracle_weights(W_G), xh[:, :T])
    for o in outs:
        assert np.abs(o - yo).max() < TOL
        assert np.abs(o - outs[0]).max() < (2e-6 if o is not outs[2] else TOL)     # outs[2]: the opt-in f16x3 engine
    # a laboratory variant asked of the product library is refused with a pointer to the right library
    y = torch.empty(B, YS, device="cuda")
    rc = L.ntm_gru_forward_ex(*[ctypes.c_void_p(sd[k].data_ptr()) for k in
                                ["GRU.weight_ih_l0", "GRU.weight_hh_l0", "GRU.bias_ih_l0", "GRU.bias_hh_l0",
                                 "output.weight", "output.bias"]],
                              64, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()), B, T, XS, YS, None,
                              ntm._lib.VARIANTS["valu"], None)
    assert rc == -1 and b"libntm_lab.so" in L.ntm_last_error()


def test_abi_errors(ntm):
    L = ntm._lib.lib()
    assert L.ntm_gru_forward(None, None, None, None, None, None, 1025, None, None, 1, 1, 1, 1, None, None) == -1
    assert b"[1, 1024]" in L.ntm_last_error()
    assert L.ntm_gru_forward(None, None, None, None, None, None, 24, None, None, 1, 1, 1, 1, None, None) == -1
    assert b"null pointer" in L.ntm_last_error()                  # round 5: any hidden size in range is taken
    assert L.ntm_gru_forward(None, None, None, None, None, None, 64, None, None, 0, 10, 10, 10, None, None) == 0
    assert L.ntm_gru_forward(None, None, None, None, None, None, 64, None, None, 2, 10, 10, 10, None, None) == -1
    m = make_rnn(ntm)
    m.hidden = torch.zeros(1, 1, 64, device="cuda")
    with pytest.raises(RuntimeError, match=r"Expected hidden size \(1, 2, 64\)"):
        m(torch.zeros(2, 1, 8, device="cuda"))                # the reference's B>1-after-warm-start error
    with pytest.raises(RuntimeError, match="HIP device only"):
        m(torch.zeros(1, 1, 8))


# ----------------------------------------------------------------------------- delay line
def test_g4_delay_line_bit_exact(ntm):
    g = load("g4_delay_line.npz")
    D = int(g["D"])
    dl = ntm.TimeVaryingDelayLine(max_delay=D)
    dl.init_buffer(3, D)
    x, d = dev(g["x"]), dev(g["d"])
    bounds = g["bounds"]
    for i, (a, b) in enumerate(zip(bounds[:-1], bounds[1:])):
        y = dl(x[:, :, a:b], d[:, :, a:b]).cpu().numpy()
        assert np.array_equal(y, g["y"][:, :, a:b]), f"chunk {i}"
        assert np.array_equal(dl.buffer.cpu().numpy(), g["buf_after_each"][i]), f"buffer {i}"
    dl2 = ntm.TimeVaryingDelayLine(max_delay=D)
    dl2.init_buffer(3, D)
    yw = dl2(x[:, :, :50], d[:, :, :50], warmup=True).cpu().numpy()
    assert np.array_equal(yw, g["y_warm"]) and np.array_equal(dl2.buffer.cpu().numpy(), g["buf_warm"])
    assert np.array_equal(dl2(x[:, :, 50:100], d[:, :, 50:100]).cpu().numpy(), g["y_after_warm"])


def test_delay_random_vs_oracle_and_assert(ntm):
    rng = np.random.default_rng(11)
    B, T, D = 5, 5000, 1847
    x = rng.standard_normal((B, T)).astype(np.float32)
    d = rng.uniform(0, D, (B, T)).astype(np.float32)
    buf0 = rng.standard_normal((B, D)).astype(np.float32)
    dl = ntm.TimeVaryingDelayLine(max_delay=D)
    dl.init_buffer(B, D)
    dl.buffer = dev(buf0).view(B, 1, D)
    yo, bo = oracle.delay_forward(x, d, buf0)
    y = dl(dev(x).unsqueeze(1), dev(d).unsqueeze(1)).cpu().numpy()[:, 0, :]
    assert np.array_equal(y, yo) and np.array_equal(dl.buffer.cpu().numpy()[:, 0, :], bo)
    # d > D: AssertionError like code/model.py:284, state untouched
    before = dl.buffer.clone()
    d[2, 100] = D + 0.5
    with pytest.raises(AssertionError):
        dl(dev(x).unsqueeze(1), dev(d).unsqueeze(1))
    assert torch.equal(dl.buffer, before)


# ----------------------------------------------------------------------------- DiffDelGRU
@pytest.mark.parametrize("variant", VARIANTS)
def test_g5_diffdel_predict(ntm, variant):
    g = load("g5_diffdel_predict.npz")
    m = make_ddr(ntm, int(g["max_delay"]), str(g["weights"]), variant)
    y, pre = m.predict(dev(g["x"]), dev(g["d"]))
    assert m.diffdel.max_delay == int(g["D_effective"])
    assert np.abs(pre.cpu().numpy() - g["pre_d"]).max() < TOL
    assert np.abs(y.cpu().numpy() - g["y"]).max() < TOL
    assert np.abs(m.diffdel.buffer.cpu().numpy() - g["buffer"]).max() < TOL
    assert np.abs(m.hidden.cpu().numpy() - g["hidden"]).max() < TOL
    # batched predict: every stream equals the B=1 result
    xb, db = dev(np.repeat(g["x"], 3, 0)), dev(np.repeat(g["d"], 3, 0))
    yb, _ = m.predict(xb, db, segment_length=2048)
    for b in range(3):
        assert np.abs(yb[b].cpu().numpy() - g["y"][0]).max() < TOL


def test_g8_diffdel_batched_validate_style(ntm):
    g = load("g8_diffdel_batched.npz")
    m = make_ddr(ntm, int(g["max_delay"]), str(g["weights"]))
    x, d = dev(g["x"]), dev(g["d"])
    B, T = x.shape[0], x.shape[2]
    init, chunk = int(g["init_len"]), int(g["chunk"])
    m.initialize_hidden(B, m.max_delay)
    ys, ps = [], []
    y, p = m(x[:, :, :init], d[:, :, :init], warmup=True)
    ys.append(y); ps.append(p)
    for off in range(init, T, chunk):
        y, p = m(x[:, :, off:off + chunk], d[:, :, off:off + chunk])
        ys.append(y); ps.append(p)
    assert np.abs(torch.cat(ys, 2).cpu().numpy() - g["y"]).max() < TOL
    assert np.abs(torch.cat(ps, 2).cpu().numpy() - g["pre_d"]).max() < TOL
    assert np.abs(m.diffdel.buffer.cpu().numpy() - g["buffer"]).max() < TOL
    assert np.abs(m.hidden.cpu().numpy() - g["hidden"]).max() < TOL


# ----------------------------------------------------------------------------- ESR
def test_esr_sums_vs_oracle(ntm):
    rng = np.random.default_rng(21)
    B, T, skip = 9, 12345, 1024
    t = rng.standard_normal((B, T)).astype(np.float32)
    y = (t + 0.1 * rng.standard_normal((B, T))).astype(np.float32)
    s = ntm.esr_sums(dev(y).unsqueeze(1), dev(t).unsqueeze(1), skip).cpu().numpy()
    so = oracle.esr_sums(y, t, skip)
    assert np.allclose(s, so, rtol=1e-9)
    e = ntm.esr_per_segment(dev(y).unsqueeze(1), dev(t).unsqueeze(1), skip).cpu().numpy()
    assert np.allclose(e, oracle.esr_per_segment(y, t, skip), rtol=1e-9)
    tot = float(ntm.ESRLoss()(dev(y).unsqueeze(1), dev(t).unsqueeze(1)))
    s0 = oracle.esr_sums(y, t, 0).sum(0)
    assert abs(tot - (s0[0] / (B * T)) / (s0[1] / (B * T) + 1e-5)) < 1e-6


# ----------------------------------------------------------------------------- full-size properties
def test_full_batch_properties(ntm):
    """B = 4096 streams (BASELINE cfg 2 batch) at T = 4096: (i) first 16 streams == oracle,
    (ii) replicated inputs give bit-identical outputs wherever the stream sits in the grid,
    (iii) two half-length calls with carried state == one call, bit for bit."""
    B, T = 4096, 4096
    rng = np.random.default_rng(2024)
    base = rng.uniform(-0.5, 0.5, (64, T)).astype(np.float32)
    x = dev(np.tile(base, (B // 64, 1))).unsqueeze(1)
    m = make_rnn(ntm)
    y = m.predict(x)
    yo, _ = oracle.gru_predict(oracle_weights(W_G), base[:16], threads=4)
    assert np.abs(y[:16, 0].cpu().numpy() - yo).max() < TOL
    yv = y.view(B // 64, 64, T)
    assert torch.equal(yv, yv[:1].expand_as(yv))
    m.initialize_hidden(); m.warm_start(); m.hidden = m.hidden.expand(1, B, 64).contiguous()
    ya = m(x[:, :, :T // 2 + 13]); yb = m(x[:, :, T // 2 + 13:])
    assert torch.equal(torch.cat([ya, yb], 2), y)


@pytest.mark.skipif(os.environ.get("NTM_SKIP_FULL") == "1", reason="full 4096x65536 pass skipped on request")
def test_full_size_cfg2_checksum(ntm):
    """BASELINE cfg 2 at full size (4096 x 65536): stream-replication property over the whole grid
    plus the oracle on stream 0's last 4096 samples (state carried 61 440 steps)."""
    B, T = 4096, 65536
    g = load("g6_long_65536.npz")
    x1 = dev(g["x"][0])                                  # (1, 65536)
    x = x1.expand(B, T).contiguous().unsqueeze(1)
    m = make_rnn(ntm)
    y = m.predict(x)
    assert np.abs(y[0, 0].cpu().numpy() - g["y"][0, 0]).max() < TOL
    assert torch.equal(y, y[:1].expand_as(y))


@pytest.mark.skipif(os.environ.get("NTM_SKIP_FULL") == "1", reason="full 4096x65536 pass skipped on request")
def test_full_size_cfg3_diffdel(ntm):
    """BASELINE cfg 3 at full size (DiffDelGRU, 4096 x 65536, D = 1847): two distinct (signal, trajectory) pairs tiled
    over the batch must come back tiled (bit for bit, both outputs and both states), and equal the oracle."""
    B, T = 4096, 65536
    rng = np.random.default_rng(12)
    n = np.arange(T)
    x2 = rng.uniform(-0.5, 0.5, (2, T)).astype(np.float32)
    d2 = (44100 * (0.0271 + np.array([[0.004], [0.0025]]) * np.sin(2 * np.pi * np.array([[1.3], [0.7]]) * n / 44100))).astype(np.float32)
    m = ntm.harness.build_model(W_D, max_delay_seconds=0.0335)
    y, pre = m.predict(dev(np.tile(x2, (B // 2, 1))).unsqueeze(1), dev(np.tile(d2, (B // 2, 1))).unsqueeze(1))
    assert torch.equal(y[2:], y[:-2]) and torch.equal(pre[2:], pre[:-2])
    assert torch.equal(m.hidden[0, 2:], m.hidden[0, :-2]) and torch.equal(m.diffdel.buffer[2:], m.diffdel.buffer[:-2])
    yo, preo, _, _ = oracle.diffdel_predict(oracle_weights(W_D), x2, d2, m.max_delay)
    assert np.abs(pre[:2, 0].cpu().numpy() - preo).max() < TOL and np.abs(y[:2, 0].cpu().numpy() - yo).max() < TOL


@pytest.mark.skipif(os.environ.get("NTM_SKIP_FULL") == "1", reason="full 4096x65536 pass skipped on request")
def test_full_size_cfg4_tcn(ntm):
    """BASELINE cfg 4 at full size (TCN, 4096 x 65536, dilations 1/10/100/1000; 2 x 34 GB of activations in HBM): two
    distinct streams tiled over the batch come back tiled bit for bit and equal the oracle."""
    B, T = 4096, 65536
    rng = np.random.default_rng(13)
    x2 = rng.uniform(-0.8, 0.8, (2, T)).astype(np.float32)
    m = ntm.TCN().to("cuda")
    y = m(dev(np.tile(x2, (B // 2, 1))).unsqueeze(1))
    assert torch.equal(y[2:], y[:-2])
    yo = oracle.tcn_forward(m.packed_params().cpu().numpy(), len(m.dilations), m.channels, m.kernel_size, m.dilations, x2)
    assert np.abs(y[:2, 0].cpu().numpy() - yo).max() < TOL
    del y
    torch.cuda.empty_cache()


@pytest.mark.skipif(os.environ.get("NTM_SKIP_FULL") == "1", reason="full 4096x65536 pass skipped on request")
def test_full_size_loss_pass_properties(ntm):
    """The loss dict at BASELINE's full size (4096 x 65536) through size-independent properties: replicated streams
    give replicated per-segment values; the oracle on one stream; a common gain leaves ESR, DCPreESR and the
    spectral-convergence + log-magnitude loss unchanged (sums scale with the square of the gain)."""
    B, T, skip = 4096, 65536, 1024
    rng = np.random.default_rng(6)
    t1 = (0.3 * rng.standard_normal((3, T))).astype(np.float32)
    y1 = (t1 + 0.03 * rng.standard_normal((3, T))).astype(np.float32)
    reps = -(-B // 3)
    t = dev(np.tile(t1, (reps, 1))[:B]).unsqueeze(1)
    y = dev(np.tile(y1, (reps, 1))[:B]).unsqueeze(1)
    e, d = ntm.esr_sums(y, t, skip), ntm.esr_dcpre_sums(y, t, skip)
    st = ntm.MRSTFTLoss().per_segment(y, t, skip)
    for v in (e, d, st):
        assert torch.equal(v[3:], v[:-3])                                   # streams 0,1,2 repeat over the batch
    assert np.allclose(e[:3].cpu().numpy(), oracle.esr_sums(y1, t1, skip), rtol=1e-9)
    assert np.allclose(d[:3].cpu().numpy(), oracle.esr_dcpre_sums(y1, t1, skip), rtol=2e-5)
    assert np.allclose(st[:1].cpu().numpy(), oracle.mrstft_per_segment(y1[:1], t1[:1], skip), rtol=1e-4)
    e2, st2 = ntm.esr_sums(2 * y, 2 * t, skip), ntm.MRSTFTLoss().per_segment(2 * y, 2 * t, skip)
    assert torch.allclose(e2, 4 * e, rtol=1e-12) and torch.allclose(st2, st, rtol=1e-5)


# ----------------------------------------------------------------------------- harness (test-model.py loss loop)
def test_harness_compute_loss_gru_and_diffdel(ntm):
    """code/test-model.py:332-398 on in-memory segments: predict, cut INIT_LEN, per-segment ESR, mean."""
    g = load("g1_predict_16x8192.npz")
    rng = np.random.default_rng(31)
    x = g["x"]
    target = (g["y"] + 0.01 * rng.standard_normal(g["y"].shape)).astype(np.float32)
    m = ntm.harness.build_model(str(g["weights"]))
    res, out = ntm.harness.compute_loss(m, dev(x).unsqueeze(1), dev(target).unsqueeze(1), INIT_LEN=1024)
    yo, _ = oracle.gru_predict(oracle_weights(str(g["weights"])), x, threads=4)
    want = float(np.mean(oracle.esr_per_segment(yo, target, 1024)))
    assert res["segments"] == 16 and abs(res["ESR"] - want) < 1e-4 * want
    assert np.abs(out.cpu().numpy()[:, 0] - g["y"]).max() < TOL
    # DiffDelGRU through the same loop
    g5 = load("g5_diffdel_predict.npz")
    md = ntm.harness.build_model(str(g5["weights"]), max_delay_seconds=0.0335)
    assert md.max_delay == int(g5["max_delay"])
    tgt = dev(g5["y"])
    res, out = ntm.harness.compute_loss(md, dev(g5["x"]), tgt, d_traj=dev(g5["d"]), INIT_LEN=ntm.harness.init_len(0.0335))
    assert res["ESR"] < 1e-9 and res["segments"] == 1          # target == reference output
    with pytest.raises(SystemExit):
        ntm.harness.build_model("LSTM-HS[64]-L[ESR]-DS[x]")


# ----------------------------------------------------------------------------- TCN (builder-defined)
@pytest.mark.parametrize("B,T,dil", [(3, 700, (1, 3, 9, 27)), (2, 4000, (1, 10, 100, 1000)), (1, 1, (1, 10, 100, 1000)),
                                     (2, 900, (2, 5, 1, 3)), (2, 300, (1,)), (3, 1030, (1, 7)),
                                     # dilations >= 512 take the phase-group kernel: as inner blocks (activation written), as
                                     # the last one (output conv fused), with T not a multiple of anything
                                     (2, 5000, (1, 1000, 600, 10)), (1, 40001, (1, 3000, 512)), (2, 2047, (1, 700))])
def test_tcn_vs_oracle(ntm, B, T, dil):
    m = ntm.TCN(dilations=dil).to("cuda")
    rng = np.random.default_rng(B * 7 + T)
    x = rng.uniform(-0.8, 0.8, (B, T)).astype(np.float32)
    y = m(dev(x).unsqueeze(1)).cpu().numpy()[:, 0, :]
    yo = oracle.tcn_forward(m.packed_params().cpu().numpy(), len(dil), 32, 13, dil, x)
    assert np.abs(y - yo).max() < TOL
    # causality: changing the future does not change the past
    x2 = x.copy(); x2[:, T // 2:] += 1.0
    y2 = m(dev(x2).unsqueeze(1)).cpu().numpy()[:, 0, :]
    assert np.array_equal(y2[:, :T // 2], y[:, :T // 2])


def test_f16x3_engine_against_exact_fp32_on_hot_input(ntm):
    """The opt-in f16x3 GEMV engine on a loud, noisy input (|x| up to 0.95, 8192 steps): within the
    path's 1e-5 bar of the oracle, and as close to the exact-fp32 kernel as two exact-fp32 kernels
    with different summation order are to each other."""
    B, T = 512, 8192
    rng = np.random.default_rng(77)
    x = rng.uniform(-0.95, 0.95, (B, T)).astype(np.float32)
    xd = dev(x).unsqueeze(1)
    ys = {}
    for variant in ("mfma2", "f16x3", "mfma"):
        ys[variant] = make_rnn(ntm, W_G, variant).predict(xd)[:, 0]
    yo, _ = oracle.gru_predict(oracle_weights(W_G), x[:8], threads=4)
    assert np.abs(ys["f16x3"][:8].cpu().numpy() - yo).max() < TOL
    d_f16 = (ys["f16x3"] - ys["mfma2"]).abs().max().item()
    d_f32 = (ys["mfma"] - ys["mfma2"]).abs().max().item()
    assert d_f16 < TOL
    assert d_f16 < 3 * max(d_f32, 1e-6), (d_f16, d_f32)


def test_apply_delay_working_version(ntm):
    """`--ADD_DELAY` path of code/test-model.py:259-290 (broken in the reference: missing max_d)."""
    rng = np.random.default_rng(5)
    B, T, D = 3, 10000, 1846
    y = rng.standard_normal((B, T)).astype(np.float32)
    d = (900 + 800 * np.sin(np.arange(T) / 700.0))[None, :].repeat(B, 0).astype(np.float32)
    dl = ntm.TimeVaryingDelayLine(max_delay=D)
    a = ntm.harness.apply_delay(dl, dev(d).unsqueeze(1), dev(y).unsqueeze(1))
    b = ntm.harness.apply_delay(dl, dev(d).unsqueeze(1), dev(y).unsqueeze(1), segment_length=2**12)
    yo, _ = oracle.delay_forward(y, d, np.zeros((B, D), np.float32))
    assert np.array_equal(a.cpu().numpy()[:, 0], yo) and torch.equal(a, b)


def test_diagnostic_entry_points_run(ntm):
    """ntm_debug_gru_stamps / ntm_debug_gru_ablate stay callable (they back tools/stamp_profile.py and
    tools/ablate.py); the stamped build must still produce the right numbers."""
    L = ntm._lib.lab()
    m = make_rnn(ntm)
    B, T = 32, 256
    rng = np.random.default_rng(9)
    xh = rng.uniform(-0.5, 0.5, (B, T)).astype(np.float32)
    x = dev(xh)
    g, o = m.GRU, m.output
    P = lambda t: ctypes.c_void_p(t.data_ptr())       # noqa: E731
    yo, _ = oracle.gru_forward(oracle_weights(W_G), xh)
    for variant in (1, 3):
        y = torch.empty_like(x); h = torch.zeros(B, 64, device="cuda")
        st = torch.zeros(2, 4, 12, dtype=torch.int64, device="cuda")      # [..., :6] every step, [..., 6:] the phase-2 steps alone (MFMA2)
        rc = L.ntm_debug_gru_stamps(P(g.weight_ih_l0), P(g.weight_hh_l0), P(g.bias_ih_l0), P(g.bias_hh_l0), P(o.weight),
                                    P(o.bias), P(x), P(y), B, T, P(h), P(st), variant, None)
        assert rc == 0, L.ntm_last_error()
        torch.cuda.synchronize()
        assert np.abs(y.cpu().numpy() - yo).max() < TOL
        assert (st.cpu().numpy()[:, :, :6] > 0).all() and (variant == 1 or (st.cpu().numpy()[:, :, 6:] > 0).all())
    y = torch.empty_like(x); h = torch.zeros(B, 64, device="cuda")
    assert L.ntm_debug_gru_ablate(P(g.weight_ih_l0), P(g.weight_hh_l0), P(g.bias_ih_l0), P(g.bias_hh_l0), P(o.weight),
                                  P(o.bias), P(x), P(y), B, T, P(h), 4, None) == 0
    torch.cuda.synchronize()


def test_esr_dcpre_sums_vs_oracle(ntm):
    rng = np.random.default_rng(23)
    for B, T, skip in [(5, 20000, 1024), (2, 100, 0), (3, 4097, 7), (1, 16, 16)]:
        t = (rng.standard_normal((B, T)) + 0.3).astype(np.float32)       # with a DC offset
        y = (t + 0.1 * rng.standard_normal((B, T))).astype(np.float32)
        s = ntm.esr_dcpre_sums(dev(y).unsqueeze(1), dev(t).unsqueeze(1), skip).cpu().numpy()
        so = oracle.esr_dcpre_sums(y, t, skip)
        assert np.allclose(s, so, rtol=2e-5, atol=1e-12), (B, T, skip)
    # the whole-tensor loss object used like `loss_fcn(output, target)` in code/test-model.py:386-388
    tot = float(ntm.DCPreESR()(dev(y).unsqueeze(1), dev(t).unsqueeze(1)))
    s0 = oracle.esr_dcpre_sums(y, t, 0).sum(0)
    n = y.size
    assert abs(tot - (s0[0] / n) / (s0[1] / n + 1e-5)) < 1e-5 * tot


def test_cli_loss_over_synthetic_dataset(ntm, tmp_path):
    """tools/test_model.py (the --COMPUTE_LOSS path of code/test-model.py) on WAV files: feeder -> batched
    predict -> per-segment ESR / DCPreESR -> mean over segments, against the oracle."""
    import importlib.util
    from scipy.io import wavfile
    spec = importlib.util.spec_from_file_location("ntm_cli", os.path.join(os.path.dirname(os.path.dirname(__file__)),
                                                                          "tools", "test_model.py"))
    cli = importlib.util.module_from_spec(spec); spec.loader.exec_module(cli)
    d = tmp_path / "ToySet" / "Test"
    d.mkdir(parents=True)
    rng = np.random.default_rng(12)
    w = oracle_weights(W_G)
    L, segs = 4096, []
    for i in range(2):
        x = (rng.uniform(-0.5, 0.5, 3 * L + 100) * 32767).astype(np.int16)
        xf = x.astype(np.float32) / 32768
        for k in range(3):
            segs.append(xf[k * L:(k + 1) * L])
        # target = oracle output of the whole-file stream + a little noise, stored as float32 WAV
        tf = np.concatenate([oracle.gru_predict(w, s[None])[0][0] for s in segs[-3:]] + [np.zeros(100, np.float32)])
        tf = (tf + 0.01 * rng.standard_normal(tf.shape)).astype(np.float32)
        wavfile.write(str(d / f"input_{i}_.wav"), 44100, x)
        wavfile.write(str(d / f"target_{i}_.wav"), 44100, tf)
    f = ntm.feeder.SegmentFeeder(str(tmp_path / "ToySet"), "test", L)
    X = np.stack([f[i][0][0].numpy() for i in range(len(f))]); Tg = np.stack([f[i][1][0].numpy() for i in range(len(f))])
    yo, _ = oracle.gru_predict(w, X, threads=4)
    assert len(f) == 6
    # default INIT_LEN: the reference's nextpow2(int(0 * fs)) == 2 for a dataset without delay (code/test-model.py:323-324)
    for init, extra in ((2, []), (1024, ["--INIT_LEN", "1024"])):
        got = cli.main(["--DATASET_DIR", str(tmp_path / "ToySet"), "--SUBSET", "Test", "--WEIGHTS", W_G, "--SEGMENT_LENGTH", str(L),
                        "--BATCH_SIZE", "4", "--COMPUTE_LOSS", "--NO_SHUFFLE", "--TEMP_PATH", str(tmp_path / f"t{init}")] + extra)
        want = float(np.mean(oracle.esr_per_segment(yo, Tg, init)))
        sd = oracle.esr_dcpre_sums(yo, Tg, init); n = L - init
        want_dc = float(np.mean((sd[:, 0] / n) / (sd[:, 1] / n + 1e-5)))
        assert abs(got["ESR"] - want) < 1e-3 * want and abs(got["DCPreESR"] - want_dc) < 1e-3 * want_dc
        want_st = float(np.mean(oracle.mrstft_per_segment(yo, Tg, init)))
        assert abs(got["MultiSTFT"] - want_st) < 1e-3 * want_st


@pytest.mark.parametrize("variant", VARIANTS)
def test_saturating_inputs(ntm, variant):
    """|x| up to 100 drives every gate into saturation (2^x -> inf / 0 inside sigmoid and tanh): the fast
    exp2/rcp forms must still land on the oracle, with no NaN."""
    rng = np.random.default_rng(17)
    x = (rng.uniform(-1, 1, (6, 400)) * np.array([[1], [10], [100], [100], [1e-3], [0]])).astype(np.float32)
    yo, ho = oracle.gru_forward(oracle_weights(W_G), x)
    m = make_rnn(ntm, W_G, variant)
    m.initialize_hidden()
    y = m(dev(x).unsqueeze(1)).cpu().numpy()[:, 0]
    assert np.isfinite(y).all()
    assert np.abs(y - yo).max() < TOL
    assert np.abs(m.hidden.cpu().numpy()[0] - ho).max() < TOL


def test_g9_tape_hmag(ntm):
    """N4: Jiles-Atherton RK4 stage against the reference's own Tape.H_mag output (fp64), state carried."""
    g = load("g9_tape_hmag.npz")
    H, split = g["H"], int(g["split"])
    tp = ntm.TapeMagnetization(batch_size=H.shape[0])
    assert abs(tp.Ts_OS - float(g["Ts_OS"])) < 1e-18
    M = torch.cat([tp.H_mag(torch.from_numpy(H[:, :split]).cuda()), tp.H_mag(torch.from_numpy(H[:, split:]).cuda())], 1)
    scale = tp.TAPE_Ms
    assert np.abs(M.cpu().numpy() - g["M"]).max() < 1e-6 * scale
    assert np.abs(tp.M_prev.cpu().numpy() - g["M_prev"]).max() < 1e-6 * scale
    assert np.allclose(tp.H_prev.cpu().numpy(), g["H_prev"]) and np.allclose(tp.Hprime_prev.cpu().numpy(), g["Hprime_prev"])
    # a batch that does not fill a 64-stream block, against the oracle
    rng = np.random.default_rng(1)
    H2 = 8000.0 * rng.standard_normal((70, 300)).cumsum(1) / 20
    Mo, _ = oracle.tape_hmag(H2)
    tp2 = ntm.TapeMagnetization(batch_size=70)
    assert np.abs(tp2.H_mag(torch.from_numpy(H2).cuda()).cpu().numpy() - Mo).max() < 1e-6 * scale


STFT_RES = [(1024, 120, 600), (2048, 240, 1200), (512, 50, 240), (256, 64, 256), (512, 128, 500)]


@pytest.mark.parametrize("res", STFT_RES)
def test_stft_sums_vs_oracle(ntm, res):
    """N1: one STFT resolution (FFT kernel, fp32) against the fp64 oracle on the g10 pairs; the last two
    resolutions are not auraloss defaults (n_fft 256; an even window that is not centred on a 64-lane boundary)."""
    g = load("g10_mrstft.npz")
    n_fft, hop, win = res
    for skip in (int(g["skip"]), 0, 8192 - n_fft // 2 - 1):
        s, cells = ntm.stft_sums(dev(g["pred"]).unsqueeze(1), dev(g["targ"]).unsqueeze(1), skip, n_fft, hop, win)
        so, cells_o = oracle.stft_sums(g["pred"], g["targ"], skip, n_fft, hop, win)
        assert cells == cells_o
        assert np.allclose(s.cpu().numpy(), so, rtol=1e-4), (res, skip, s.cpu().numpy() / so - 1)


def test_g10_mrstft_loss(ntm):
    """MultiResolutionSTFTLoss() per segment against torch.stft + the auraloss formula (golden g10), and the
    whole-batch form against the oracle's sums."""
    g = load("g10_mrstft.npz")
    skip = int(g["skip"])
    y, t = dev(g["pred"]).unsqueeze(1), dev(g["targ"]).unsqueeze(1)
    loss = ntm.MRSTFTLoss()
    per = loss.per_segment(y, t, skip).cpu().numpy()
    assert np.allclose(per, g["loss"], rtol=1e-4), per / g["loss"] - 1
    whole, cells_all = 0.0, 0
    for n_fft, hop, win in oracle.MRSTFT_RESOLUTIONS:
        so, cells = oracle.stft_sums(g["pred"], g["targ"], 0, n_fft, hop, win)
        so = so.sum(0)
        whole += np.sqrt(so[0] / so[1]) + so[2] / (cells * len(g["pred"]))
    assert abs(float(loss(y, t)) - whole / 3) < 1e-4 * whole / 3
    # the linear-magnitude weight (off by default upstream)
    lin = ntm.MRSTFTLoss(w_sc=0.0, w_log_mag=0.0, w_lin_mag=1.0).per_segment(y, t, skip).cpu().numpy()
    assert np.allclose(lin, g["terms"][:, :, 2].mean(1), rtol=1e-4)


def test_stft_sums_many_streams_chunking(ntm):
    """Chunked launches (frames of a stream split over workgroups) and single-chunk launches give the same
    sums; identical signals give (numerically) zero distance; silence hits the eps clamp on both sides."""
    rng = np.random.default_rng(5)
    B, T = 3, 40000
    t = (0.3 * rng.standard_normal((B, T))).astype(np.float32)
    y = (t + 0.01 * rng.standard_normal((B, T))).astype(np.float32)
    y[2] = t[2]
    t[1, 10000:30000] = 0.0; y[1, 10000:30000] = 0.0
    s, cells = ntm.stft_sums(dev(y).unsqueeze(1), dev(t).unsqueeze(1), 100)          # B = 3 -> many chunks
    so, _ = oracle.stft_sums(y, t, 100)
    assert np.allclose(s.cpu().numpy()[:2], so[:2], rtol=1e-4)
    # identical signals: the pair shares one complex FFT, so the distance is rounding noise rather than exactly 0
    s2 = s.cpu().numpy()[2]
    assert s2[0] / s2[1] < 1e-12 and s2[2] / cells < 1e-5 and abs(s2[1] / so[2, 1] - 1) < 1e-4
    yy, tt = np.tile(y, (1000, 1)), np.tile(t, (1000, 1))                             # B = 3000 -> one chunk each
    s1, _ = ntm.stft_sums(dev(yy).unsqueeze(1), dev(tt).unsqueeze(1), 100)
    assert np.allclose(s1.cpu().numpy()[:3], s.cpu().numpy(), rtol=1e-9)
    assert (s1.cpu().numpy()[3:6] == s1.cpu().numpy()[:3]).all()


def test_stft_sums_errors(ntm):
    y = dev(np.zeros((2, 4096), np.float32)).unsqueeze(1)
    with pytest.raises(ntm.NtmError):
        ntm.stft_sums(y, y, 0, 1000, 120, 600)            # n_fft not a supported power of two
    with pytest.raises(ntm.NtmError):
        ntm.stft_sums(y, y, 4096 - 512, 1024, 120, 600)   # T - skip <= n_fft/2: reflect padding impossible
    with pytest.raises(ntm.NtmError):
        ntm.stft_sums(y, y, 0, 1024, 120, 1025)           # window longer than the frame


def test_g11_demodulate(ntm):
    """N3: the device demodulation against the reference's own DelayAnalyzer.demodulate output (fp64 there,
    rounded to fp32 here): positive, zero and negative input/output offsets, truncated pulse trains."""
    from ntm_amd.feeder import demodulate
    g = load("g11_demodulate.npz")
    for i in range(int(g["n"])):
        dem = demodulate(dev(g[f"out{i}"]), g[f"x{i}"], g[f"y{i}"]).cpu().numpy()
        want = g[f"dem{i}"]
        assert dem.shape == want.shape
        assert np.array_equal(dem, want.astype(np.float32)), (i, np.abs(dem - want).max())
    with pytest.raises(ntm.NtmError):
        demodulate(dev(g["out0"]), np.array([5, 5]), g["y0"])            # period 0
    with pytest.raises(AssertionError):
        demodulate(dev(g["out0"]), g["x0"], g["y0"][:1])                 # a single pulse


def test_feeder_demodulated_targets(ntm, tmp_path):
    """Feeder with demodulate=True (code/dataset.py:395-408): target demodulated on the device, mean delay cut."""
    from scipy.io import wavfile
    from ntm_amd.feeder import SegmentFeeder, segment_peaks
    g = load("g11_demodulate.npz")
    d = tmp_path / "Set" / "Test"
    d.mkdir(parents=True)
    out, x_idx, y_idx = g["out0"], g["x0"], g["y0"]
    N = out.shape[1]
    wavfile.write(str(d / "input_1_.wav"), 44100, np.stack([out[0], np.zeros(N, np.float32)], 1))
    wavfile.write(str(d / "target_1_.wav"), 44100, out.T.copy())
    traj = np.full(N, 1200 / 44100.0)
    np.save(str(d / "trajectory_1_.npy"), {"input_peaks": x_idx, "input_meta": {}, "output_peaks": y_idx, "output_meta": {},
                                           "delay_trajectory": traj})
    f = SegmentFeeder(str(tmp_path / "Set"), subset="test", length=14000, demodulate=True)
    assert len(f) == 2
    cut = int(f.mean_delay * 44100)
    for k in range(2):
        x, t, meta = f[k]
        o = 14000 * k
        pi, po = segment_peaks(x_idx, y_idx, o, o + 14000, 14000)
        want = oracle.demodulate(out[:, o:o + 14000], pi, po)[:, :-cut]
        assert t.shape == (2, 14000 - cut) and x.shape == (2, 14000 - cut)
        assert np.array_equal(t.numpy(), want.astype(np.float32))
        assert len(meta["delay_trajectory"]) == 14000 - cut and np.array_equal(meta["output_peaks"], meta["input_peaks"])


def test_abi_alignment_checks(ntm):
    """Kernels that use 16-byte loads on caller memory refuse misaligned pointers instead of faulting."""
    import ctypes
    L = ntm._lib.lib()
    buf = torch.zeros(200000, device="cuda")
    p = lambda off: ctypes.c_void_p(buf.data_ptr() + 4 * off)       # noqa: E731
    dil = (ctypes.c_int * 1)(1)
    rc = L.ntm_tcn_forward(p(1), 1, 32, 13, dil, p(4096), p(8192), 1, 64, p(16384), None)
    assert rc != 0 and b"aligned" in L.ntm_last_error()
    LAB = ntm._lib.lab()
    rc = LAB.ntm_lab_gru_forward(p(0), p(1025), p(256), p(512), p(768), None, 64, p(20000), p(30000), 2, 16, 16, 16, None,
                                 ntm._lib.VARIANTS["valu"], None)
    assert rc != 0 and b"aligned" in LAB.ntm_lab_last_error()


@pytest.mark.parametrize("resident", [False, True])
def test_streamed_predict_from_pinned_host(ntm, tmp_path, resident):
    """N2: time-pipelined predict straight from the feeder's pinned files (pitched DMA chunks on a side stream, the
    GRU launch per chunk on the compute stream, state carried) == one resident launch, bit for bit; input and
    target arrive intact; runs of consecutive segments are found across file boundaries."""
    from scipy.io import wavfile
    from ntm_amd.feeder import SegmentFeeder
    d = tmp_path / "Set" / "Test"
    d.mkdir(parents=True)
    rng = np.random.default_rng(3)
    L = 3000
    for i, nseg in enumerate((3, 1, 4)):
        x = (rng.uniform(-0.5, 0.5, nseg * L + 17 * i) * 32767).astype(np.int16)
        wavfile.write(str(d / f"input_{i}_.wav"), 44100, x)
        wavfile.write(str(d / f"target_{i}_.wav"), 44100, (0.5 * x).astype(np.int16))
    f = SegmentFeeder(str(tmp_path / "Set"), subset="test", length=L, resident=resident)     # round 5: resident = the set on the device
    assert f.resident is resident
    assert len(f) == 8 and [r[:2] for r in f.runs(1, 8)] == [(0, 2), (2, 1), (3, 4)]
    m = make_rnn(ntm, W_G, "mfma2")
    xin, tgt, _, _ = next(f.batches(8, "cuda"))
    want = m.predict(xin)
    for chunk in (1000, 700, 3000, 8192):
        y, x2, t2 = f.predict_streamed(m, 0, 8, chunk=chunk)
        torch.cuda.synchronize()
        assert torch.equal(x2, xin) and torch.equal(t2, tgt), chunk
        assert torch.equal(y, want), chunk
    host = torch.empty(8, 1, L).pin_memory()
    y, _, _ = f.predict_streamed(m, 0, 8, chunk=900, out_host=host)          # results copied back as they are produced
    torch.cuda.synchronize()
    assert torch.equal(host, want.cpu())
    y, x2, _ = f.predict_streamed(m, 2, 7, chunk=512)          # a sub-range that starts inside a file
    assert torch.equal(x2, xin[2:7]) and torch.equal(y, m.predict(xin[2:7]))


def test_random_shapes_all_exact_kernels_agree_with_oracle(ntm):
    """Seeded sweep over ragged shapes (tile edges of every kernel: 16 streams per MFMA group, 64- and 256-sample
    tiles, 2048-sample chunks) with a carried random state: every exact-fp32 kernel against the oracle."""
    rng = np.random.default_rng(2024)
    w = oracle_weights(W_G)
    shapes = [(1, 1), (1, 63), (1, 257), (2, 64), (3, 65), (15, 129), (16, 256), (17, 255), (31, 511), (33, 513),
              (64, 100), (100, 7), (129, 300), (255, 66), (5, 2049), (1025, 70)]
    for B, T in shapes:
        x = rng.uniform(-0.6, 0.6, (B, T)).astype(np.float32)
        h0 = rng.uniform(-0.9, 0.9, (B, 64)).astype(np.float32)
        yo, ho = oracle.gru_forward(w, x, h0.copy(), threads=4)
        for variant in ("auto", "lat", "mfma2", "mfma", "valu"):
            m = make_rnn(ntm, W_G, variant)
            m.hidden = dev(h0).unsqueeze(0)
            y = m(dev(x).unsqueeze(1)).cpu().numpy()[:, 0]
            assert np.abs(y - yo).max() < TOL, (variant, B, T)
            assert np.abs(m.hidden.cpu().numpy()[0] - ho).max() < TOL, (variant, B, T)


def test_cli_diffdel_on_raw_stereo_dataset(ntm, tmp_path):
    """The whole evaluation path for DiffDelGRU from a RAW stereo dataset (audio + pilot pulses, no side-cars):
    pulse analysis -> side-car -> INIT_LEN and delay-line length from the dataset's max delay
    (code/test-model.py:223,323-324) -> batched predict with the per-sample trajectory -> losses, against the oracle."""
    import importlib.util
    from scipy.io import wavfile
    spec = importlib.util.spec_from_file_location("ntm_cli2", os.path.join(os.path.dirname(os.path.dirname(__file__)),
                                                                           "tools", "test_model.py"))
    cli = importlib.util.module_from_spec(spec); spec.loader.exec_module(cli)
    g = load("g12_delay_analysis.npz")
    fs, N = int(g["fs"]), len(g["in0"])
    d = tmp_path / "Wow" / "Test"
    d.mkdir(parents=True)
    rng = np.random.default_rng(21)
    audio = rng.uniform(-0.4, 0.4, N).astype(np.float32)
    tgt_audio = (0.3 * np.roll(audio, 1200) + 0.01 * rng.standard_normal(N)).astype(np.float32)
    wavfile.write(str(d / "input_0_.wav"), fs, np.stack([audio, g["in0"]], 1))
    wavfile.write(str(d / "target_0_.wav"), fs, np.stack([tgt_audio, g["out0"]], 1))
    L = 11000
    got = cli.main(["--MODEL", "DiffDelGRU", "--DATASET_DIR", str(tmp_path / "Wow"), "--SUBSET", "Test", "--NO_SHUFFLE", "--WEIGHTS", W_D,
                    "--SEGMENT_LENGTH", str(L), "--COMPUTE_LOSS", "--TEMP_PATH", str(tmp_path / "tmp")])
    # expected, from the reference's own analysis result (golden g12) and the oracle
    T = g["T0"]
    max_delay_n = int(1.25 * T.max() * fs)
    init = 1 << (int(T.max() * fs) - 1).bit_length()
    wd = oracle_weights(W_D)
    nseg = N // L
    X = np.stack([audio[k * L:(k + 1) * L] for k in range(nseg)])
    D = np.stack([(T[k * L:(k + 1) * L]).astype(np.float32) * np.float32(fs) for k in range(nseg)])
    yo, _, _, _ = oracle.diffdel_predict(wd, X, D, max_delay_n)
    Tg = np.stack([tgt_audio[k * L:(k + 1) * L] for k in range(nseg)])
    want = float(np.mean(oracle.esr_per_segment(yo, Tg, init)))
    assert init == 2048 and abs(got["ESR"] - want) < 1e-3 * want, (got, want)


def test_cli_add_delay_mode(ntm, tmp_path):
    """`--ADD_DELAY` with a GRU model (code/test-model.py:236-240, 355-364; the reference's own helper raises
    TypeError): the measured trajectory is applied to the model output before the loss."""
    import importlib.util
    from scipy.io import wavfile
    spec = importlib.util.spec_from_file_location("ntm_cli3", os.path.join(os.path.dirname(os.path.dirname(__file__)),
                                                                           "tools", "test_model.py"))
    cli = importlib.util.module_from_spec(spec); spec.loader.exec_module(cli)
    g = load("g12_delay_analysis.npz")
    fs, N = int(g["fs"]), len(g["in1"])
    d = tmp_path / "Wow" / "Test"
    d.mkdir(parents=True)
    rng = np.random.default_rng(4)
    audio = rng.uniform(-0.4, 0.4, N).astype(np.float32)
    tgt_audio = (0.2 * np.roll(audio, 1200)).astype(np.float32)
    wavfile.write(str(d / "input_0_.wav"), fs, np.stack([audio, g["in1"]], 1))
    wavfile.write(str(d / "target_0_.wav"), fs, np.stack([tgt_audio, g["out1"]], 1))
    L = 14000
    got = cli.main(["--DATASET_DIR", str(tmp_path / "Wow"), "--SUBSET", "Test", "--NO_SHUFFLE", "--WEIGHTS", W_G, "--SEGMENT_LENGTH", str(L),
                    "--COMPUTE_LOSS", "--ADD_DELAY", "--TEMP_PATH", str(tmp_path / "tmp")])
    T = g["T1"]
    D = int(1.25 * T.max() * fs)
    init = 1 << (int(T.max() * fs) - 1).bit_length()
    w = oracle_weights(W_G)
    nseg = N // L
    X = np.stack([audio[k * L:(k + 1) * L] for k in range(nseg)])
    Dt = np.stack([(T[k * L:(k + 1) * L]).astype(np.float32) * np.float32(fs) for k in range(nseg)])
    yo, _ = oracle.gru_predict(w, X)
    yd, _ = oracle.delay_forward(yo, Dt, np.zeros((nseg, D), np.float32))
    Tg = np.stack([tgt_audio[k * L:(k + 1) * L] for k in range(nseg)])
    want = float(np.mean(oracle.esr_per_segment(yd, Tg, init)))
    assert abs(got["ESR"] - want) < 1e-3 * want, (got, want)


@pytest.mark.parametrize("B,D,chunks", [(3, 8900, (2048, 2048, 2048, 857)), (2, 1, (5, 1, 7)), (4, 37, (10, 37, 36, 38, 100)),
                                        (1, 8900, (12000,)), (130, 300, (299, 301))])
def test_delay_line_sizes_and_chunking_bit_exact(ntm, B, D, chunks):
    """Delay line at the real-data buffer length (D = 8900), degenerate D = 1, chunks shorter / equal / longer than
    the buffer, state carried: bit-exact against the oracle (itself bit-exact against the reference, golden g4)."""
    rng = np.random.default_rng(D + B)
    dl = ntm.TimeVaryingDelayLine(max_delay=D)
    dl.init_buffer(B, D)
    buf = np.zeros((B, D), np.float32)
    for T in chunks:
        x = rng.standard_normal((B, T)).astype(np.float32)
        d = rng.uniform(0, D, (B, T)).astype(np.float32)
        d[:, ::7] = np.floor(d[:, ::7])                       # integer delays
        d[0, :min(T, 3)] = [D, 0.0, D - 0.25][:min(T, 3)]      # the bounds
        yo, buf = oracle.delay_forward(x, d, buf)
        y = dl(dev(x).unsqueeze(1), dev(d).unsqueeze(1)).cpu().numpy()[:, 0, :]
        assert np.array_equal(y, yo), (D, T)
        assert np.array_equal(dl.buffer.cpu().numpy()[:, 0, :], buf), (D, T)


@pytest.mark.parametrize("B,block,use_graph", [(16, 128, True), (3, 64, True), (1, 512, False)])
def test_block_streamer_graph_replay(ntm, B, block, use_graph):
    """Real-time style blocks through a captured HIP graph == predict() on the concatenated signal."""
    rng = np.random.default_rng(B + block)
    nblk = 9
    x = rng.uniform(-0.5, 0.5, (B, nblk * block)).astype(np.float32)
    m = make_rnn(ntm, W_G, "auto")
    want = m.predict(dev(x).unsqueeze(1))
    s = ntm.harness.BlockStreamer(m, B, block, use_graph=use_graph)
    got = torch.cat([s.process(dev(x[:, k * block:(k + 1) * block]).unsqueeze(1)).clone() for k in range(nblk)], 2)
    assert torch.equal(got, want)


def test_c_abi_from_a_plain_cpp_process(ntm, tmp_path):
    """The boundary really is a C ABI: tools/cabi/cabi_demo (C++, raw HIP allocations, no torch, no Python) calls
    ntm_gru_forward on the raw weight file and gets the numbers of the Python layer / the oracle."""
    import subprocess
    root = os.path.dirname(os.path.dirname(__file__))
    exe = os.path.join(root, "tools", "cabi", "cabi_demo.bin")
    if not os.path.exists(exe):
        subprocess.run(["make", "-s", "-C", os.path.dirname(exe)], check=True)
    rng = np.random.default_rng(8)
    B, T = 19, 700
    x = rng.uniform(-0.5, 0.5, (B, T)).astype(np.float32)
    x.tofile(str(tmp_path / "x.f32"))
    wfile = os.path.join(root, "neural-tape-modeling_amd", "weights", "w0.bin")
    r = subprocess.run([exe, wfile, str(tmp_path / "x.f32"), str(B), str(T), str(tmp_path / "y.f32"), str(tmp_path / "h.f32")],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "hidden size must lie in [1, 1024]" in r.stdout        # the refusal message of the error-path call
    y = np.fromfile(str(tmp_path / "y.f32"), np.float32).reshape(B, T)
    h = np.fromfile(str(tmp_path / "h.f32"), np.float32).reshape(B, 64)
    yo, ho = oracle.gru_forward(oracle_weights(W_G), x)
    assert np.abs(y - yo).max() < TOL and np.abs(h - ho).max() < TOL
    m = make_rnn(ntm, W_G, "auto")
    m.initialize_hidden()
    assert np.array_equal(m(dev(x).unsqueeze(1)).cpu().numpy()[:, 0], y)      # same kernel, same bits


@pytest.mark.parametrize("n_fft", [64, 128, 256, 2048])
def test_spec_sums_vs_oracle(ntm, n_fft):
    """Power-spectrogram sums (code/evaluation.py:75-84) incl. the small scales, where a wave transforms 4 (n_fft 64)
    or 2 (n_fft 128) frames side by side; and the auraloss-style sums at those small sizes."""
    g = load("g10_mrstft.npz")
    y, t = dev(g["pred"]).unsqueeze(1), dev(g["targ"]).unsqueeze(1)
    for skip in (0, 777):
        s, cells = ntm.spec_sums(y, t, skip, n_fft)
        so, cells_o = oracle.spec_sums(g["pred"], g["targ"], skip, n_fft)
        assert cells == cells_o and np.allclose(s.cpu().numpy(), so, rtol=1e-4), (n_fft, skip, s.cpu().numpy() / so - 1)
    if n_fft <= 128:
        s, cells = ntm.stft_sums(y, t, 100, n_fft, n_fft // 4, n_fft - 4)
        so, _ = oracle.stft_sums(g["pred"], g["targ"], 100, n_fft, n_fft // 4, n_fft - 4)
        assert np.allclose(s.cpu().numpy(), so, rtol=1e-4)


def test_val_loss_supervised_bundle(ntm):
    """The validation metric bundle of code/evaluation.py:18-100 (without its mel entries) against golden g13
    (torch.stft) and the oracle."""
    g10, g = load("g10_mrstft.npz"), load("g13_ms_spec.npz")
    y, t = dev(g10["pred"]), dev(g10["targ"])
    got = ntm.ValLossSupervised()(y, t)
    assert abs(got["ms_spec_loss"] / float(g["ms_spec_loss"]) - 1) < 1e-4
    assert abs(got["ms_log_spec_loss"] / float(g["ms_log_spec_loss"]) - 1) < 1e-4
    e = oracle.esr_sums(g10["pred"], g10["targ"]).sum(0); n = g10["pred"].size
    assert abs(got["MSE"] - e[0] / n) < 1e-9 * e[0] / n and abs(got["ESR"] - (e[0] / n) / (e[1] / n + 1e-5)) < 1e-9
    d = oracle.esr_dcpre_sums(g10["pred"], g10["targ"]).sum(0)
    assert abs(got["ESRDCPre"] - (d[0] / n) / (d[1] / n + 1e-5)) < 1e-4 * got["ESRDCPre"]
    # the two mel entries (code/evaluation.py:86-92) against the oracle (librosa's filter bank restated: unpinned)
    m, cells = oracle.mel_sums(g10["pred"], g10["targ"])
    assert abs(got["mel_spec_loss"] / (m[:, 0].sum() / (cells * len(m))) - 1) < 1e-4
    assert abs(got["log_mel_spec_loss"] / (m[:, 1].sum() / (cells * len(m))) - 1) < 1e-4
    assert set(got) == {"ms_spec_loss", "ms_log_spec_loss", "mel_spec_loss", "log_mel_spec_loss", "ESR", "MSE", "ESRDCPre"}
    for skip, n_fft in ((0, 2048), (333, 2048), (0, 1024)):
        s, c = ntm.mel_sums(y.unsqueeze(1), t.unsqueeze(1), skip, n_fft)
        so, co = oracle.mel_sums(g10["pred"], g10["targ"], skip, n_fft)
        assert c == co and np.allclose(s.cpu().numpy(), so, rtol=1e-4), (skip, n_fft, s.cpu().numpy() / so - 1)


@pytest.mark.parametrize("resident", [False, True])
def test_streamed_predict_diffdel(ntm, tmp_path, resident):
    """Time-pipelined predict for DiffDelGRU (audio + delay trajectory sent chunk by chunk, GRU and delay-line state
    carried) == the resident one-shot predict, bit for bit."""
    from scipy.io import wavfile
    from ntm_amd.feeder import SegmentFeeder
    g = load("g12_delay_analysis.npz")
    fs, N = int(g["fs"]), len(g["in0"])
    d = tmp_path / "Wow" / "Test"
    d.mkdir(parents=True)
    rng = np.random.default_rng(9)
    audio = rng.uniform(-0.4, 0.4, N).astype(np.float32)
    wavfile.write(str(d / "input_0_.wav"), fs, np.stack([audio, g["in0"]], 1))
    wavfile.write(str(d / "target_0_.wav"), fs, np.stack([0.5 * audio, g["out0"]], 1))
    L = 7000
    f = SegmentFeeder(str(tmp_path / "Wow"), subset="test", length=L, resident=resident)
    m = ntm.harness.build_model(W_D, max_delay_seconds=f.max_delay, fs=fs)
    xin, tgt, dt, _ = next(f.batches(len(f), "cuda"))
    want, _ = m.predict(xin, dt * fs)
    for chunk in (2048, 1500, 7000):
        y, x2, t2 = f.predict_streamed(m, 0, len(f), chunk=chunk)
        torch.cuda.synchronize()
        assert torch.equal(x2, xin) and torch.equal(y, want), chunk


def test_auto_splits_a_ragged_batch_between_the_two_kernels(ntm):
    """B = one full device round (16 streams x CUs) + a small remainder: AUTO runs the matrix-pipe kernel on the full
    round and the low-latency kernel on the remainder; every stream (state included) matches the oracle."""
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    B, T = 16 * cus + 21, 96
    rng = np.random.default_rng(1)
    x = rng.uniform(-0.5, 0.5, (B, T)).astype(np.float32)
    h0 = rng.uniform(-0.5, 0.5, (B, 64)).astype(np.float32)
    idx = np.r_[0:3, 16 * cus - 2:16 * cus + 21]
    yo, ho = oracle.gru_forward(oracle_weights(W_G), x[idx], h0[idx].copy())
    m = make_rnn(ntm, W_G, "auto")
    m.hidden = dev(h0).unsqueeze(0)
    y = m(dev(x).unsqueeze(1)).cpu().numpy()[:, 0]
    assert np.abs(y[idx] - yo).max() < TOL and np.abs(m.hidden.cpu().numpy()[0][idx] - ho).max() < TOL
    # the two halves really come from different kernels: they agree with the explicit variants bit for bit
    for variant, sl in (("mfma2", slice(0, 16 * cus)), ("lat", slice(16 * cus, B))):
        mv = make_rnn(ntm, W_G, variant)
        mv.hidden = dev(h0[sl]).unsqueeze(0)
        assert np.array_equal(mv(dev(x[sl]).unsqueeze(1)).cpu().numpy()[:, 0], y[sl]), variant


def test_g14_tape_record_field_and_chain(ntm):
    """bias + H_rec on the device (ntm_tape_record_field) bit-identical to the reference's H over two stateful calls;
    then the chain record_field -> H_mag against the oracle's magnetisation of the reference's own H."""
    g = load("g14_tape_stages.npz")
    sp = int(g["split"])
    tp = ntm.TapeMagnetization(batch_size=2)
    H1 = tp.record_field(dev(g["I_in"][:, :sp]))
    H2 = tp.record_field(dev(g["I_in"][:, sp:]))
    H = torch.cat([H1, H2], 1)
    assert np.array_equal(H.cpu().numpy(), g["H"])
    M = tp.H_mag(H[:, :3000]).cpu().numpy()
    Mo, _ = oracle.tape_hmag(g["H"][:, :3000], None, tp.Ts_OS)
    assert np.abs(M - Mo).max() < 1e-6 * tp.TAPE_Ms
    tp2 = ntm.TapeMagnetization(batch_size=2, bias_enable=False)
    assert np.array_equal(tp2.record_field(dev(g["I_in"][:, :100])).cpu().numpy(), (10.0 * g["I_in"][:, :100]) / 6e-6)


def test_concurrent_callers_on_two_streams(ntm):
    """include/ntm.h promises re-entrant entry points (no global state, thread-local error text): two host threads,
    each with its own model object and HIP stream, run different batches at the same time and get the numbers of
    the serial runs; a failing call on one thread does not leak its message to the other."""
    import threading
    rng = np.random.default_rng(33)
    xs = [rng.uniform(-0.5, 0.5, (B, 3000)).astype(np.float32) for B in (40, 1200)]     # low-latency and matrix-pipe kernels
    want = []
    for x in xs:
        m = make_rnn(ntm, W_G, "auto")
        want.append(m.predict(dev(x).unsqueeze(1)).clone())
    got, errs = [None, None], [None, None]

    def worker(i):
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                m = make_rnn(ntm, W_G, "auto")
                for _ in range(3):
                    y = m.predict(dev(xs[i]).unsqueeze(1))
                if i == 0:
                    rc = ntm._lib.lib().ntm_stft_sums(None, None, 1, 100, 0, 1000, 1, 1, 1e-8, 1, None, None)
                    assert rc != 0 and b"n_fft" in ntm._lib.lib().ntm_last_error()
                else:
                    assert b"n_fft" not in ntm._lib.lib().ntm_last_error()
                s.synchronize()
            got[i] = y
        except Exception as e:          # noqa: BLE001
            errs[i] = e

    th = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert errs == [None, None], errs
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
