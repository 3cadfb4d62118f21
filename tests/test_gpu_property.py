"""Property-style parity tests (hypothesis): the HIP path against the oracle on RANDOM shapes, chunkings, delays and
dilations -- the cases nobody thought of writing down.  Integer / interpolation work bit for bit, the fp32 paths within
the path's 1e-5.  All through the Python host layer, i.e. through the C ABI.  Example counts are small: every example
is one or more kernel launches plus an oracle evaluation."""
import numpy as np
import pytest
import torch
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

import oracle

pytestmark = pytest.mark.gpu
TOL = 1e-5
SET = dict(deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)


@pytest.fixture(scope="module")
def ntm():
    import ntm_amd
    assert torch.cuda.is_available()
    return ntm_amd


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


# ----------------------------------------------------------------------------- K2: time-varying delay line, bit-exact
@settings(max_examples=120, **SET)
@given(B=st.integers(1, 5), T=st.integers(1, 300), D=st.integers(1, 70), seed=st.integers(0, 2**31 - 1),
       cuts=st.lists(st.integers(1, 299), max_size=3), integer_delays=st.booleans())
def test_delay_line_random_shapes_and_chunkings_bit_exact(ntm, B, T, D, seed, cuts, integer_delays):
    """code/model.py:269-320 in closed form: any batch, length (also T < D), buffer length, chunking; delays anywhere in
    [-0.9, D] including exact integers and d = D: output and carried buffer equal the oracle's bit for bit."""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((B, T)).astype(np.float32)
    d = rng.uniform(-0.9, D, (B, T)).astype(np.float32)
    if integer_delays:
        d = np.round(d).clip(0, D).astype(np.float32)
    d[:, rng.integers(0, T)] = D
    buf0 = rng.standard_normal((B, D)).astype(np.float32)
    yo, bo = oracle.delay_forward(x, d, buf0)
    dl = ntm.TimeVaryingDelayLine(max_delay=D)
    dl.init_buffer(B, D)
    dl.buffer = dev(buf0).view(B, 1, D)
    edges = [0] + sorted({c for c in cuts if c < T}) + [T]
    ys = [dl(dev(x[:, a:b]).unsqueeze(1), dev(d[:, a:b]).unsqueeze(1)) for a, b in zip(edges, edges[1:])]
    y = torch.cat(ys, dim=2)[:, 0].cpu().numpy()
    assert np.array_equal(y, yo)
    assert np.array_equal(dl.buffer[:, 0].cpu().numpy(), bo)


# ----------------------------------------------------------------------------- K1: GRU + head, all compiled hidden sizes
@settings(max_examples=60, **SET)
@given(H=st.sampled_from([8, 16, 32, 64]), B=st.integers(1, 40), T=st.integers(1, 200), seed=st.integers(0, 2**31 - 1),
       cuts=st.lists(st.integers(1, 199), max_size=2))
def test_gru_random_shapes_hidden_sizes_and_chunkings(ntm, H, B, T, seed, cuts):
    """RNN.forward (code/model.py:67-88) with random weights of every compiled hidden size: within 1e-5 of the oracle,
    carried state included, and a chunked run equals the one-shot run bit for bit."""
    rng = np.random.default_rng(seed)
    k = 1.0 / np.sqrt(H)
    sd = {"GRU.weight_ih_l0": rng.uniform(-k, k, (3 * H, 1)), "GRU.weight_hh_l0": rng.uniform(-k, k, (3 * H, H)),
          "GRU.bias_ih_l0": rng.uniform(-k, k, 3 * H), "GRU.bias_hh_l0": rng.uniform(-k, k, 3 * H),
          "output.weight": rng.uniform(-k, k, (1, H)), "output.bias": rng.uniform(-k, k, 1)}
    sd = {n: v.astype(np.float32) for n, v in sd.items()}
    m = ntm.RNN(1, H, 1)
    m.load_state_dict({n: torch.from_numpy(v) for n, v in sd.items()})
    m = m.to("cuda").eval()
    x = rng.uniform(-0.9, 0.9, (B, T)).astype(np.float32)
    h0 = rng.uniform(-0.5, 0.5, (B, H)).astype(np.float32)
    w = oracle.Weights.from_state_dict(sd)
    yo, ho = oracle.gru_forward(w, x, h0.copy())
    m.hidden = dev(h0).view(1, B, H).clone()
    y1 = m(dev(x).unsqueeze(1))
    h1 = m.hidden.clone()
    assert np.abs(y1[:, 0].cpu().numpy() - yo).max() < TOL
    assert np.abs(h1[0].cpu().numpy() - ho).max() < TOL
    m.hidden = dev(h0).view(1, B, H).clone()
    edges = [0] + sorted({c for c in cuts if c < T}) + [T]
    y2 = torch.cat([m(dev(x[:, a:b]).unsqueeze(1)) for a, b in zip(edges, edges[1:])], dim=2)
    assert torch.equal(y1, y2) and torch.equal(h1, m.hidden)


@settings(max_examples=40, **SET)
@given(B=st.integers(1, 70), T=st.integers(1, 400), seed=st.integers(0, 2**31 - 1), cuts=st.lists(st.integers(1, 399), max_size=2),
       scale=st.sampled_from([0.01, 0.125, 1.0, 3.0]), tiny=st.booleans())
def test_bf16x3_engine_random_weights_magnitudes_and_chunkings(ntm, B, T, seed, cuts, scale, tiny):
    """The operand-exact split engine (NTM_GRU_BF16X3, round 6) on RANDOM H = 64 weights whose magnitudes span five decades
    (`scale` x the PyTorch init range; `tiny`: a third of W_hh multiplied by 1e-4, so that the second and third bf16 pieces of
    some operands fall ten binades below the first) and random states: within 1e-5 of the oracle, carried state included, the
    exact engine within 2e-6 / 5e-6 beside it, chunked == one-shot bit for bit.  Any batch size: the variant bypasses the
    dispatch, a group of fewer than 16 streams included."""
    H = 64
    rng = np.random.default_rng(seed)
    k = scale / np.sqrt(H)
    whh = rng.uniform(-k, k, (3 * H, H))
    if tiny:
        whh[rng.random((3 * H, H)) < 1.0 / 3.0] *= 1e-4
    sd = {"GRU.weight_ih_l0": rng.uniform(-k, k, (3 * H, 1)), "GRU.weight_hh_l0": whh,
          "GRU.bias_ih_l0": rng.uniform(-k, k, 3 * H), "GRU.bias_hh_l0": rng.uniform(-k, k, 3 * H),
          "output.weight": rng.uniform(-k, k, (1, H)), "output.bias": rng.uniform(-k, k, 1)}
    sd = {n: v.astype(np.float32) for n, v in sd.items()}
    x = rng.uniform(-0.9, 0.9, (B, T)).astype(np.float32)
    h0 = (rng.uniform(-0.9, 0.9, (B, H)) * rng.choice([1.0, 1e-3, 1e-6], (B, 1))).astype(np.float32)
    w = oracle.Weights.from_state_dict(sd)
    yo, ho = oracle.gru_forward(w, x, h0.copy())
    outs = {}
    for variant in ("bf16x3", "mfma2"):
        m = ntm.RNN(1, H, 1)
        m.load_state_dict({n: torch.from_numpy(v) for n, v in sd.items()})
        m = m.to("cuda").eval()
        m.kernel_variant = variant
        m.hidden = dev(h0).view(1, B, H).clone()
        y1 = m(dev(x).unsqueeze(1))
        h1 = m.hidden.clone()
        assert np.abs(y1[:, 0].cpu().numpy() - yo).max() < TOL and np.abs(h1[0].cpu().numpy() - ho).max() < TOL, variant
        outs[variant] = (y1, h1)
        if variant == "bf16x3":
            m.hidden = dev(h0).view(1, B, H).clone()
            edges = [0] + sorted({c for c in cuts if c < T}) + [T]
            y2 = torch.cat([m(dev(x[:, a:b]).unsqueeze(1)) for a, b in zip(edges, edges[1:])], dim=2)
            assert torch.equal(y1, y2) and torch.equal(h1, m.hidden)
    assert (outs["bf16x3"][0] - outs["mfma2"][0]).abs().max().item() < 2e-6 * max(1.0, scale)
    assert (outs["bf16x3"][1] - outs["mfma2"][1]).abs().max().item() < 5e-6


# ----------------------------------------------------------------------------- K4: TCN, both tilings
DIL = st.one_of(st.integers(1, 40), st.integers(512, 700))


@settings(max_examples=40, **SET)
@given(B=st.integers(1, 3), T=st.integers(1, 3000), dil=st.lists(DIL, min_size=0, max_size=3), seed=st.integers(0, 1000))
def test_tcn_random_dilations_and_lengths(ntm, B, T, dil, seed):
    """Builder-defined TCN: random inner dilations (polyphase tiles below 512, phase-group tiles from 512 up, either as
    inner or as last block with the fused output conv), random ragged lengths: within 1e-5 of the oracle."""
    dil = tuple([1] + dil)
    m = ntm.TCN(dilations=dil, seed=seed).to("cuda")
    rng = np.random.default_rng(seed + T)
    x = rng.uniform(-0.8, 0.8, (B, T)).astype(np.float32)
    y = m(dev(x).unsqueeze(1)).cpu().numpy()[:, 0, :]
    yo = oracle.tcn_forward(m.packed_params().cpu().numpy(), len(dil), 32, 13, dil, x)
    assert np.abs(y - yo).max() < TOL


# ----------------------------------------------------------------------------- K3: ESR sums
@settings(max_examples=60, **SET)
@given(B=st.integers(1, 9), T=st.integers(2, 5000), skip_frac=st.floats(0.0, 0.9), seed=st.integers(0, 2**31 - 1))
def test_esr_sums_random_shapes(ntm, B, T, skip_frac, seed):
    """sum (t - y)^2 and sum t^2 per stream over [skip, T): fp64 accumulation on the device, 1e-9 relative to numpy fp64."""
    rng = np.random.default_rng(seed)
    y = rng.standard_normal((B, T)).astype(np.float32)
    t = rng.standard_normal((B, T)).astype(np.float32)
    skip = int(skip_frac * (T - 1))
    s = ntm.model.esr_sums(dev(y).unsqueeze(1), dev(t).unsqueeze(1), skip=skip).cpu().numpy()
    so = oracle.esr_sums(y, t, skip=skip)
    assert np.allclose(s, so, rtol=1e-9, atol=1e-12)


# ----------------------------------------------------------------------------- N3: demodulate, bit-exact
@settings(max_examples=40, **SET)
@given(N=st.integers(3000, 30000), period=st.integers(200, 500), delay0=st.integers(-40, 1500), wow=st.floats(0.0, 60.0),
       first=st.integers(0, 400), seed=st.integers(0, 2**31 - 1))
def test_demodulate_random_pulse_trains_bit_exact(ntm, N, period, delay0, wow, first, seed):
    """DelayAnalyzer.demodulate (code/utilities/utilities.py:408-465): random pulse period, static delay (also negative:
    no roll), wow depth and first pulse position; the searches start from a guess, the result is the oracle's to the bit."""
    rng = np.random.default_rng(seed)
    x_idx = np.arange(first, N - 3 * period, period)
    if len(x_idx) < 3:
        return
    dly = delay0 + wow * np.sin(2 * np.pi * 1.3 * x_idx / 44100) + 3 * np.sin(2 * np.pi * 11.0 * x_idx / 44100)
    y_idx = np.round(x_idx + dly).astype(np.int64)
    keep = (y_idx >= 0) & (y_idx < N)
    x_idx, y_idx = x_idx[keep].astype(np.int64), y_idx[keep]
    if len(y_idx) < 3 or np.any(np.diff(y_idx) <= 0):
        return
    out = rng.standard_normal((2, N)).astype(np.float32)
    ref = oracle.demodulate(out, x_idx, y_idx).astype(np.float32)
    got = ntm.feeder.demodulate(dev(out), x_idx, y_idx).cpu().numpy()
    assert np.array_equal(got, ref)


# ----------------------------------------------------------------------------- K1: dispatch boundaries of NTM_GRU_AUTO
@pytest.mark.parametrize("B", [1023, 1024, 1025, 4095, 4096, 4097, 4111, 4112, 4113, 5120, 6144, 8191, 8192, 8193, 12288])
def test_gru_auto_dispatch_boundaries(ntm, B):
    """kernel_variant="auto" around every batch size where the dispatch changes (low-latency kernel up to 1024 streams,
    one device round of the matrix-pipe kernel + remainder, several groups per CU, the small-LDS build from 8192): all
    streams against the oracle for tile-edge lengths, carried state included, chunked == one-shot bit for bit."""
    from helpers import oracle_weights
    w = oracle_weights(ntm.weights.W_GRU)
    m = ntm.harness.build_model(ntm.weights.W_GRU)
    m.kernel_variant = "auto"
    rng = np.random.default_rng(B)
    for T in (1, 63, 64, 65, 129):
        x = rng.uniform(-0.7, 0.7, (B, T)).astype(np.float32)
        h0 = rng.uniform(-0.3, 0.3, (B, 64)).astype(np.float32)
        yo, ho = oracle.gru_forward(w, x, h0.copy(), threads=8)
        m.hidden = dev(h0).view(1, B, 64).clone()
        y = m(dev(x).unsqueeze(1))
        h = m.hidden.clone()
        assert np.abs(y[:, 0].cpu().numpy() - yo).max() < TOL, (B, T)
        assert np.abs(h[0].cpu().numpy() - ho).max() < TOL, (B, T)
        if T > 1:
            m.hidden = dev(h0).view(1, B, 64).clone()
            c = T // 2
            y2 = torch.cat([m(dev(x[:, :c]).unsqueeze(1)), m(dev(x[:, c:]).unsqueeze(1))], dim=2)
            assert torch.equal(y, y2) and torch.equal(h, m.hidden), (B, T)


# ----------------------------------------------------------------------------- DiffDelGRU.predict, random delay trajectories
@settings(max_examples=20, **SET)
@given(B=st.integers(1, 12), T=st.integers(1, 2500), delay_s=st.floats(0.0005, 0.02), wow=st.floats(0.0, 0.3),
       seed=st.integers(0, 2**31 - 1), chunk=st.sampled_from([None, 256, 2048]))
def test_diffdel_predict_random_batches_and_trajectories(ntm, B, T, delay_s, wow, seed, chunk):
    """DiffDelRNN.predict (code/model.py:618-653; batched): random batch, length, model max_delay and wow depth; y and pre_d
    within 1e-5 of the oracle, y exactly the oracle's delay line applied to the device's own pre_d (bit for bit), and the
    reference's 2048-sample chunk loop (`segment_length`) equal to the single launch bit for bit."""
    from helpers import oracle_weights
    from test_gpu_parity import W_D
    fs = 44100
    m = ntm.harness.build_model(W_D, max_delay_seconds=delay_s)
    D = m.max_delay
    rng = np.random.default_rng(seed)
    x = rng.uniform(-0.6, 0.6, (B, T)).astype(np.float32)
    n = np.arange(T)
    base = delay_s * fs * rng.uniform(0.3, 0.9, (B, 1))
    d = base * (1.0 + wow * np.sin(2 * np.pi * rng.uniform(0.5, 3.0, (B, 1)) * n / fs + rng.uniform(0, 6.28, (B, 1))))
    d = np.clip(d, 0.0, D).astype(np.float32)
    kw = {} if chunk is None else {"segment_length": chunk}
    y, pre = m.predict(dev(x).unsqueeze(1), dev(d).unsqueeze(1), **kw)
    yo, preo, _, _ = oracle.diffdel_predict(oracle_weights(W_D), x, d, D, threads=4)
    assert np.abs(pre[:, 0].cpu().numpy() - preo).max() < TOL
    assert np.abs(y[:, 0].cpu().numpy() - yo).max() < TOL
    if chunk is not None:
        y1, pre1 = m.predict(dev(x).unsqueeze(1), dev(d).unsqueeze(1))
        assert torch.equal(y, y1) and torch.equal(pre, pre1)


# ----------------------------------------------------------------------------- the fused DiffDelRNN step (round 3)
@settings(max_examples=60, **SET)
@given(B=st.integers(1, 40), T=st.integers(1, 700), D=st.integers(1, 400), seed=st.integers(0, 2**31 - 1),
       cuts=st.lists(st.integers(1, 699), max_size=3), kind=st.sampled_from(["wow", "white", "integer", "tiny", "mixed"]))
def test_fused_diffdel_step_random_shapes_bit_identical_to_two_pass(ntm, B, T, D, seed, cuts, kind):
    """ntm_diffdel_gru_forward: the delay line fused into the GRU kernel (NTM_DIFFDEL_FUSED) against the GRU launch +
    streaming pass (NTM_DIFFDEL_TWO_PASS, same GRU kernel) on random batches, lengths (T < 64, T < D, T % 4 != 0), delay-line
    lengths, carried state and chunkings: pre_d, y, hidden state and delay buffer bit for bit; y also equals the oracle's
    delay line on the GPU's own pre_d."""
    rng = np.random.default_rng(seed)
    x = rng.uniform(-0.5, 0.5, (B, T)).astype(np.float32)
    n = np.arange(T)
    if kind == "wow":
        d = 0.5 * D + 0.45 * D * np.sin(2 * np.pi * n[None, :] / rng.uniform(20, 900, (B, 1)) + rng.uniform(0, 6, (B, 1)))
    elif kind == "white":
        d = rng.uniform(-0.9, D, (B, T))
    elif kind == "integer":
        d = np.round(rng.uniform(0, D, (B, T)))
    elif kind == "tiny":
        d = np.abs(2.5 * np.sin(n[None, :] / rng.uniform(3, 30, (B, 1))))
    else:
        d = np.where(rng.uniform(size=(B, T)) < 0.05, rng.uniform(-0.9, D, (B, T)), 0.3 * D + 0.1 * n[None, :] % max(D * 0.6, 1))
    d = np.clip(d, -0.9, D).astype(np.float32)
    d[rng.integers(0, B), rng.integers(0, T)] = D
    h0 = rng.uniform(-0.3, 0.3, (B, 64)).astype(np.float32)
    b0 = rng.uniform(-0.3, 0.3, (B, D)).astype(np.float32)
    edges = [0] + sorted({c for c in cuts if c < T}) + [T]
    res = {}
    for mode in ("two_pass", "fused"):
        m = ntm.DiffDelRNN(1, 64, 1, max_delay=D - 1)
        m.load_state_dict(ntm.weights.load_state_dict(ntm.weights.W_DIFFDEL))
        m = m.to("cuda").eval()
        m.delay_mode = mode
        if mode == "two_pass":
            m.kernel_variant = "mfma2"
        m.initialize_hidden(B, D - 1)
        m.hidden, m.diffdel.buffer = dev(h0).view(1, B, 64), dev(b0).view(B, 1, D)
        outs = [m(dev(x[:, a:b]).unsqueeze(1), dev(d[:, a:b]).unsqueeze(1)) for a, b in zip(edges, edges[1:])]
        res[mode] = (torch.cat([o[0] for o in outs], 2)[:, 0], torch.cat([o[1] for o in outs], 2)[:, 0], m.hidden.clone(), m.diffdel.buffer.clone())
    for a, b, what in zip(res["two_pass"], res["fused"], ("y", "pre_d", "hidden", "buffer")):
        assert torch.equal(a, b), what
    yo, bo = oracle.delay_forward(res["fused"][1].cpu().numpy(), d, b0)
    assert np.array_equal(res["fused"][0].cpu().numpy(), yo) and np.array_equal(res["fused"][3][:, 0].cpu().numpy(), bo)


@settings(max_examples=30, **SET)
@given(B=st.integers(1, 24), T=st.integers(1, 3000), D=st.integers(1800, 12000), seed=st.integers(0, 2**31 - 1),
       cuts=st.lists(st.integers(1, 2999), max_size=2), kind=st.sampled_from(["tape", "sweep", "white", "edge"]))
def test_fused_diffdel_step_real_tape_delay_lengths(ntm, B, T, D, seed, cuts, kind):
    """Round 4: the same property at the delay-line lengths the harness builds (code/test-model.py:222-230: D = 1847 for the toy
    data, about 11 000 for the real tape; round 3 drew D <= 400): every tap of a short call lies in the carried history (T < D),
    trajectories near the head distance with wow, full-range sweeps, white delays, and the edges d = D / d = 0 / d just below 0."""
    rng = np.random.default_rng(seed)
    x = rng.uniform(-0.5, 0.5, (B, T)).astype(np.float32)
    n = np.arange(T)
    if kind == "tape":
        d = D * rng.uniform(0.6, 0.8, (B, 1)) + D * 0.04 * np.sin(2 * np.pi * n[None, :] / rng.uniform(200, 5000, (B, 1)) + rng.uniform(0, 6, (B, 1)))
    elif kind == "sweep":
        d = 0.5 * D * (1 + np.sin(2 * np.pi * n[None, :] / rng.uniform(300, 4000, (B, 1)) + rng.uniform(0, 6, (B, 1))))
    elif kind == "white":
        d = rng.uniform(-0.9, D, (B, T))
    else:
        d = rng.choice(np.array([0.0, -0.5, D, D - 0.25, D - 1.0, 1.0, 0.5 * D]), size=(B, T))
    d = np.clip(d, -0.9, D).astype(np.float32)
    h0 = rng.uniform(-0.3, 0.3, (B, 64)).astype(np.float32)
    b0 = rng.uniform(-0.3, 0.3, (B, D)).astype(np.float32)
    edges = [0] + sorted({c for c in cuts if c < T}) + [T]
    res = {}
    for mode in ("two_pass", "fused"):
        m = ntm.DiffDelRNN(1, 64, 1, max_delay=D - 1)
        m.load_state_dict(ntm.weights.load_state_dict(ntm.weights.W_DIFFDEL))
        m = m.to("cuda").eval()
        m.delay_mode = mode
        if mode == "two_pass":
            m.kernel_variant = "mfma2"
        m.initialize_hidden(B, D - 1)
        m.hidden, m.diffdel.buffer = dev(h0).view(1, B, 64), dev(b0).view(B, 1, D)
        outs = [m(dev(x[:, a:b]).unsqueeze(1), dev(d[:, a:b]).unsqueeze(1)) for a, b in zip(edges, edges[1:])]
        res[mode] = (torch.cat([o[0] for o in outs], 2)[:, 0], torch.cat([o[1] for o in outs], 2)[:, 0], m.hidden.clone(), m.diffdel.buffer.clone())
    for a, b, what in zip(res["two_pass"], res["fused"], ("y", "pre_d", "hidden", "buffer")):
        assert torch.equal(a, b), what
    yo, bo = oracle.delay_forward(res["fused"][1].cpu().numpy(), d, b0)
    assert np.array_equal(res["fused"][0].cpu().numpy(), yo) and np.array_equal(res["fused"][3][:, 0].cpu().numpy(), bo)


# ----------------------------------------------------------------------------- the loss leg inside the launch (round 3)
@settings(max_examples=40, **SET)
@given(B=st.sampled_from([1, 17, 1030, 1100, 2070]), T=st.integers(1, 900), skip4=st.integers(0, 230), odd=st.booleans(),
       seed=st.integers(0, 2**31 - 1), diffdel=st.booleans())
def test_forward_esr_random_shapes(ntm, B, T, skip4, odd, seed, diffdel):
    """RNN.forward_esr / DiffDelRNN.forward_esr on random shapes and skips (multiples of 4: inside the launch where the
    matrix-pipe kernel runs; odd ones and small batches: forward + streaming pass): outputs and state bit-identical to
    forward(), sums equal to esr_sums() of that output to fp64 summation order."""
    rng = np.random.default_rng(seed)
    skip = min(T, 4 * skip4 + (1 if odd else 0))
    x = dev(rng.uniform(-0.5, 0.5, (B, 1, T)).astype(np.float32))
    t = dev(rng.uniform(-0.5, 0.5, (B, 1, T)).astype(np.float32))
    h0 = dev(rng.uniform(-0.3, 0.3, (1, B, 64)).astype(np.float32))
    if diffdel:
        D = 64
        d = dev(np.clip(30 + 25 * np.sin(np.arange(T)[None, None, :] / rng.uniform(5, 90, (B, 1, 1))), 0, D).astype(np.float32))
        b0 = dev(rng.uniform(-0.3, 0.3, (B, 1, D)).astype(np.float32))
        res = []
        for fused in (True, False):
            m = ntm.DiffDelRNN(1, 64, 1, max_delay=D - 1)
            m.load_state_dict(ntm.weights.load_state_dict(ntm.weights.W_DIFFDEL))
            m = m.to("cuda").eval()
            m.initialize_hidden(B, D - 1)
            m.hidden, m.diffdel.buffer = h0.clone(), b0.clone()
            if fused:
                y, pre, s = m.forward_esr(x, d, t, skip)
            else:
                y, pre = m.forward(x, d)
                s = ntm.model.esr_sums(y, t, skip)
            res.append((y, pre, m.hidden, m.diffdel.buffer, s))
        for a, b in zip(res[0][:4], res[1][:4]):
            assert torch.equal(a, b)
        sa, sb = res[0][4].cpu().numpy(), res[1][4].cpu().numpy()
    else:
        m = ntm.harness.build_model(ntm.weights.W_GRU)
        m.hidden = h0.clone()
        y1, s1 = m.forward_esr(x, t, skip)
        h1 = m.hidden.clone()
        m.hidden = h0.clone()
        y2 = m.forward(x)
        assert torch.equal(y1, y2) and torch.equal(h1, m.hidden)
        sa, sb = s1.cpu().numpy(), ntm.model.esr_sums(y2, t, skip).cpu().numpy()
    assert np.abs(sa - sb).max() <= 1e-12 * max(1.0, np.abs(sb).max())
