"""Round 6: the bf16x3 engine of the matrix-pipe GRU kernel (NTM_GRU_BF16X3, opt-in, csrc/gru_mfma2.hip ENGINE 2): W_hh and h
are each split into three bf16 pieces -- 24 significant bits, the fp32 operands EXACTLY -- and W.h is the sum of eight of the
nine partial products (W_3.h_3 <= 2^-32 |W||h| is dropped), each exact in fp32, accumulated in fp32 on v_mfma_f32_16x16x32_bf16;
state, gates and head are the exact engine's fp32 code.  What runs at 4096 streams is a hand-scheduled step (four asm statements
with fixed registers); the same arithmetic scheduled by the compiler is kept as libntm_bf16x3c.so (`make -C csrc exp`).
Reference: code/model.py:81-82 (torch.nn.GRU + Linear); tolerance 1e-5 abs fp32 against the oracle / the reference's goldens."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import oracle
from helpers import ROOT, load, oracle_weights

TOL = 1e-5
W_G = "GRU-HS[64]-L[DCPreESR]-DS[ReelToReel_Dataset_MiniPulse100_CHOWTAPE]_BEST"
W_D = "DiffDelGRU-HS[64]-L[DCPreESR]-DS[ReelToReel_Dataset_MiniPulse100_CHOWTAPE_WOWFLUTTER]_BEST"


@pytest.fixture(scope="module")
def ntm():
    import ntm_amd
    assert torch.cuda.is_available(), "these tests need the MI355X"
    ntm_amd._lib.lib()
    return ntm_amd


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.gpu
@pytest.mark.parametrize("B,T", [(1040, 1), (1040, 2), (1040, 63), (1040, 64), (1040, 65), (1040, 129), (1056, 777), (4112, 300), (8200, 192),
                                 (16, 500), (3, 130)])
def test_bf16x3_against_the_oracle_ragged_shapes_and_carried_state(ntm, B, T):
    """The engine called directly (kernel_variant "bf16x3" bypasses the batch-size dispatch, so 3 and 16 streams run it too):
    one sample to a dozen tiles, odd and even step counts, one workgroup per CU (YPN = 16) and the many-groups build
    (B > 4096: YPN = 4), a remainder group of fewer than 16 streams; state carried over two calls; every stream against
    the oracle and against the exact engine (two fp32 evaluations of the same recurrence: 2e-6 in y, 5e-6 in the state)."""
    rng = np.random.default_rng(B * 1000 + T)
    x = rng.uniform(-0.6, 0.6, (B, T)).astype(np.float32)
    w = oracle_weights(W_G)
    rows = sorted({0, 1, 15, 16, B // 2, B - 2, B - 1} & set(range(B)))
    m = ntm.harness.build_model(W_G)
    m.kernel_variant = "bf16x3"
    m.initialize_hidden()
    cut = T // 3
    y = torch.cat([m(dev(x[:, :cut]).unsqueeze(1)), m(dev(x[:, cut:]).unsqueeze(1))], dim=2) if cut else m(dev(x).unsqueeze(1))
    yo, ho = oracle.gru_forward(w, x[rows])
    assert np.abs(y[rows, 0].cpu().numpy() - yo).max() < TOL and np.abs(m.hidden[0, rows].cpu().numpy() - ho).max() < TOL
    e = ntm.harness.build_model(W_G)
    e.kernel_variant = "mfma2"
    e.initialize_hidden()
    ye = e(dev(x).unsqueeze(1))
    assert (y - ye).abs().max().item() < 2e-6 and (m.hidden - e.hidden).abs().max().item() < 5e-6      # (|h| reaches 1, |y| 0.3)


@pytest.mark.gpu
def test_bf16x3_hand_scheduled_form_is_bit_identical_to_the_compiler_scheduled_form(tmp_path):
    """The asm statements of step_b fix registers, order and wait states by hand; the same arithmetic compiled without them
    (NTM3_ASM = 0, libntm_bf16x3c.so) must give the same BITS -- outputs and carried state, ragged lengths, both LDS layouts, and
    bit-pattern checksums of every stream of 4096 x 16 384 and 8200 x 4096 batches of the headline's own input generator.
    A stale register, a missing wait state or a wrong operand in the strings shows here as a difference, not as noise."""
    libdir = os.path.join(ROOT, "neural-tape-modeling_amd")
    outs = {}
    for lib in ("libntm.so", "libntm_bf16x3c.so"):
        path = os.path.join(libdir, lib)
        assert os.path.exists(path), f"{path} is missing: __graft_entry__.build() runs `make -C csrc all exp`"
        dst = str(tmp_path / (lib + ".npz"))
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "bf16x3_dump.py"), dst], env=dict(os.environ, NTM_LIB_PATH=path),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and f"lib {lib} ok" in r.stdout, r.stderr[-2000:]
        outs[lib] = np.load(dst)
    a, b = outs["libntm.so"], outs["libntm_bf16x3c.so"]
    assert sorted(a.files) == sorted(b.files) and len(a.files) >= 17           # 11 arrays + 2 x 3 checksums of the large batches
    for k in a.files:
        assert a[k].shape == b[k].shape and np.isfinite(a[k]).all() and np.array_equal(a[k], b[k]), k


def test_bf16x3_split_is_exact_and_the_dropped_term_is_below_fp32_rounding():
    """Host-side statement of what the engine's operand split does (numpy restatement of split_bf16x3, round to nearest even):
    three bf16 pieces reproduce EVERY fp32 value bit for bit -- weights of every shipped GRU-HS[64] checkpoint after the
    log2(e) prescale, and 10^6 random states in (-1, 1) -- and the one partial product the engine drops, W_3.h_3, is bounded by
    2^-32 |W||h|: 1/256 of a single fp32 rounding of the product."""
    def bf16_rne(v):
        u = v.astype(np.float32).view(np.uint32).astype(np.uint64)
        r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
        return r.astype(np.uint32).view(np.float32)

    def split(v):
        p0 = bf16_rne(v)
        r1 = (v - p0).astype(np.float32)
        p1 = bf16_rne(r1)
        r2 = (r1 - p1).astype(np.float32)
        return p0, p1, bf16_rne(r2)

    rng = np.random.default_rng(6)
    vals = [np.tanh(rng.standard_normal(10**6)).astype(np.float32), rng.uniform(-1, 1, 10**6).astype(np.float32) * np.float32(1e-3)]
    g = load("g19_checkpoints.npz")
    import ntm_amd
    for name in [str(n) for n in g["names"] if str(n).startswith("GRU-HS[64]")]:
        w = ntm_amd.weights.load_state_dict(name)["GRU.weight_hh_l0"].numpy()
        vals += [(w[:128] * np.float32(-1.44269504088896340736)).astype(np.float32), (w[128:] * np.float32(2 * 1.44269504088896340736)).astype(np.float32)]
    for v in vals:
        v = v.reshape(-1)
        p0, p1, p2 = split(v)
        assert np.array_equal((p0.astype(np.float64) + p1.astype(np.float64) + p2.astype(np.float64)).astype(np.float32).view(np.uint32), v.view(np.uint32))
        assert np.array_equal(p0.astype(np.float64) + p1 + p2, v.astype(np.float64))            # exactly, not just after rounding
        nz = v != 0
        assert (np.abs(p2[nz]) <= np.abs(v[nz]) * 2.0 ** -16).all() and (np.abs(p1[nz]) <= np.abs(v[nz]) * 2.0 ** -8).all()


@pytest.mark.gpu
def test_bf16x3_diffdel_two_pass_and_variant_refusals(ntm):
    """DiffDelRNN with kernel_variant "bf16x3": the GRU launch on the split engine + the streaming delay pass (the fused step is
    the exact engine's); against the oracle.  Hidden sizes other than 64 refuse the variant with NTM_EINVAL."""
    B, T = 1040, 700
    rng = np.random.default_rng(66)
    x = rng.uniform(-0.5, 0.5, (B, T)).astype(np.float32)
    d = (150.0 + 140.0 * np.sin(np.arange(T) / 53.0 + rng.uniform(0, 6, (B, 1)))).astype(np.float32)
    m = ntm.DiffDelRNN(1, 64, 1, skip=False, max_delay=299)
    m.load_state_dict(ntm.weights.load_state_dict(W_D))
    m = m.to("cuda").eval()
    m.kernel_variant = "bf16x3"
    y, pre = m.predict(dev(x).unsqueeze(1), dev(d).unsqueeze(1))
    rows = [0, 17, 519, B - 1]
    yo, po, _, _ = oracle.diffdel_predict(oracle_weights(W_D), x[rows], d[rows], 299)
    assert np.abs(y[rows, 0].cpu().numpy() - yo).max() < TOL and np.abs(pre[rows, 0].cpu().numpy() - po).max() < TOL
    torch.manual_seed(16)
    m16 = ntm.RNN(1, 16, 1).to("cuda").eval()
    m16.kernel_variant = "bf16x3"
    with pytest.raises(ntm._lib.NtmError, match="hidden size 64"):
        m16(dev(x[:4, :8]).unsqueeze(1))


# ----------------------------------------------------------------------------- A1: any input_size / output_size
def _g22_cases():
    g = load("g22_io_sizes.npz")
    return [str(c) for c in g["cases"]]


@pytest.mark.gpu
@pytest.mark.parametrize("name", _g22_cases())
def test_g22_general_input_and_output_sizes_against_the_reference(ntm, name):
    """`RNN(input_size, hidden_size, output_size[, skip])` for sizes other than 1 (code/model.py:22,44-45; the only part of the
    object protocol round 5 still refused): forward() of the REFERENCE on seeded weights and inputs (golden g22,
    tools/make_goldens_io.py) -- nn.GRU(I, H) + nn.Linear(H, O), the (B, C, T) <-> (B, T, C) `reshape` of code/model.py:77,87 that is
    a reinterpretation and not a transpose, state carried over two calls, the skip connection -- within 1e-5, carried state too.
    warm_start() / predict() raise what the reference raises for input_size != 1 (it feeds zeros((1, 1, 1024)))."""
    g = load("g22_io_sizes.npz")
    I, H, O, skip, cut = (int(v) for v in g[f"{name}__meta"])
    m = ntm.RNN(I, H, O, skip=bool(skip))
    m.load_state_dict({k.split("__", 1)[1]: torch.from_numpy(g[k]) for k in g.files if "__GRU." in k and k.startswith(name + "__")
                       or "__output." in k and k.startswith(name + "__")})
    m = m.to("cuda").eval()
    x = dev(g[f"{name}__x"])
    m.initialize_hidden()
    y = torch.cat([m(x[:, :, :cut]), m(x[:, :, cut:])], dim=2)
    assert tuple(y.shape) == g[f"{name}__y"].shape and y.dtype == torch.float32
    assert np.abs(y.cpu().numpy() - g[f"{name}__y"]).max() < TOL and np.abs(m.hidden[0].cpu().numpy() - g[f"{name}__h"]).max() < TOL
    with pytest.raises(RuntimeError, match="must be equal to input_size"):          # torch.nn.GRU's message
        m(x[:, :1, :8].repeat(1, I + 1, 1))
    if I != 1:
        with pytest.raises(RuntimeError, match="Expected %d, got 1" % I):
            m.predict(x)


@pytest.mark.gpu
def test_general_sizes_random_shapes_against_the_oracle_and_the_size_one_kernels(ntm):
    """ntm_gru_forward_io on seeded random sizes (I, O up to 70, H up to 300) against the oracle; and at input_size = output_size = 1
    against the kernels every caller of the reference really runs (same weights through RNN's normal path): within 2e-6."""
    rng = np.random.default_rng(22)
    for it in range(14):
        I, O = int(rng.integers(1, 71)), int(rng.integers(1, 71))
        H = int(rng.choice([1, 3, 8, 24, 64, 65, 130, 300]))
        B, T = int(rng.integers(1, 9)), int(rng.integers(1, 90))
        torch.manual_seed(it)
        m = ntm.RNN(I, H, O).to("cuda").eval()
        sd = {k: v.cpu().numpy() for k, v in m.state_dict().items()}
        x = rng.uniform(-0.7, 0.7, (B, I, T)).astype(np.float32)
        h0 = rng.uniform(-0.5, 0.5, (B, H)).astype(np.float32)
        m.hidden = dev(h0).view(1, B, H).clone()
        y = m(dev(x))
        yo, ho = oracle.gru_forward_io(sd, x, h0)
        assert np.abs(y.cpu().numpy() - yo).max() < TOL and np.abs(m.hidden[0].cpu().numpy() - ho).max() < TOL, (I, H, O, B, T)
    # the C entry point at I = O = 1 against the product kernels
    L = ntm._lib
    for H in (16, 64, 96):
        torch.manual_seed(H)
        m = ntm.RNN(1, H, 1).to("cuda").eval()
        x = dev(rng.uniform(-0.6, 0.6, (5, 1, 333)).astype(np.float32))
        y_ref = m(x)
        y = torch.empty(5, 333, device="cuda")
        g, o = m.GRU, m.output
        rc = L.lib().ntm_gru_forward_io(L.ptr(g.weight_ih_l0), L.ptr(g.weight_hh_l0), L.ptr(g.bias_ih_l0), L.ptr(g.bias_hh_l0), L.ptr(o.weight),
                                        L.ptr(o.bias), H, 1, 1, L.ptr(x), L.ptr(y), 5, 333, 333, 333, None, L.current_stream())
        assert rc == 0, L.lib().ntm_last_error()
        assert (y - y_ref[:, 0]).abs().max().item() < 2e-6
    assert L.lib().ntm_gru_forward_io(None, None, None, None, None, None, 64, 0, 1, None, None, 1, 1, 1, 1, None, None) == -1
    assert b"input_size and output_size" in L.lib().ntm_last_error()
