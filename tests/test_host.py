"""CPU: host logic, the C-ABI surface, name parsers, weights, and the no-fallback guarantee."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

import ntm_amd
from helpers import GOLDEN, ROOT, state_dict_np


def header_symbols(name="ntm.h"):
    text = open(os.path.join(ROOT, "include", name)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ntm_[a-z0-9_]+)\s*\(", text)))


def test_cabi_library_loads_and_exports_every_declared_symbol():
    syms = header_symbols()
    assert {"ntm_gru_forward", "ntm_gru_forward_ex", "ntm_delay_forward", "ntm_diffdel_gru_forward",
            "ntm_esr_sums", "ntm_tcn_forward", "ntm_last_error", "ntm_abi_version"} <= set(syms)
    lib = ctypes.CDLL(ntm_amd._lib.LIB_PATH)
    for s in syms:
        assert hasattr(lib, s), f"libntm.so does not export {s}"
    assert set(ntm_amd._lib._SIGNATURES) == set(syms)       # the ctypes table covers the whole header
    assert not any(s.startswith("ntm_debug") or s.startswith("ntm_lab") for s in syms)
    # the laboratory library (older / experimental kernels, diagnostic builds) has its own header and is separate
    lab_syms = header_symbols("ntm_lab.h")
    lab = ctypes.CDLL(ntm_amd._lib.LAB_PATH)
    for s in lab_syms:
        assert hasattr(lab, s), f"libntm_lab.so does not export {s}"
        assert not hasattr(lib, s), f"the product library exports the laboratory symbol {s}"
    assert set(ntm_amd._lib._LAB_SIGNATURES) == set(lab_syms)
    assert ntm_amd._lib.lib().ntm_abi_version() == 9 == ntm_amd._lib.ABI_VERSION


def test_rccl_helper_library_loads_and_exports_its_header():
    """include/ntm_rccl.h -> libntm_rccl.so (the one collective of the sharded path for torch-free callers): loads without a
    GPU, exports every declared symbol, validates its arguments before RCCL is touched; libntm.so itself has no RCCL dependency."""
    import subprocess
    syms = header_symbols("ntm_rccl.h")
    assert syms == ["ntm_rccl_allreduce_f64", "ntm_rccl_comm_create", "ntm_rccl_comm_destroy", "ntm_rccl_last_error", "ntm_rccl_unique_id"]
    path = os.path.join(os.path.dirname(ntm_amd._lib.LIB_PATH), "libntm_rccl.so")
    R = ctypes.CDLL(path)
    for s in syms:
        assert hasattr(R, s)
    R.ntm_rccl_last_error.restype = ctypes.c_char_p
    comm = ctypes.c_void_p()
    idb = (ctypes.c_ubyte * 128)()
    assert R.ntm_rccl_comm_create(None, 1, 0, idb) == -1 and b"null pointer" in R.ntm_rccl_last_error()
    assert R.ntm_rccl_comm_create(ctypes.byref(comm), 2, 2, idb) == -1 and b"[0, nranks)" in R.ntm_rccl_last_error()
    assert R.ntm_rccl_comm_create(ctypes.byref(comm), 0, 0, idb) == -1
    assert R.ntm_rccl_allreduce_f64(None, ctypes.c_int64(0), None, None) == 0             # nothing to do
    assert R.ntm_rccl_allreduce_f64(None, ctypes.c_int64(4), None, None) == -1
    assert R.ntm_rccl_comm_destroy(None) == 0
    needed = subprocess.run(["readelf", "-d", ntm_amd._lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "rccl" not in needed.lower()
    # the four loss scalars: argument checks on the host
    L = ntm_amd._lib.lib()
    one = ctypes.c_void_p(16)
    assert L.ntm_loss_scalars(one, 4, 0, 1e-5, one, None) == -1 and b"bad size" in L.ntm_last_error()
    assert L.ntm_loss_scalars(None, 4, 10, 1e-5, one, None) == -1 and L.ntm_loss_scalars(one, 4, 10, 1e-5, None, None) == -1


def test_cabi_pure_host_entry_points():
    L = ntm_amd._lib.lib()
    assert L.ntm_tcn_scratch_floats(2, 100, 32) == 2 * (2 * 100 * 32 + 16 * 32)      # two activation buffers + one row block of padding each
    # the TCN scratch is bounded for any batch: chunks of streams on two lanes, <= 2e9 floats in all (round 3: 2 x 34.4 GB at
    # 4096 x 65536, nothing at all from 8192 streams up)
    assert L.ntm_tcn_chunk_streams(2, 100, 32) == 2 and L.ntm_tcn_chunk_streams(4096, 65536, 32) == 228
    for B in (4096, 8192, 16384, 32768):
        bc = L.ntm_tcn_chunk_streams(B, 65536, 32)
        assert 4 * bc * 65536 * 32 <= 2 * 10**9 and L.ntm_tcn_scratch_floats(B, 65536, 32) == 4 * (bc * 65536 * 32 + 512) <= 2.0e9 + 2048
    assert L.ntm_tcn_chunk_streams(200, 65536, 32) == 200 and L.ntm_tcn_scratch_floats(200, 65536, 32) == 2 * (200 * 65536 * 32 + 512)
    assert L.ntm_tcn_chunk_streams(3, 1 << 28, 32) == 1            # one stream longer than the budget: that stream alone
    # argument validation happens before anything touches a device
    assert L.ntm_gru_forward(None, None, None, None, None, None, 1025, None, None, 1, 1, 1, 1, None, None) == -1
    assert b"[1, 1024]" in L.ntm_last_error()
    assert L.ntm_gru_forward(None, None, None, None, None, None, 24, None, None, 1, 1, 1, 1, None, None) == -1
    assert b"null pointer" in L.ntm_last_error()                   # round 5: any hidden size in range is accepted
    assert L.ntm_gru_forward(None, None, None, None, None, None, 8, None, None, 1, 1, 1, 1, None, None) == -1
    assert b"null pointer" in L.ntm_last_error()
    # the matrix-pipe variants exist for H = 64 only (checked before any device call)
    one = ctypes.c_void_p(16)
    assert L.ntm_gru_forward_ex(one, one, one, one, one, None, 16, one, one, 1, 1, 1, 1, None,
                                ntm_amd._lib.NTM_GRU_MFMA2, None) == -1
    assert b"hidden size 64 only" in L.ntm_last_error()
    assert L.ntm_esr_sums(None, None, 1, 10, 11, 1, None, None) == -1
    two = ctypes.c_void_p(32)
    # ntm_gru_forward_esr rejects a strided y for streams outside the matrix-pipe launch BEFORE anything is enqueued
    # (round 3 launched the forward first and then returned NTM_EINVAL with y half-written)
    assert L.ntm_gru_forward_esr(one, one, one, one, one, None, 64, one, one, 4, 100, 100, 128, None, two, 0, one, None) == -1
    assert b"contiguous y rows" in L.ntm_last_error()
    assert L.ntm_gru_forward_esr(one, one, one, one, one, None, 2000, one, one, 4, 100, 100, 100, None, two, 0, one, None) == -1
    assert b"[1, 1024]" in L.ntm_last_error()
    assert L.ntm_gru_forward_esr(one, one, one, one, one, None, 64, one, None, 4, 100, 100, 100, None, two, 0, one, None) == -1
    assert b"null pointer" in L.ntm_last_error()
    # the DiffDelGRU entries check signs first (a negative B used to reach a launch with (unsigned)B blocks), B == 0 is a no-op
    for mode in (ntm_amd._lib.DIFFDEL_MODES["auto"], ntm_amd._lib.DIFFDEL_MODES["two_pass"]):
        assert L.ntm_diffdel_gru_forward_ex(one, one, one, one, one, 64, one, one, one, ctypes.c_void_p(32), -1, 10, None, one, 5, 0, None, mode, None) == -1
        assert b"negative size" in L.ntm_last_error()
        assert L.ntm_diffdel_gru_forward_ex(one, one, one, one, one, 64, one, one, one, ctypes.c_void_p(32), 0, 10, None, one, 5, 0, None, mode, None) == 0
    assert L.ntm_diffdel_gru_forward_ex(one, one, one, one, one, 64, one, one, one, ctypes.c_void_p(32), 3, 10, None, one, -5, 0, None, 0, None) == -1
    assert L.ntm_esr_splits(4096, 65536, 1024) == 1 and L.ntm_esr_splits(1, 65536, 0) == 16 and L.ntm_esr_splits(16, 8192, 0) == 2


def test_name_parsers_match_reference_table():
    tab = json.load(open(os.path.join(GOLDEN, "g7_name_parsers.json")))
    assert len(tab["names"]) == 44
    for row in tab["names"]:
        assert ntm_amd.parse_model(row["name"]) == row["model"]
        assert ntm_amd.parse_hidden_size(row["name"]) == row["hidden"]
        assert ntm_amd.parse_loss(row["name"]) == row["loss"]
    for n, v in tab["nextpow2"].items():
        assert ntm_amd.nextpow2(int(n)) == v
    with pytest.raises(AttributeError):                        # reference behaviour on its own default
        ntm_amd.parse_loss("GRU-HS[64]-DS[foo]")              # --WEIGHTS (no -L[..]): re.search -> None


def test_state_dict_protocol():
    m = ntm_amd.RNN(input_size=1, hidden_size=64, output_size=1, skip=False)
    assert list(m.state_dict()) == ["GRU.weight_ih_l0", "GRU.weight_hh_l0", "GRU.bias_ih_l0",
                                    "GRU.bias_hh_l0", "output.weight", "output.bias"]
    assert sum(p.numel() for p in m.parameters()) == 12929
    d = ntm_amd.DiffDelRNN(input_size=1, hidden_size=64, output_size=1, skip=False, max_delay=1846)
    assert "output.bias" not in d.state_dict() and sum(p.numel() for p in d.parameters()) == 12928
    assert d.diffdel.max_delay == 1847 and tuple(d.diffdel.buffer.shape) == (2, 1, 1847)   # code/model.py:370-375
    for name in ntm_amd.weights.available():
        sd = ntm_amd.weights.load_state_dict(name)
        ref = state_dict_np(name)
        assert set(sd) == set(ref)
        model = ntm_amd.harness.build_model(name, device="cpu")
        for k, v in model.state_dict().items():
            assert np.array_equal(v.numpy(), ref[k])
    with pytest.raises(RuntimeError):                          # strict load like torch: bias key missing
        m.load_state_dict(ntm_amd.weights.load_state_dict(ntm_amd.weights.W_DIFFDEL))
    # the reference's own defaults construct (code/model.py:22: hidden_size=8; code/train.py:50: 16)
    for H in (8, 16, 32, 5, 24, 48, 96, 200):                   # round 5: ANY hidden size (`--HIDDEN_SIZE`, code/train.py:50)
        r = ntm_amd.RNN() if H == 8 else ntm_amd.RNN(1, H, 1)
        assert r.hidden_size == H and tuple(r.GRU.weight_hh_l0.shape) == (3 * H, H)
        assert sum(p.numel() for p in r.parameters()) == 3 * H * H + 3 * H + 6 * H + H + 1
        torch_ref = torch.nn.GRU(1, H, batch_first=True)
        assert {k: tuple(v.shape) for k, v in r.GRU.state_dict().items()} == \
               {k: tuple(v.shape) for k, v in torch_ref.state_dict().items()}
    for bad in (0, -3, 1025, 2.5):
        with pytest.raises(ValueError):
            ntm_amd.RNN(1, bad, 1)                             # outside [1, NTM_MAX_HIDDEN] / not an integer
    # round 6: any input_size / output_size (code/model.py:22,44-45), parameter shapes as torch.nn.GRU / Linear make them
    r = ntm_amd.RNN(3, 24, 5)
    ref_g, ref_o = torch.nn.GRU(3, 24, batch_first=True), torch.nn.Linear(24, 5)
    assert {k: tuple(v.shape) for k, v in r.GRU.state_dict().items()} == {k: tuple(v.shape) for k, v in ref_g.state_dict().items()}
    assert {k: tuple(v.shape) for k, v in r.output.state_dict().items()} == {k: tuple(v.shape) for k, v in ref_o.state_dict().items()}
    with pytest.raises(RuntimeError, match="Expected 3, got 1"):      # the reference's warm_start feeds zeros((1, 1, 1024))
        r.warm_start()
    for bad in (0, 1025, 1.5):
        with pytest.raises(ValueError):
            ntm_amd.RNN(bad, 8, 1)
    with pytest.raises(ValueError, match="DiffDelRNN"):
        ntm_amd.DiffDelRNN(2, 8, 1)                            # its delay line is single-channel


def test_no_cpu_fallback():
    m = ntm_amd.harness.build_model(ntm_amd.weights.W_GRU, device="cpu")
    with pytest.raises(RuntimeError, match="HIP device only"):
        m(torch.zeros(1, 1, 16))
    with pytest.raises(RuntimeError, match="HIP device only"):
        m.predict(torch.zeros(1, 1, 16))
    d = ntm_amd.TimeVaryingDelayLine(max_delay=8)
    with pytest.raises(RuntimeError, match="HIP device only"):
        d(torch.zeros(2, 1, 16), torch.zeros(2, 1, 16))
    with pytest.raises(RuntimeError, match="HIP device only"):
        ntm_amd.esr_sums(torch.zeros(1, 1, 4), torch.zeros(1, 1, 4))
    # the product never imports, loads or links the oracle
    src_dir = os.path.join(ROOT, "neural-tape-modeling_amd")
    for dirpath, _, files in os.walk(src_dir):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", "Makefile")):
                text = open(os.path.join(dirpath, fn)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", text, flags=re.M), fn
                assert "ntm_oracle" not in text and "ntmo_" not in text, fn
    # nor do the developer tools (anything that checks against the oracle lives under tests/); the only files outside
    # tests/ that may touch oracle/ are __graft_entry__.py (smoke) and bench.py (its CPU legs)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "tools")):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hip", "Makefile")):
                text = open(os.path.join(dirpath, fn)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b|import ntm_amd, oracle", text, flags=re.M), fn
                assert "ntm_oracle" not in text and "ntmo_" not in text, fn


def test_shape_errors():
    m = ntm_amd.harness.build_model(ntm_amd.weights.W_GRU, device="cpu")
    with pytest.raises(RuntimeError):
        m(torch.zeros(4, 16))
    with pytest.raises(RuntimeError, match="input_size 1"):
        m(torch.zeros(1, 2, 16))


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 4096, 32768):
        for world in (1, 2, 3, 8):
            spans = [ntm_amd.distributed.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_g14_tape_stages_host_side():
    """The stages of the reference's Tape around H_mag that run on the host or are plain gains (code/tape.py:466-510,
    565-579): H_pre, the stateful bias waveform (start-up ramp, phase carried over two calls), H_play, H_post --
    bit-identical to the reference's own output (golden g14)."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "g14_tape_stages.npz"))
    tp = ntm_amd.TapeMagnetization(batch_size=2, device="cpu")
    sp, N = int(g["split"]), g["V"].shape[1]
    assert np.array_equal(tp.H_pre(g["V"]), g["I_in"])
    b1, b2 = tp.bias_signal(sp), tp.bias_signal(N - sp)
    assert np.array_equal(g["I_in"][:, :sp] + b1, g["I_rec"][:, :sp]) and np.array_equal(g["I_in"][:, sp:] + b2, g["I_rec"][:, sp:])
    assert tp.bias_phase == float(g["bias_phase_end"]) and not tp.FLAG_STARTUP
    assert np.array_equal(tp.H_play(g["M"]), g["V_play"]) and np.array_equal(tp.H_post(tp.H_play(g["M"])), g["V_out"])


def test_step_barrier_wait_count_matches_the_disassembly():
    """The step barrier of gru_mfma2_kernel is `s_waitcnt lgkmcnt(1); s_barrier` in inline asm: valid only while exactly
    one LDS/SMEM op (the y-partial ds_write_b32) sits between the h-exchange write and the barrier.  The compiler is not
    bound by that, so the disassembly of every product instantiation is checked (tools/check_barrier_asm.py)."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_barrier_asm.py")], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


def test_no_dpp_instruction_can_run_under_a_partial_exec_mask():
    """tools/check_dpp_exec.py: a forward data-flow over the control-flow graph of every kernel that uses cross-lane DPP operations
    (the DCPreESR scan of the recurrent kernel's flush, the quad / wave sums of the low-latency and small-H kernels) proves that none
    of them can execute while a divergent region is open.  Round 5 lost a result to exactly that: hipcc sank a DPP move written
    under a select into the branch, where lanes read EXEC-disabled neighbours (the checker flags that build: verified by hand)."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_dpp_exec.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "none under a partial EXEC mask: ok" in r.stdout, r.stderr[-3000:]


def test_product_kernels_use_no_scratch_and_keep_their_register_budget():
    """tools/kernel_resources.py --check on the built libntm.so: the code objects' own metadata must show no scratch and no
    VGPR spill for ANY kernel, and the product instantiations of the recurrent kernel must not exceed the VGPR counts of the
    binary the committed profiles were measured on (a silent register-allocation change is a performance event); the record
    the Makefile wrote beside the library describes exactly this library (sha256)."""
    import hashlib
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_resources.py"), "--check"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "VGPR pins hold: ok" in r.stdout, r.stderr[-2000:]
    info = json.load(open(os.path.join(ROOT, "neural-tape-modeling_amd", "build_info.json")))
    with open(ntm_amd._lib.LIB_PATH, "rb") as f:
        assert hashlib.sha256(f.read()).hexdigest() == info["library_sha256"]
    names = [k["kernel"] for k in info["kernels"]]
    assert any("gru_mfma2_kernel<true, false, 0, 0, 16, false, true, true>" in n for n in names) and any("gru_wide_kernel<true>" in n for n in names)
    assert info["compiler"]["hip"] and all(k["scratch_bytes"] == 0 and k["vgpr_spills"] == 0 for k in info["kernels"])


def test_recorded_bench_line_follows_the_contract():
    """The last default `python bench.py` run recorded on the MI355X (profiles/): the compact line -- what the driver parses, < 4 KB,
    round 5's 20 KB line was lost -- carries every field of the driver's contract, the roofline of the dominant kernel and the CPU
    baseline; the detail file beside it carries the rest; their numbers are self-consistent and agree with each other."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*bench_default*.json")))
    assert files, "no recorded default bench line under profiles/"
    raw = open(files[-1]).read().strip().splitlines()[-1]
    c = json.loads(raw)
    assert "detail_file" in c and len(raw) < 4096, (files[-1], len(raw))
    d = json.load(open(files[-1].replace("bench_default", "bench_detail")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d and k in c, k
        if k not in ("config", "roofline", "cpu_baseline"):
            assert c[k] == d[k], k
    rc = c["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms"):
        assert rc[k] == d["roofline"][k], k
    assert abs(rc["frac"] - rc["achieved"] / rc["peak"]) < 1e-9 and rc["traffic_live"] is True and 1.0 <= rc["traffic_ratio"] < 1.01
    assert c["cpu_baseline"]["value"] == d["cpu_baseline"]["value"] and c["cpu_baseline"]["kind"] == "port" and c["cpu_baseline"]["cores"] >= 1
    assert c["config"]["segments_total"] == 4096 and c["config"]["samples_per_segment"] == 65536 and len(c["config"]["workload"]) <= 120
    assert c["checks"]["streams_vs_oracle_max_abs"] < 1e-5 and c["checks"]["esr_sums_max_rel"] < 1e-9 and c["checks"]["deterministic"] is True
    assert {"diffdel", "tcn", "gru_B8192", "gru_B16384", "gru_B32768", "cli"} <= set(c["legs"])
    for k, v in c["legs"].items():
        assert v["value"] > 0 and (k == "cli" or (0.3 < v["frac"] < 1.0 and v["vs_oracle_max_abs"] < 1e-5)), k
    # round 6: the two stale traffic figures re-measured with this round's binary -- every leg states PMC bytes over algorithmic bytes
    assert 1.0 <= c["legs"]["diffdel"]["traffic_ratio"] < 1.4 and c["legs"]["tcn"]["traffic_ratio"] > 50
    # round 6: the opt-in split engines ride beside `value` (never in it): kernel time and whole-batch distance from the exact pass;
    # the detail file prices the bf16x3 engine both ways (executed bf16 MFMA flops / bf16 peak, algorithmic flops / fp32 peak)
    e = c["opt_in_engines"]
    assert 20 < e["f16x3"]["kernel_ms"] < e["bf16x3"]["kernel_ms"] < rc["kernel_ms"] and e["bf16x3"]["vs_exact_fp32_max_abs"] < 1e-5
    b3 = d["other_kernels"]["bf16x3"]
    assert abs(b3["roofline_executed"]["frac"] - 196608.0 * 4096 * 65536 / (b3["kernel_ms"] * 1e-3) / 2.5e15) < 1e-9
    assert abs(b3["roofline_algorithmic"]["frac"] - 25088.0 * 4096 * 65536 / (b3["kernel_ms"] * 1e-3) / 157.3e12) < 1e-9
    assert b3["speedup_vs_exact_fp32_kernel"] > 1.3 and b3["stream0_vs_reference_max_abs"] < 1e-5
    assert b3["at_8192_streams"]["speedup_vs_exact_fp32_kernel"] > 1.5        # two groups per CU: one's exchange under the other's MFMAs
    assert c["build"]["hip"] and c["build"]["runtime_hip"] and len(c["build"]["library_sha256"]) == 16
    assert d["build"]["library_sha256"].startswith(c["build"]["library_sha256"])
    assert d["unit"] == "samples/s" and d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["dtype"] == "f32" and d["scaling"] in ("weak", "strong") and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["unit"] in ("GB/s", "TFLOP/s")
    assert r["traffic"] is None or r["traffic"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    # whole-job throughput = segments x samples x steps / time;  kernel time <= step time
    seg, T = d["config"]["segments_total"], d["config"]["samples_per_segment"]
    assert abs(d["value"] - seg * T / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert r["kernel_ms"] <= d["ms_per_step"]
    assert abs(r["achieved"] - 25088.0 * seg * T / d["n_gpus"] / (r["kernel_ms"] * 1e-3) / 1e12) < 1e-6 * r["achieved"]
    # round 3: the BASELINE configs no other driver-run line covers ride along (never part of `value`): configs[2], [3] and
    # the GRU workload at the per-GPU shapes of configs[4]'s strong-scaling legs, each checked against the oracle
    ow = d["other_workloads"]
    assert {"diffdel", "tcn", "gru_B8192", "gru_B16384", "gru_B32768"} <= set(ow) and "not a scaling curve" in ow["note"]
    for k, v in ow.items():
        if k in ("note", "cli"):
            continue
        assert v["kernel"] and 0 < v["kernel_ms"] <= v["ms_per_step"] * 1.001 and 0.3 < v["roofline"]["frac"] < 1.0
        assert v["checks"]["deterministic"] is True and len(v["checks"]["streams_checked"]) == 4
        assert v["checks"]["vs_oracle_max_abs"] < 1e-5 and v["checks"]["samples_each"] == 65536
    for b in (8192, 16384, 32768):
        v = ow[f"gru_B{b}"]
        flops = 25088.0 * b * 65536
        assert abs(v["roofline"]["achieved"] - flops / (v["kernel_ms"] * 1e-3) / 1e12) < 1e-6 * v["roofline"]["achieved"]
        # round 4: no `traffic: null` legs -- PMC bytes per launch within 1 % of the 8 algorithmic bytes per sample
        assert 1.0 <= v["roofline"]["traffic"] / (8.0 * b * 65536) < 1.01 and "pmc_traffic_gru_B" in v["roofline"]["traffic_source"]
    # round 4: the TCN states what it MOVES (not the 8 B/sample of a fused network), runs at the per-GPU shapes of configs[4] too
    t = ow["tcn"]
    assert 700 < t["bytes_per_sample"] < 850 and t["roofline"]["bytes_per_sample_if_fused"] == 8
    assert t["roofline"]["scratch_bytes"] <= 8.0e9 + 8192 and {"tcn_B8192", "tcn_B16384"} <= set(ow)
    assert abs(ow["tcn_B8192"]["kernel_ms"] / t["kernel_ms"] - 2.0) < 0.04            # same time per 4096 streams
    # round 4: the loss leg of the timed step runs against a real target and is checked against the oracle
    c = d["checks"]
    assert c["job_esr"] > 1e-4 and c["every_timed_step_same_loss"] is True and c["last_output_equals_first_pass_bitwise"] is True
    assert c["esr_sums_vs_oracle"]["max_rel"] < 1e-9 and c["esr_sums_vs_oracle"]["esr_max_rel_diff"] < 1e-3
    assert c["streams_vs_oracle"]["max_abs"] < 1e-5 and c["stream0_vs_reference_max_abs"] < 1e-5
    assert d["ms_per_step"] <= d["ms_per_step_no_warm_cache"] < 1.03 * d["ms_per_step"]
    assert d["roofline"]["traffic"] is not None and 1.0 <= d["roofline"]["traffic"] / (12.0 * seg * T) < 1.01
    # round 5: the line names its point on BASELINE's "1/2/4/8 GPU" axis, keeps the facts inside the driver's 120-character cut,
    # carries the record of the binary, the no-warm-cache value, the aggregate roofline, and the evaluation command end to end
    assert d["metric"] == "audio samples/sec (44.1 kHz) GRU-HS[64], batch=4096x65536, 1 GPU" and len(d["config"]["workload"]) <= 120
    assert "predict+ESR fused" in d["config"]["workload"] and "warm-cache on" in d["config"]["workload"]
    assert abs(d["value_no_warm_cache"] - seg * T / (d["ms_per_step_no_warm_cache"] * 1e-3)) < 1e-6 * d["value"]
    b = d["build"]
    assert b["compiler"]["hip"] and b["runtime_hip"] and len(b["library_sha256"]) == 64
    assert any("gru_mfma2_kernel<true, false, 0, 0, 16, false, true, false>" in k["kernel"] and k["scratch_bytes"] == 0 for k in b["kernels"])
    g = r["aggregate"]
    assert g["n_gpus"] == 1 and abs(g["frac"] - r["frac"]) < 1e-9 and len(g["kernel_ms_by_rank"]) == 1
    cli = ow["cli"]
    assert cli["unit"] == "samples/s" and abs(cli["value"] - 128 * 441000 / cli["command_s"]) < 1e-6 * cli["value"]
    assert cli["bound_by"] in cli["stages_ms"] and all(v > 0 for v in cli["stages_ms"].values()) and len(cli["stages_ms"]) == 7
    assert 0 < cli["gpu_busy_fraction_of_command"] <= cli["gpu_busy_fraction_of_loss_loop"] <= 1
    assert cli["cpu_baseline"]["kind"] == "port" and cli["cpu_baseline"]["value"] > 0 and "scripts/test-model-loss.sh" in cli["reference_command"]
    for k in ("ESR", "DCPreESR"):
        assert abs(cli["losses"][k] / cli["cpu_baseline"]["losses"][k] - 1) < 2e-3          # (128 segments vs the first 2: close, not equal)
    e = d["other_kernels"]["esr_dcpre_fused"]
    assert e["esr_sums_identical"] is True and e["dcpre_sums_max_rel_diff"] < 2e-6 and e["saved_ms"] > 0.3
