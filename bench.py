#!/usr/bin/env python3
"""Headline benchmark: audio samples/s through GRU-HS[64] (CHOWTAPE weights), batch 4096 segments
x 65 536 samples fp32 per GPU (BASELINE.json configs[1]); weak scaling over N GPUs (configs[4]:
N x 4096 segments, sharded by stream, one RCCL all-reduce of the ESR scalars).

A "step" = one pass of the hot path over the rank's batch, inputs resident in HBM:
    warm-start (1024 zero samples, B=1) -> persistent GRU kernel over [B,T] -> per-stream ESR sums
    against a resident target -> all-reduce of 4 fp64 scalars.
The target is the output of the first (untimed) pass, so the ESR of every timed pass must be
exactly 0.0 -- a full-size determinism check -- and stream 0 carries the input of golden G6 so the
result is also checked against the REFERENCE's own output on 65 536 samples.

    python bench.py [--gpus N --steps K --warmup W]          (N>1: launched by torch.distributed.run)
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FS = 44100
FLOP_PER_SAMPLE = 25088          # 12 544 FMA: W_hh GEMV 12 288 + W_ih 192 + head 64
BYTES_PER_SAMPLE = 8             # 4 B x in + 4 B y out
PEAK_FP32_TFLOPS = 157.3         # MI355X fp32 matrix == vector peak (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0


def synth_input(B, T, device, seed):
    """per-stream sinusoid (log-uniform 40 Hz-8 kHz) x 1-Hz raised-cosine
    envelope in [0.1,1] + 0.05 N(0,1); generated on the device."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    f = 40.0 * (8000.0 / 40.0) ** torch.rand(B, 1, generator=g, device=device)
    ph = 2 * np.pi * torch.rand(B, 1, generator=g, device=device)
    pe = 2 * np.pi * torch.rand(B, 1, generator=g, device=device)
    x = torch.empty(B, T, device=device)
    n = torch.arange(T, device=device, dtype=torch.float32).unsqueeze(0)
    CH = 256
    for b0 in range(0, B, CH):
        sl = slice(b0, min(B, b0 + CH))
        env = 0.55 - 0.45 * torch.cos(2 * np.pi * 1.0 * n / FS + pe[sl])
        x[sl] = 0.5 * torch.sin(2 * np.pi * f[sl] * n / FS + ph[sl]) * env
        x[sl] += 0.05 * torch.randn(x[sl].shape, generator=g, device=device)
    return x.unsqueeze(1)


def cpu_baseline(w_name, seconds_budget=12.0):
    """Reference CPU path timed on this host: stock torch.nn.GRU + Linear on CPU (the modules the
    reference builds at code/model.py:44-45) on a bounded sample of the workload (sized from a short
    calibration run to take ~seconds_budget), on the cores this process may use; the C oracle
    (OpenMP over streams) beside it."""
    import oracle
    from ntm_amd import weights
    sd = weights.load_state_dict(w_name)
    w = oracle.Weights.from_state_dict({k: v.numpy() for k, v in sd.items()})
    rng = np.random.default_rng(0)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 32))              # tiny per-step GEMMs do not scale past a few dozen threads
    torch.set_num_threads(cores)
    f = oracle.torch_gru_port(w)
    B = 16                                      # the reference's best CPU shape
    h0 = np.repeat(oracle.warm_state(w), B, 0)
    xc = rng.uniform(-0.5, 0.5, (B, 512)).astype(np.float32)
    f(xc, h0)                                   # thread-pool warm-up
    t0 = time.perf_counter()
    f(xc, h0)
    rate = xc.size / (time.perf_counter() - t0)
    T = int(min(65536, max(2048, 2 ** int(np.log2(max(rate * seconds_budget / B, 2048))))))
    x = rng.uniform(-0.5, 0.5, (B, T)).astype(np.float32)
    t0 = time.perf_counter()
    f(x, h0)
    dt = time.perf_counter() - t0
    res = {"value": x.size / dt, "unit": "samples/s", "cores": cores, "kind": "port",
           "sample": f"{B}x{T} samples of the same workload, torch.nn.GRU+Linear on CPU, {cores} threads"}
    xo = rng.uniform(-0.5, 0.5, (2 * cores, 4096)).astype(np.float32)
    t0 = time.perf_counter()
    oracle.gru_forward(w, xo, threads=cores)
    dt = time.perf_counter() - t0
    res["oracle_c"] = {"value": xo.size / dt, "unit": "samples/s", "cores": cores,
                       "sample": f"{xo.shape[0]}x4096 samples, C restatement, OpenMP over streams"}
    return res


def side_workload(a):
    """BASELINE configs[2] (DiffDelGRU-HS[64], CHOWTAPE_WOWFLUTTER weights) and configs[3] (TCN contrast
    point) at the same batch; single GPU, not the headline metric."""
    import ntm_amd
    from ntm_amd import weights
    assert torch.cuda.is_available()
    dev = torch.device("cuda", 0)
    B, T = a.batch, a.samples
    x = synth_input(B, T, dev, seed=1234)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if a.workload == "diffdel":
        model = ntm_amd.harness.build_model(weights.W_DIFFDEL, max_delay_seconds=0.0335, device=dev)   # D = 1847
        g = torch.Generator(device=dev); g.manual_seed(77)
        amp = 0.002 + 0.003 * torch.rand(B, 1, generator=g, device=dev)
        wv = 0.5 + 1.5 * torch.rand(B, 1, generator=g, device=dev)
        psi = 2 * np.pi * torch.rand(B, 1, generator=g, device=dev)
        n = torch.arange(T, device=dev, dtype=torch.float32).unsqueeze(0)
        d = torch.empty(B, T, device=dev)
        for b0 in range(0, B, 256):
            sl = slice(b0, min(B, b0 + 256))
            d[sl] = FS * (0.0271 + amp[sl] * torch.sin(2 * np.pi * wv[sl] * n / FS + psi[sl])
                          + 0.0005 * torch.sin(2 * np.pi * 23 * n / FS))
        d = d.clamp_(0, model.max_delay).unsqueeze(1)
        run = lambda: model.predict(x, d)[0]                                 # noqa: E731
        name, bytes_per_sample = "DiffDelGRU-HS[64] CHOWTAPE_WOWFLUTTER weights, D=1847", 16
    else:
        model = ntm_amd.TCN().to(dev)
        run = lambda: model(x)                                               # noqa: E731
        name, bytes_per_sample = "TCN 4x(k13, dil 1/10/100/1000, 32 ch) seeded weights", 8
    # CPU leg of the side workloads (skipped with --no-cpu-baseline, like the headline's cpu_baseline): stream 0
    # against the oracle over the whole sequence -- the oracle is the checker here, never the thing measured
    ref0 = None
    if not a.no_cpu_baseline:
        import oracle
    if a.no_cpu_baseline:
        pass
    elif a.workload == "diffdel":
        w_or = oracle.Weights.from_state_dict({k: v.numpy() for k, v in weights.load_state_dict(weights.W_DIFFDEL).items()})
        ref0 = oracle.diffdel_predict(w_or, x[:1, 0].cpu().numpy(), d[:1, 0].cpu().numpy(), model.max_delay)[0][0]
    else:
        ref0 = oracle.tcn_forward(model.packed_params().cpu().numpy(), len(model.dilations), model.channels,
                                  model.kernel_size, model.dilations, x[:1, 0].cpu().numpy())[0]
    for _ in range(max(a.warmup, 1)):
        y0 = run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ms = []
    for _ in range(a.steps):
        ev0.record(); y = run(); ev1.record(); torch.cuda.synchronize()
        ms.append(ev0.elapsed_time(ev1))
    elapsed = time.perf_counter() - t0
    print(json.dumps({
        "metric": f"audio samples/sec (44.1 kHz) {a.workload}, batch={B}x{T}", "value": B * T * a.steps / elapsed,
        "unit": "samples/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{name}, {B} segments x {T} samples fp32"},
        "device_ms_per_step": float(np.mean(ms)), "bytes_per_sample": bytes_per_sample,
        # whole-step roofline (several kernels per step): algorithmic flops over the device time of a step against the
        # fp32 peak; DiffDelGRU = the GRU flops (the delay line adds 3 flop/sample), TCN = 43 520 FMA per sample
        "roofline": (lambda fl: {"bound": "mfma", "achieved": fl * B * T / (np.mean(ms) * 1e-3) / 1e12, "peak": PEAK_FP32_TFLOPS,
                                 "unit": "TFLOP/s", "frac": fl * B * T / (np.mean(ms) * 1e-3) / 1e12 / PEAK_FP32_TFLOPS,
                                 "traffic": None, "flop_per_sample": fl})(
            FLOP_PER_SAMPLE if a.workload == "diffdel" else 2 * (13 * 32 + 32 + 3 * (32 * 13 * 32 + 32 * 32) + 32)),
        "checks": {"deterministic": bool(torch.equal(y, y0)),
                   "stream0_vs_oracle_max_abs": None if ref0 is None else float(np.abs(y[0, 0].cpu().numpy() - ref0).max())}}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4096, help="segments per GPU")
    ap.add_argument("--samples", type=int, default=65536, help="samples per segment")
    ap.add_argument("--variant", default="auto", choices=["auto", "mfma2", "mfma", "valu", "f16x3"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the opt-in f16x3 kernel leg")
    ap.add_argument("--workload", default="gru", choices=["gru", "diffdel", "tcn"],
                    help="gru = BASELINE configs[1] (the headline metric); diffdel = configs[2]; tcn = configs[3]")
    a = ap.parse_args()

    import ntm_amd
    from ntm_amd import distributed as D, weights
    from ntm_amd.model import esr_sums, esr_dcpre_sums, MRSTFTLoss, ESR_EPS

    if a.workload != "gru":
        return side_workload(a)
    assert torch.cuda.is_available(), "bench.py needs the MI355X"
    # NTM_DIST_BACKEND=gloo lets several ranks share one GPU (dev box with a single MI355X): it exercises
    # the N>1 control flow; the driver's multi-GPU runs use the default, nccl (= RCCL), one rank per GPU.
    backend = os.environ.get("NTM_DIST_BACKEND", "nccl")
    if backend != "nccl":
        os.environ["LOCAL_RANK"] = str(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
    rank, world, local = D.init_from_env(backend)
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    B, T = a.batch, a.samples

    model = ntm_amd.harness.build_model(weights.W_GRU, device=dev)
    model.kernel_variant = a.variant
    x = synth_input(B, T, dev, seed=1234 + rank)
    gold = None
    if rank == 0 and T == 65536:
        gold = np.load(os.path.join(ROOT, "tests", "golden", "g6_long_65536.npz"))
        x[0, 0] = torch.from_numpy(gold["x"][0, 0]).to(dev)
    INIT_LEN = 1024

    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    side = torch.cuda.Stream(device=dev)       # the loss leg (ESR sums + the job-wide reduction) runs here

    def one_pass(target, evs=(ev0, ev1)):
        """One step: model.predict(x) (code/model.py:218-246, unrolled so the events bracket the main launch) and,
        when a target is given, the ESR sums and this rank's loss scalars (code/test-model.py:386-398) on the side stream -- the
        next step's launches do not wait for them (the K timed steps are bracketed by synchronisation on both
        sides; inside, the HBM-bound loss leg of step k overlaps the compute-bound GRU launch of step k+1).
        -> (y, this rank's loss scalars (device) or None); the job-wide all-reduce of all K steps' scalars is ONE RCCL
        call issued before the closing synchronisation"""
        model.initialize_hidden()
        model.warm_start()
        model.hidden = model.hidden.expand(1, B, 64).contiguous()
        evs[0].record()
        y = model.forward(x)
        evs[1].record()
        pend = None
        if target is not None:
            side.wait_event(evs[1])
            y.record_stream(side)
            with torch.cuda.stream(side):
                s = esr_sums(y, target, skip=INIT_LEN)
                n = T - INIT_LEN
                pend = D.local_loss_sums((s[:, 0] / n) / (s[:, 1] / n + ESR_EPS), s)
        return y, pend

    target = None
    for _ in range(max(a.warmup, 0)):
        y, pend = one_pass(target)
        if pend is not None:
            torch.cuda.current_stream().wait_stream(side)
            D.reduce_many([pend])
        if target is None:
            target = y.clone()
    if target is None:                      # --warmup 0: still need the determinism target
        target = one_pass(None)[0].clone()

    step_evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pends = []
    for k in range(a.steps):
        y, pend = one_pass(target, step_evs[k])
        pends.append(pend)
    torch.cuda.current_stream().wait_stream(side)
    results = D.reduce_many(pends)          # ONE all-reduce of the K x 4 scalars (RCCL), inside the timed region
    torch.cuda.synchronize()
    D.barrier()
    elapsed = D.max_over_ranks(time.perf_counter() - t0, dev)
    res = results[-1] if results else None
    steps_identical = all(r["mean_segment_loss"] == results[0]["mean_segment_loss"] for r in results)
    kern_ms = [e0.elapsed_time(e1) for e0, e1 in step_evs]

    # ---- opt-in kernel variant, reported beside the headline (never part of `value`): the f16x3 GEMV
    #      engine, checked over the whole batch against the exact-fp32 pass
    extra = {}
    if a.variant in ("auto", "mfma2") and not a.no_extra:
        model.kernel_variant = "f16x3"
        ms2 = []
        for i in range(2 + 5):
            y2, _ = one_pass(None)
            torch.cuda.synchronize()
            if i >= 2:
                ms2.append(ev0.elapsed_time(ev1))
        diff = (y2 - target).abs().max().item()
        s2 = esr_sums(y2, target, skip=INIT_LEN).sum(dim=0)
        k2 = float(np.mean(ms2)) / 1e3
        extra["f16x3"] = {
            "what": "same kernel with W.h evaluated as three fp16 hi/lo MFMA products, fp32 accumulate (opt-in)",
            "kernel_ms": 1e3 * k2, "samples_per_s_kernel": B * T / k2,
            "max_abs_diff_vs_exact_fp32_whole_batch": diff,
            "esr_vs_exact_fp32_whole_batch": float(s2[0] / (s2[1] + ESR_EPS * B * (T - INIT_LEN)))}
        # yardstick: two EXACT fp32 kernels that only differ in summation order (MFMA K order)
        model.kernel_variant = "mfma"
        y3, _ = one_pass(None)
        extra["f16x3"]["yardstick_max_abs_diff_between_two_exact_fp32_kernels"] = (y3 - target).abs().max().item()
        del y3
        if gold is not None:
            extra["f16x3"]["stream0_vs_reference_max_abs"] = float(
                np.abs(gold["y"][0, 0] - y2[0, 0].cpu().numpy()).max())
        model.kernel_variant = a.variant
        # the loss dict that follows the path in code/test-model.py:250-254 (never part of `value` beyond the ESR
        # sums the timed step already contains): ESR, DCPreESR and MultiSTFT over the whole batch, f16x3 output
        # against the exact-fp32 output
        lm = {}
        for name, fn in (("ESR", lambda: esr_sums(y2, target, skip=INIT_LEN)),
                         ("DCPreESR", lambda: esr_dcpre_sums(y2, target, skip=INIT_LEN)),
                         ("MultiSTFT", lambda: MRSTFTLoss().per_segment(y2, target, skip=INIT_LEN))):
            for _ in range(2):
                ev0.record(); r = fn(); ev1.record(); torch.cuda.synchronize()
            lm[name + "_ms"] = ev0.elapsed_time(ev1)
        lm["MultiSTFT_f16x3_vs_exact_mean"] = float(r.mean())
        extra["loss_pass"] = lm
        del y2
    # achievable HBM rate on this box (state it beside the 8 TB/s vendor peak): device copy of the
    # input batch, read + write bytes over the event time
    cp = torch.empty_like(x)
    for _ in range(3):
        ev0.record(); cp.copy_(x); ev1.record(); torch.cuda.synchronize()
    hbm_copy_gbs = 2.0 * x.numel() * 4 / (ev0.elapsed_time(ev1) * 1e-3) / 1e9
    del cp
    if rank != 0:
        return
    total_samples = float(B) * T * world * a.steps
    kern_s = float(np.mean(kern_ms)) / 1e3
    # exact-fp32 kernels: algorithmic flops against the fp32 matrix peak.  --variant f16x3 (opt-in): the MFMA
    # flops it actually executes (three fp16 products per W.h term) against the dense fp16 MFMA peak.
    flop_per_sample, peak_tflops, dtype = FLOP_PER_SAMPLE, PEAK_FP32_TFLOPS, "f32"
    if a.variant == "f16x3":
        flop_per_sample, peak_tflops, dtype = 3 * 2 * 12288 + 2 * (192 + 64), 2500.0, "f16x3 products, f32 accumulate"
    tflops = flop_per_sample * B * T / kern_s / 1e12
    hbm_gbs = BYTES_PER_SAMPLE * B * T / kern_s / 1e9
    checks = {"esr_vs_first_pass": res["mean_segment_loss"] if res else None, "segments": res["segments"] if res else None,
              "every_timed_step_same_loss": steps_identical}
    if gold is not None:
        yg = y[0, 0].cpu().numpy()
        e = gold["y"][0, 0] - yg
        checks["stream0_vs_reference_max_abs"] = float(np.abs(e).max())
        checks["stream0_vs_reference_esr"] = float((e[INIT_LEN:] ** 2).mean() /
                                                   ((gold["y"][0, 0][INIT_LEN:] ** 2).mean() + ESR_EPS))
    # HBM bytes per launch from the PMC counters (collected in separate rocprofv3 --pmc passes, gfx950
    # FETCH_SIZE correction applied; see profiles/*pmc_traffic*.json) -- only for the matching workload
    traffic = None
    if (B, T) == (4096, 65536) and a.variant in ("auto", "mfma2"):
        import glob
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic_mfma2*.json")))
        if files:
            traffic = json.load(open(files[-1]))["hbm_bytes_per_launch_corrected"]
    out = {
        "metric": "audio samples/sec (44.1 kHz) GRU-HS[64], batch=4096x65536",
        "value": total_samples / elapsed, "unit": "samples/s", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
        "config": {"workload": f"GRU-HS[64] CHOWTAPE weights, {B} segments x {T} samples fp32 per GPU, "
                               f"predict (warm-start + persistent GRU kernel) + ESR sums on a side stream under the next step's launch + one all-reduce of the per-step loss scalars",
                   "segments_per_gpu": B, "samples_per_segment": T, "kernel": a.variant,
                   "parallelism": f"streams sharded over {world} GPU(s), no data-path collective"},
        "realtime_factor": total_samples / elapsed / FS,
        "roofline": {"bound": "mfma", "achieved": tflops, "peak": peak_tflops, "unit": "TFLOP/s",
                     "frac": tflops / peak_tflops, "traffic": traffic,
                     "kernel": {"auto": "gru_mfma2_kernel", "mfma2": "gru_mfma2_kernel", "mfma": "gru_mfma_kernel",
                                "valu": "gru_valu_kernel", "f16x3": "gru_mfma2_kernel<f16x3>"}[a.variant],
                     "kernel_ms": 1e3 * kern_s, "flop_per_sample": flop_per_sample,
                     "hbm": {"achieved": hbm_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                             "frac": hbm_gbs / PEAK_HBM_GBS, "bytes_per_sample": BYTES_PER_SAMPLE,
                             "copy_kernel_measured": hbm_copy_gbs}},
        "checks": checks,
    }
    if extra:
        out["other_kernels"] = extra
    if world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(weights.W_GRU)
        out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
