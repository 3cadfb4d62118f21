#!/usr/bin/env python3
"""Headline benchmark: audio samples/s through GRU-HS[64] (CHOWTAPE weights), batch 4096 segments
x 65 536 samples fp32 per GPU (BASELINE.json configs[1]); weak scaling over N GPUs (configs[4]:
N x 4096 segments, sharded by stream, one RCCL all-reduce of the ESR scalars).

A "step" = one pass of the hot path over the rank's batch, inputs resident in HBM:
    warm-start state (1024 zero samples, B=1: a pure function of the weights, computed by the kernel in the first pass
    and kept per parameter version, ntm_amd.model warm_cache) -> persistent GRU kernel over [B,T] with the per-stream ESR
    sums against a resident target accumulated in its output flush (RNN.forward_esr; --esr pass: a separate streaming
    pass) -> all-reduce of 4 fp64 scalars.
The target is NOT the model's output: target = 0.9 * (output of the first, untimed pass) + 0.02 * x, built once on the device,
so every timed step accumulates non-zero sums; after the timed region the fused sums of scattered streams are compared with
the oracle's esr_sums on the same rows (rel 1e-9), the last output with the first pass bit for bit (full-size determinism),
and stream 0 carries the input of golden G6 so the result is also checked against the REFERENCE's own output on 65 536 samples.

    python bench.py [--gpus N --steps K --warmup W] [--scaling weak|strong]
Rank 0 prints ONE compact JSON line (< 4 KB: headline, roofline, cpu_baseline, checks, one entry per side workload) as the
LAST line on stdout and writes the full record to --detail-file (default bench_detail.json beside this file; also on stderr
behind `bench_detail: `).

N > 1 runs one process per GPU over RCCL.  Either the caller provides the ranks (`python -m torch.distributed.run
--nproc-per-node N ... bench.py --gpus N`: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment), or --
when WORLD_SIZE is NOT set -- `python bench.py --gpus N` starts them itself: the parent process, which never touches
the GPU, spawns N fresh interpreters of this file with the rank environment, relays rank 0's JSON line and exits
non-zero if any rank fails (spawn_ranks below).  `--scaling weak` (default): --batch segments PER GPU (N = 8 is
BASELINE configs[4], 8 x 4096 = 32 768 segments); `--scaling strong`: --total-batch segments (default 32 768) split
over the N ranks by distributed.shard_range.
"""
import argparse
import json
import os
import socket
import subprocess
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


def cpu_baseline(w_name, small=False):
    """Reference CPU path timed on this host: stock torch.nn.GRU + Linear on CPU -- the modules the
    reference builds at code/model.py:44-45 -- with the CHOWTAPE weights under inference_mode, on the two shapes the
    survey names: 16 x 8192 (BASELINE configs[0], the reference's best CPU shape; `value`) and 64 x 8192; a bounded
    sample of the workload (the full 4096 x 65 536 would take ~30 min), on the cores this process may use.  The C
    oracle (OpenMP over streams) is timed beside it."""
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
    f(rng.uniform(-0.5, 0.5, (16, 512)).astype(np.float32), np.repeat(oracle.warm_state(w), 16, 0))   # thread-pool warm-up
    shapes = {}
    for B, T, reps in (((16, 8192, 1),) if small else ((16, 8192, 3), (64, 8192, 1))):
        h0 = np.repeat(oracle.warm_state(w), B, 0)
        x = rng.uniform(-0.5, 0.5, (B, T)).astype(np.float32)
        best = None
        for _ in range(reps):
            t0 = time.perf_counter()
            f(x, h0)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        shapes[f"{B}x{T}"] = {"value": x.size / best, "seconds": best, "repetitions": reps}
    res = {"value": shapes["16x8192"]["value"], "unit": "samples/s", "cores": cores, "kind": "port",
           "sample": f"16 x 8192 samples (BASELINE configs[0] shape) of the same workload, best of {1 if small else 3}, torch.nn.GRU+Linear "
                     f"on CPU under inference_mode, {cores} threads" + ("" if small else "; 64 x 8192 beside it"),
           "shapes": shapes}
    xo = rng.uniform(-0.5, 0.5, (2 * cores, 8192)).astype(np.float32)
    t0 = time.perf_counter()
    oracle.gru_forward(w, xo, threads=cores)
    dt = time.perf_counter() - t0
    res["oracle_c"] = {"value": xo.size / dt, "unit": "samples/s", "cores": cores,
                       "sample": f"{xo.shape[0]}x8192 samples, C restatement, OpenMP over streams"}
    return res


TCN_FLOP_PER_SAMPLE = 2 * (13 * 32 + 32 + 3 * (32 * 13 * 32 + 32 * 32) + 32)     # 43 488 FMA per sample


def delay_trajectories(B, T, dev, max_delay):
    """d_b[n] = fs (0.0271 + a_b sin(2 pi w_b n/fs + psi_b) + 0.0005 sin(2 pi 23 n/fs)) samples,
    a_b in U(0.002, 0.005) s, w_b in U(0.5, 2) Hz (wow) plus a 23 Hz flutter line; clamped to [0, max_delay]."""
    g = torch.Generator(device=dev)
    g.manual_seed(77)
    amp = 0.002 + 0.003 * torch.rand(B, 1, generator=g, device=dev)
    wv = 0.5 + 1.5 * torch.rand(B, 1, generator=g, device=dev)
    psi = 2 * np.pi * torch.rand(B, 1, generator=g, device=dev)
    n = torch.arange(T, device=dev, dtype=torch.float32).unsqueeze(0)
    d = torch.empty(B, T, device=dev)
    for b0 in range(0, B, 256):
        sl = slice(b0, min(B, b0 + 256))
        d[sl] = FS * (0.0271 + amp[sl] * torch.sin(2 * np.pi * wv[sl] * n / FS + psi[sl])
                      + 0.0005 * torch.sin(2 * np.pi * 23 * n / FS))
    return d.clamp_(0, max_delay).unsqueeze(1)


_LIVE_CACHE = {}


def profile_traffic(pattern, key):
    """(value, source note) from the newest tracked profiles/<pattern> (rocprofv3 --pmc passes of that leg's own command,
    tools/attic/gpu_r04_pmc.sh; not re-measured in this run), or (None, None)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    if not files:
        return None, None
    j = json.load(open(files[-1]))
    return j.get(key), ("profiles/" + os.path.basename(files[-1]) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this leg's own "
                        "command, gfx950 correction applied; not re-measured in this run)")


def measure_workload(workload, B, T, steps, warmup, check, dev, threads=1, delay_mode="auto", live=False):
    """One single-GPU workload beside the headline, `steps` timed passes over a resident batch of B distinct streams:
      "diffdel"  BASELINE configs[2]: DiffDelGRU-HS[64] (CHOWTAPE_WOWFLUTTER weights, D = 1847) predict
      "tcn"      BASELINE configs[3]: the TCN contrast point
      "gru"      the headline's model at another batch size -- the per-GPU shapes of configs[4]'s strong-scaling legs
    -> dict with the whole-pass rate, the dominant kernel's event-timed launch duration and roofline, determinism of the
    output and -- when `check` (the CPU leg; the oracle is the checker, never the thing measured) -- scattered streams
    against the C oracle over the whole sequence."""
    import ntm_amd
    from ntm_amd import weights
    x = synth_input(B, T, dev, seed=1234)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    kev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    d = None
    if workload == "diffdel":
        model = ntm_amd.harness.build_model(weights.W_DIFFDEL, max_delay_seconds=0.0335, device=dev)   # D = 1847
        model.delay_mode = delay_mode
        d = delay_trajectories(B, T, dev, model.max_delay)
        run = lambda: model.predict(x, d, _events=kev)[0]                    # noqa: E731
        name, bytes_per_sample, fl = "DiffDelGRU-HS[64] CHOWTAPE_WOWFLUTTER weights, D=1847", 16, FLOP_PER_SAMPLE
        kernel = "gru_mfma2_kernel"
    elif workload == "tcn":
        model = ntm_amd.TCN().to(dev)

        def run():
            kev[0].record(); y = model(x); kev[1].record()
            return y
        name, bytes_per_sample, fl = "TCN 4x(k13, dil 1/10/100/1000, 32 ch) seeded weights", 8, TCN_FLOP_PER_SAMPLE
        kernel = "tcn_forward (first block + 3 MFMA blocks, output conv fused)"
    else:
        model = ntm_amd.harness.build_model(weights.W_GRU, device=dev)

        def run():      # RNN.predict (code/model.py:218-246) unrolled so that the events bracket the main launch
            model.initialize_hidden()
            model.warm_start()
            model.hidden = model.hidden.expand(1, B, 64).contiguous()
            kev[0].record(); y = model.forward(x); kev[1].record()
            return y
        name, bytes_per_sample, fl = "GRU-HS[64] CHOWTAPE weights", 8, FLOP_PER_SAMPLE
        kernel = "gru_mfma2_kernel" if B > 1024 else "gru_lat_kernel"
    ref, rows = None, sorted({r for r in (0, 17, B // 2 + 1, B - 1) if 0 <= r < B})
    if check:
        import oracle
        xs = x[rows, 0].cpu().numpy()
        sd = lambda nm: oracle.Weights.from_state_dict({k: v.numpy() for k, v in weights.load_state_dict(nm).items()})  # noqa: E731
        if workload == "diffdel":
            ref = oracle.diffdel_predict(sd(weights.W_DIFFDEL), xs, d[rows, 0].cpu().numpy(), model.max_delay, threads=threads)[0]
        elif workload == "tcn":
            ref = oracle.tcn_forward(model.packed_params().cpu().numpy(), len(model.dilations), model.channels,
                                     model.kernel_size, model.dilations, xs, threads=threads)
        else:
            ref = oracle.gru_predict(sd(weights.W_GRU), xs, threads=threads)[0]
    for _ in range(max(warmup, 1)):
        y0 = run()
    # two more untimed passes: a timed pass allocates its output while the previous one is still alive, and the determinism
    # target y0 stays alive too -- THREE output blocks must already sit in the caching allocator, or the first timed pass
    # carries a hipMalloc of B x T x 4 bytes between its events (seen once: 8.6 GB took 245 ms on one box, gru_B32768 509 ms
    # instead of 427)
    y = run()
    y = run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ms, k1, k2 = [], [], []
    for _ in range(steps):
        ev0.record(); y = run(); ev1.record(); torch.cuda.synchronize()
        ms.append(ev0.elapsed_time(ev1))
        k1.append(kev[0].elapsed_time(kev[1]))
        if workload == "diffdel":
            k2.append(kev[1].elapsed_time(kev[2]))
    elapsed = time.perf_counter() - t0
    # dominant kernel: the GRU launch (DiffDelGRU: the delay line adds 3 flop/sample) resp. the TCN forward
    # (its five launches cannot be separated by events: profiles/ has the rocprofv3 split)
    ksec = float(np.mean(k1)) * 1e-3
    roof = {"bound": "mfma", "achieved": fl * B * T / ksec / 1e12, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
            "frac": fl * B * T / ksec / 1e12 / PEAK_FP32_TFLOPS, "traffic": None, "flop_per_sample": fl,
            "kernel": kernel, "kernel_ms": 1e3 * ksec}
    if workload == "gru" and T == 65536:
        roof["traffic"], roof["traffic_source"] = profile_traffic(f"*pmc_traffic_gru_B{B}.json", "hbm_bytes_per_launch_corrected")
        roof["algorithmic_bytes"] = 8.0 * B * T
    if workload == "tcn":
        # what the forward MOVES, not the 8 B/sample a fully fused network would: every block writes its 32-channel activations
        # (128 B/sample) and the next one reads them back -- about 780 B/sample, ~100 x the algorithmic bytes.  The blocks are
        # matrix-pipe bound (0.86 of the fp32 peak), so the traffic costs little TIME (the first block, the only HBM-bound
        # launch, is 4 % of the forward), but it is what the record must show.
        L_ = ntm_amd._lib.lib()
        moved, src = None, None
        if live and T == 65536:            # two rocprofv3 --pmc child passes of `bench.py --workload tcn` at this shape, in this run
            chunks = -(-B // int(L_.ntm_tcn_chunk_streams(B, T, 32)))
            tr, src = live_traffic(r"tcn_", ["--workload", "tcn", "--batch", str(B), "--samples", str(T)], per_forward=(r"tcn_first_d1_kernel", chunks))
            moved = None if tr is None else tr / (float(B) * T)
        if moved is not None:
            _LIVE_CACHE["tcn_bytes_per_sample"] = (moved, src)
        elif "tcn_bytes_per_sample" in _LIVE_CACHE and T == 65536:       # the other TCN legs: this run's own per-sample figure
            moved, src = _LIVE_CACHE["tcn_bytes_per_sample"]
            src += "; per-sample figure of this run's 4096 x 65536 forward scaled to this batch"
        if moved is None:
            why = src
            moved, src = profile_traffic("*pmc_traffic_tcn_forward.json", "bytes_per_sample_moved")
            if src and why:
                src += f"; live passes: {why}"
            if src:
                src += "; per-sample figure of the 4096 x 65536 forward scaled to this batch"
        roof["bytes_per_sample_if_fused"] = 8
        roof["algorithmic_bytes"] = 8.0 * B * T
        roof["bytes_per_sample_moved"] = moved
        if moved is not None and T == 65536:
            roof["traffic"], roof["traffic_source"] = moved * B * T, src
            bytes_per_sample = moved
        roof["scratch_bytes"] = 4 * int(L_.ntm_tcn_scratch_floats(B, T, 32))
        roof["stream_chunk"] = int(L_.ntm_tcn_chunk_streams(B, T, 32))
    if workload == "diffdel":
        fused = model.delay_mode != "two_pass" and model.kernel_variant == "auto" and B > 1024
        if fused:
            # ONE launch for the step: the delay line interpolates the kernel's own pre_d output inside the y-tile
            # housekeeping (taps through L2), so the step's HBM traffic is its 16 algorithmic bytes per sample; the events
            # bracket the fused launch + the small buffer-update launch behind it.  The two-pass form (GRU launch, then
            # the streaming delay pass: 12 more bytes per sample) is timed beside it for the A/B.
            roof["kernel"] = kernel = "gru_mfma2_kernel<FUSE: GRU + head + delay line> (+ delay_update_kernel)"
            roof["hbm_bytes_per_sample"] = 16
            if T == 65536:                       # PMC traffic of the fused kernel (measurement log 4 K2f: ~1.25 x the algorithmic bytes,
                import glob                      # the taps come back from beyond L2)
                why = None
                ypn = 4 if (B + 15) // 16 > torch.cuda.get_device_properties(dev).multi_processor_count else 16      # launch_gru_mfma2
                if live:                         # two rocprofv3 --pmc child passes of `bench.py --workload diffdel`, in this run
                    roof["traffic"], why = live_traffic(rf"gru_mfma2_kernel<true, false, 0, 0, {ypn}, true, false, false>",
                                                        ["--workload", "diffdel", "--batch", str(B), "--samples", str(T)])
                    roof["traffic_source"] = why
                files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic_diffdel_fused.json"))) if B == 4096 else []
                if roof["traffic"] is None and files:
                    roof["traffic"] = json.load(open(files[-1]))["hbm_bytes_per_launch_corrected"]
                    roof["traffic_source"] = ("profiles/" + os.path.basename(files[-1]) + " (rocprofv3 --pmc passes of `bench.py "
                                              "--workload diffdel`; not re-measured in this run" + (f"; live passes: {why}" if why else "") + ")")
                roof["algorithmic_bytes"] = 16.0 * B * T
            model.delay_mode = "two_pass"
            t1, g1, d1 = [], [], []
            for i in range(1 + steps):
                ev0.record(); y2 = run(); ev1.record(); torch.cuda.synchronize()
                if i:
                    t1.append(ev0.elapsed_time(ev1)); g1.append(kev[0].elapsed_time(kev[1])); d1.append(kev[1].elapsed_time(kev[2]))
            dsec = float(np.mean(d1)) * 1e-3
            alg = 12.0 * B * T + 8.0 * B * model.diffdel.max_delay
            roof["two_pass"] = {"device_ms_per_step": float(np.mean(t1)), "gru_kernel_ms": float(np.mean(g1)),
                                "delay_pass_ms": 1e3 * dsec, "delay_pass_kernel": "delay_apply_kernel + delay_update_kernel",
                                "delay_pass_hbm": {"achieved": alg / dsec / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                                   "frac": alg / dsec / 1e9 / PEAK_HBM_GBS, "bytes_per_sample": 12},
                                "identical_output": bool(torch.equal(y2, y))}
            model.delay_mode = delay_mode
            del y2
        else:
            # K2 as a separate streaming pass: pre_d and d read once, y written once = 12 algorithmic bytes per sample
            # (the two gathered taps come from pre_d, i.e. from the same bytes); the carried buffer adds 8 D bytes per stream
            dsec = float(np.mean(k2)) * 1e-3
            alg = 12.0 * B * T + 8.0 * B * model.diffdel.max_delay
            roof["delay_line"] = {"bound": "hbm", "kernel": "delay_apply_kernel + delay_update_kernel", "kernel_ms": 1e3 * dsec,
                                  "achieved": alg / dsec / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                  "frac": alg / dsec / 1e9 / PEAK_HBM_GBS, "bytes_per_sample": 12,
                                  "frac_at_16B_per_sample": 16.0 * B * T / dsec / 1e9 / PEAK_HBM_GBS}
    res = {"workload": f"{name}, {B} segments x {T} samples fp32", "value": B * T * steps / elapsed, "unit": "samples/s",
           "steps": steps, "warmup": max(warmup, 1), "ms_per_step": 1e3 * elapsed / steps,
           "device_ms_per_step": float(np.mean(ms)), "bytes_per_sample": bytes_per_sample,
           "kernel": kernel, "kernel_ms": 1e3 * ksec, "roofline": roof,
           "checks": {"deterministic": bool(torch.equal(y, y0)), "streams_checked": rows if ref is not None else [],
                      "samples_each": T, "tolerance": 1e-5,
                      "vs_oracle_max_abs": None if ref is None else float(np.abs(y[rows, 0].cpu().numpy() - ref).max())}}
    del x, y, y0, d, model
    torch.cuda.empty_cache()
    return res

CLI_WEIGHTS = "GRU-HS[64]-L[DCPreESR]-DS[ReelToReel_Dataset_MiniPulse100_CHOWTAPE_WOWFLUTTER]_BEST"
CLI_DATASET = "ReelToReel_Dataset_MiniPulse100_CHOWTAPE_WOWFLUTTER"


def cli_dataset(root, n_seg, L, dev):
    """The dataset the CLI line runs on, in the on-disk format of the reference's Zenodo sets (code/dataset.py:129-293): one
    stereo float32 WAV pair `<root>/audio/<name>/Test/{input,target}_1_.wav` (channel 0 audio, channel 1 the pilot track) of
    n_seg x L samples, and its `trajectory_1_.npy` side-car (the pickled dict DelayAnalyzer caches, utilities.py:327-335) with a
    wow-and-flutter trajectory around 27 ms like the toy data.  Synthesised on the device, written with scipy.
    -> (dataset directory, dict of host arrays for the CPU port)."""
    from scipy.io import wavfile
    from ntm_amd.feeder import write_sidecar
    N = n_seg * L + 777
    g = torch.Generator(device=dev)
    g.manual_seed(2025)
    n = torch.arange(N, device=dev, dtype=torch.float32)
    f0 = 220.0 * (1.0 + 0.3 * torch.sin(2 * np.pi * 0.2 * n / FS))
    phase = torch.cumsum(f0.double(), 0) * (2 * np.pi / FS)
    audio = (0.4 * torch.sin(phase).float() * (0.6 + 0.4 * torch.sin(2 * np.pi * 0.5 * n / FS))
             + 0.02 * torch.randn(N, generator=g, device=dev))
    tgt = 0.5 * torch.tanh(2.0 * torch.roll(audio, int(0.0271 * FS)))
    traj = (0.0271 + 0.004 * torch.sin(2 * np.pi * 1.3 * n / FS) + 0.0005 * torch.sin(2 * np.pi * 23 * n / FS)).double().cpu().numpy()
    a, t = audio.cpu().numpy(), tgt.cpu().numpy()
    del n, f0, phase, audio, tgt
    d = os.path.join(root, "audio", CLI_DATASET, "Test")
    os.makedirs(d)
    pilot = np.zeros(N, np.float32)
    wavfile.write(os.path.join(d, "input_1_.wav"), FS, np.stack([a, pilot], 1))
    wavfile.write(os.path.join(d, "target_1_.wav"), FS, np.stack([t, pilot], 1))
    peaks = np.arange(1000, N - 10000, 4410)
    meta = {"reconstruction_percentage": 0.0, "wiggle_percentage": 0.0}
    write_sidecar(os.path.join(d, "trajectory_1_.npy"), peaks, peaks + int(0.0271 * FS), traj, meta, meta)
    return os.path.join(root, "audio", CLI_DATASET), {"audio": a, "target": t, "traj": traj}


def cli_cpu_port(host, L, n_sub, init_len, max_delay_n, cores):
    """CPU leg (the oracle and stock torch are the things timed here, never the product): the SAME command in the
    reference's terms -- torch.nn.GRU + Linear under inference_mode (code/model.py:44-45,81-82), the delay line, the three
    losses -- on the first `n_sub` segments of the dataset, stage by stage, on `cores` threads."""
    import oracle
    from ntm_amd import weights
    sd = weights.load_state_dict(CLI_WEIGHTS)
    w = oracle.Weights.from_state_dict({k: v.numpy() for k, v in sd.items()})
    torch.set_num_threads(cores)
    X = np.stack([host["audio"][k * L:(k + 1) * L] for k in range(n_sub)])
    Tg = np.stack([host["target"][k * L:(k + 1) * L] for k in range(n_sub)])
    Dt = np.stack([host["traj"][k * L:(k + 1) * L].astype(np.float32) * np.float32(FS) for k in range(n_sub)])
    st = {}
    t0 = time.perf_counter()
    y, _ = oracle.torch_gru_port(w)(X, np.repeat(oracle.warm_state(w), n_sub, 0))
    st["predict_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    yd, _ = oracle.delay_forward(y, Dt, np.zeros((n_sub, max_delay_n), np.float32))
    st["apply_delay_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    esr = oracle.esr_per_segment(yd, Tg, init_len)
    st["ESR_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    sdc = oracle.esr_dcpre_sums(yd, Tg, init_len)
    st["DCPreESR_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    mr = oracle.mrstft_per_segment(yd, Tg, init_len)
    st["MultiSTFT_s"] = time.perf_counter() - t0
    n = L - init_len
    total = sum(st.values())
    return {"kind": "port", "cores": cores, "sample": f"the first {n_sub} of the dataset's segments ({n_sub} x {L} samples), every stage of the command "
            "after the decode: torch.nn.GRU+Linear on CPU under inference_mode, then the C oracle's delay line / ESR / DCPreESR and "
            "its torch.stft MultiSTFT", "stages_s": st, "value": n_sub * L / total, "unit": "samples/s",
            "losses": {"ESR": float(np.mean(esr)), "DCPreESR": float(np.mean((sdc[:, 0] / n) / (sdc[:, 1] / n + 1e-5))),
                       "MultiSTFT": float(np.mean(mr))}}


def cli_workload(dev, check, n_seg=128, L=441000):
    """`other_workloads.cli`: the reference's canonical evaluation command (scripts/test-model-loss.sh:22,57-63: 10-second
    segments, --ADD_DELAY, --COMPUTE_LOSS = ESR + DCPreESR + MultiSTFT, code/test-model.py:250-254) through tools/test_model.py,
    end to end from WAV files on disk, on a synthetic dataset in the reference's on-disk format; stage times from HIP events
    / host clocks inside the command (tools/test_model.py StageTimes), the second of two runs (the first pays library load,
    allocator growth and the page cache).  -> dict."""
    import importlib.util
    import shutil
    import tempfile
    spec = importlib.util.spec_from_file_location("ntm_cli_bench", os.path.join(ROOT, "tools", "test_model.py"))
    cli = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cli)
    tmp = tempfile.mkdtemp(prefix="ntm_cli_", dir="/tmp")
    cwd = os.getcwd()
    try:
        t0 = time.perf_counter()
        ds_dir, host = cli_dataset(tmp, n_seg, L, dev)
        t_make = time.perf_counter() - t0
        os.makedirs(os.path.join(tmp, "scripts"))
        os.chdir(os.path.join(tmp, "scripts"))           # the command resolves ../audio/<DATASET> like the reference (:107,147)
        argv = ["--MODEL", "GRU", "--WEIGHTS", CLI_WEIGHTS, "--DATASET", CLI_DATASET, "--SUBSET", "Test", "--NO_SHUFFLE",
                "--SEGMENT_LENGTH", str(L), "--ADD_DELAY", "--COMPUTE_LOSS", "--NO_CACHE", "--NO_EXAMPLE"]
        import contextlib
        import io
        runs = []
        for _ in range(2):
            prof = {}
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with contextlib.redirect_stdout(io.StringIO()):
                losses = cli.main(argv, profile=prof)
            torch.cuda.synchronize()
            prof["command_s"] = time.perf_counter() - t0
            runs.append((prof, losses))
        prof, losses = runs[-1]
        gpu_ms = sum(v for k, v in prof.items() if k.endswith("_ms") and k != "h2d_ms")
        # (a device-resident set: the decode is the host-to-device copy, the per-batch stage a device-to-device gather)
        stages = {"decode (WAV -> " + ("device" if prof.get("resident") else "pinned host") + ", side-car)": prof["decode_s"] * 1e3,
                  "gather (device to device)" if prof.get("resident") else "h2d": prof.get("h2d_ms", 0.0),
                  "predict": prof.get("predict_ms", 0.0) + prof.get("predict_streamed_ms", 0.0), "apply_delay": prof.get("apply_delay_ms", 0.0),
                  "ESR": prof["ESR_ms"], "DCPreESR": prof["DCPreESR_ms"], "MultiSTFT": prof.get("MultiSTFT_ms", 0.0)}
        bound = max(stages, key=stages.get)
        samples = float(prof["segments"]) * prof["segment_length"]
        res = {"workload": f"tools/test_model.py {' '.join(argv[:-2])}: {prof['segments']} segments x {L} samples (10 s at 44.1 kHz) from "
                           f"a stereo float32 WAV pair + trajectory side-car on disk ({2 * 8 * samples / 1e9:.2f} GB), GRU-HS[64] predict, delay "
                           "line, ESR + DCPreESR + MultiSTFT, mean over segments",
               "reference_command": "scripts/test-model-loss.sh:57-63 (SEGMENT_LENGTH :22)",
               "value": samples / prof["command_s"], "unit": "samples/s", "command_s": prof["command_s"],
               "first_run_command_s": runs[0][0]["command_s"], "dataset_synthesis_s_untimed": t_make,
               "stages_ms": stages, "dataset_resident_on_device": bool(prof.get("resident")),
               "batch_copy_GBps": (prof.get("h2d_bytes", 0) / 1e9) / (prof["h2d_ms"] * 1e-3) if prof.get("h2d_ms") else None,
               "loss_loop_s": prof.get("loss_loop_s"), "gpu_kernel_ms": gpu_ms,
               "gpu_busy_fraction_of_loss_loop": gpu_ms * 1e-3 / prof["loss_loop_s"] if prof.get("loss_loop_s") else None,
               "gpu_busy_fraction_of_command": gpu_ms * 1e-3 / prof["command_s"],
               "bound_by": bound, "losses": {k: float(v) for k, v in losses.items()},
               "kernel": "gru_lat_kernel (128 segments <= NTM_GRU_LAT_MAX_B)" if prof["segments"] <= 1024 else "gru_mfma2_kernel"}
        if check:
            md = int(1.25 * float(host["traj"].max()) * FS)
            init_len = 1 << (int(float(host["traj"].max()) * FS) - 1).bit_length()
            cpu = cli_cpu_port(host, L, 2, init_len, md, _host_threads())
            res["cpu_baseline"] = cpu
            res["speedup_vs_cpu_after_decode"] = (samples / (prof["command_s"] - prof["decode_s"])) / cpu["value"]
        return res
    finally:
        os.chdir(cwd)
        shutil.rmtree(tmp, ignore_errors=True)


COMPACT_LIMIT = 4096             # bytes: the driver parses the LAST stdout line; round 5's 20 KB line was not parsed


def _sig(v, n=6):
    """Floats of the compact line to n significant digits (the detail file keeps full precision)."""
    if isinstance(v, bool) or not isinstance(v, float):
        return v
    return float(f"{v:.{n}g}")


def _leg(v):
    """One `legs` entry of the compact line (~100 bytes): rate, roofline fraction of the dominant kernel with its duration,
    worst scattered stream against the oracle, PMC traffic over algorithmic bytes where both are known."""
    if not isinstance(v, dict) or "error" in v:
        return {"error": str(v.get("error"))[:80]} if isinstance(v, dict) else None
    r = v.get("roofline") or {}
    e = {"value": _sig(v.get("value")), "frac": _sig(r.get("frac"), 4), "kernel_ms": _sig(r.get("kernel_ms", v.get("kernel_ms")), 5),
         "vs_oracle_max_abs": _sig((v.get("checks") or {}).get("vs_oracle_max_abs"), 3)}
    if r.get("traffic") and r.get("algorithmic_bytes"):
        e["traffic_ratio"] = _sig(r["traffic"] / r["algorithmic_bytes"], 4)
    if "command_s" in v:           # the evaluation command end to end: no single kernel, the GPU-busy share instead
        e = {"value": _sig(v["value"]), "command_s": _sig(v["command_s"], 4), "gpu_busy": _sig(v.get("gpu_busy_fraction_of_command"), 3),
             "bound_by": str(v.get("bound_by"))[:40]}
    return {k: x for k, x in e.items() if x is not None}


def compact_line(out, detail_file):
    """The record the driver parses: the headline, the roofline of the dominant kernel, the CPU baseline, the checks and one
    short entry per side workload -- under COMPACT_LIMIT bytes whatever the run attached; everything else is in `detail_file`."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "value_no_warm_cache", "higher_is_better",
            "scaling", "vs_baseline", "dtype", "data")
    c = {k: out[k] for k in keep if k in out}
    cfg = out.get("config", {})
    c["config"] = {k: (cfg[k][:120] if isinstance(cfg[k], str) else cfg[k]) for k in ("workload", "segments_total", "samples_per_segment") if k in cfg}
    r = out.get("roofline") or {}
    cr = {k: _sig(r[k]) if k == "peak" else r[k]
          for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes", "kernel", "kernel_ms") if k in r}
    if isinstance(cr.get("kernel"), str):
        cr["kernel"] = cr["kernel"][:80]
    if r.get("traffic") and r.get("algorithmic_bytes"):
        cr["traffic_ratio"] = _sig(r["traffic"] / r["algorithmic_bytes"], 5)
        cr["traffic_live"] = str(r.get("traffic_source", "")).startswith("measured in this run")
    if "hbm" in r:
        cr["hbm"] = {k: _sig(r["hbm"][k]) for k in ("achieved", "peak", "frac") if k in r["hbm"]}
    g = r.get("aggregate")
    if g and g.get("n_gpus", 1) > 1:
        cr["aggregate"] = {"n_gpus": g["n_gpus"], "achieved": _sig(g["achieved"]), "peak": g["peak"], "frac": _sig(g["frac"]),
                           "slowest_rank": g.get("slowest_rank")}
    c["roofline"] = cr
    if "cpu_baseline" in out:
        b = out["cpu_baseline"]
        c["cpu_baseline"] = {"value": b["value"], "unit": b["unit"], "cores": b["cores"], "kind": b["kind"], "sample": b["sample"][:120]}
        if "oracle_c" in b:
            c["cpu_baseline"]["oracle_c_value"] = _sig(b["oracle_c"]["value"])
    k = out.get("checks") or {}
    ck = {"deterministic": k.get("last_output_equals_first_pass_bitwise"), "every_timed_step_same_loss": k.get("every_timed_step_same_loss"),
          "job_esr": _sig(k.get("job_esr")), "segments": k.get("segments"),
          "stream0_vs_reference_max_abs": _sig(k.get("stream0_vs_reference_max_abs"), 3),
          "streams_vs_oracle_max_abs": _sig((k.get("streams_vs_oracle") or {}).get("max_abs"), 3),
          "esr_sums_max_rel": _sig((k.get("esr_sums_vs_oracle") or {}).get("max_rel"), 3), "tolerance": 1e-5}
    if "vs_oracle_max_abs" in k:                  # a --workload line: measure_workload's own checks
        ck = {"deterministic": k.get("deterministic"), "streams_vs_oracle_max_abs": _sig(k.get("vs_oracle_max_abs"), 3), "tolerance": 1e-5}
    c["checks"] = {n: v for n, v in ck.items() if v is not None}
    ok = out.get("other_kernels") or {}
    eng = {n: {"kernel_ms": _sig(ok[n]["kernel_ms"], 5), "vs_exact_fp32_max_abs": _sig(ok[n]["max_abs_diff_vs_exact_fp32_whole_batch"], 3)}
           for n in ("f16x3", "bf16x3") if n in ok}
    if eng:                                       # the opt-in split engines: beside `value`, never in it
        c["opt_in_engines"] = eng
    ow = out.get("other_workloads")
    if ow:
        c["legs"] = {n: _leg(v) for n, v in ow.items() if n != "note"}
    b = out.get("build") or {}
    if "error" in b:
        c["build"] = {"error": b["error"][:100]}
    elif b:
        c["build"] = {"hip": (b.get("compiler") or {}).get("hip"), "runtime_hip": b.get("runtime_hip"),
                      "library_sha256": (b.get("library_sha256") or "")[:16]}
    for n in ("backend", "rccl_ranks", "ranks"):
        if n in out:
            c[n] = out[n]
    c["detail_file"] = detail_file
    line = json.dumps(c, separators=(",", ":"))
    if len(line) >= COMPACT_LIMIT:                # never again a line the driver cannot parse: shed the optional parts, in this order
        for drop in ("opt_in_engines", "legs", "build", "checks"):
            c.pop(drop, None)
            line = json.dumps(c, separators=(",", ":"))
            if len(line) < COMPACT_LIMIT:
                break
    assert len(line) < COMPACT_LIMIT, len(line)
    return line


def emit(out, detail_path):
    """Rank 0's output: the full record to `detail_path` (and, behind a `bench_detail: ` prefix so that it is not a JSON line,
    to stderr), then the compact line as the LAST line on stdout."""
    full = json.dumps(out)
    name = os.path.relpath(detail_path, ROOT) if os.path.abspath(detail_path).startswith(ROOT + os.sep) else detail_path
    try:
        tmp = detail_path + ".tmp"
        with open(tmp, "w") as f:
            f.write(full + "\n")
        os.replace(tmp, detail_path)
    except OSError as e:                          # a read-only tree must not cost the line: stderr still has the record
        name = f"(not written: {type(e).__name__}); stderr of this run, line `bench_detail: `"
    sys.stderr.write("bench_detail: " + full + "\n")
    sys.stderr.flush()
    # whatever native libraries left in C stdio's buffer goes out FIRST (RCCL prints its version banner with printf; on a pipe
    # that sat in the buffer until exit and landed BEHIND the JSON line -- seen on the MI355X box, round 6)
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except (OSError, AttributeError):
        pass
    sys.stdout.write(compact_line(out, name) + "\n")
    sys.stdout.flush()


def side_workload(a):
    """`--workload diffdel | tcn`: BASELINE configs[2] / configs[3] at the headline's batch as a line of their own
    (single GPU, not the headline metric).  Like the headline line, the JSON carries the roofline of the dominant kernel
    with its event-timed launch duration; for the DiffDelGRU step the HBM roofline of the delay-line pass sits beside it."""
    assert torch.cuda.is_available()
    r = measure_workload(a.workload, a.batch, a.samples, a.steps, a.warmup, not a.no_cpu_baseline, torch.device("cuda", 0),
                         threads=_host_threads(), delay_mode=a.delay_mode)
    emit({
        "metric": f"audio samples/sec (44.1 kHz) {a.workload}, batch={a.batch}x{a.samples}", "value": r["value"],
        "unit": "samples/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": r["ms_per_step"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": r["workload"], "segments_total": a.batch, "samples_per_segment": a.samples},
        "device_ms_per_step": r["device_ms_per_step"],
        "bytes_per_sample": r["bytes_per_sample"], "roofline": r["roofline"], "checks": r["checks"]}, a.detail_file)


def _host_threads():
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    return max(1, min(cores, 32))


def other_workloads(a, dev, check):
    """The BASELINE configs no other driver-run line covers, attached to the default line as `other_workloads` (never part
    of `value`): configs[2] (DiffDelGRU) and configs[3] (TCN) at the headline's batch, and the GRU workload at the per-GPU
    shapes of configs[4]'s strong-scaling legs (32 768 segments over 4 / 2 / 1 GPUs = 8192 / 16 384 / 32 768 per GPU)
    -- PER-GPU LEGS measured on one GPU, not a scaling curve.  Each entry: whole-pass rate, the dominant kernel's
    event-timed launch duration and roofline fraction, and scattered streams against the oracle."""
    out = {"note": "single-GPU measurements beside the headline; gru_B* / tcn_B* are the per-GPU shapes of configs[4]'s "
                   "strong-scaling legs (per-GPU legs, not a scaling curve)"}
    threads = _host_threads()
    jobs = ([(wl, wl, a.batch) for wl in ("diffdel", "tcn")] + [(f"gru_B{b}", "gru", b) for b in a.other_gru_batches]
            + [(f"tcn_B{b}", "tcn", b) for b in a.other_tcn_batches] + [(f"diffdel_B{b}", "diffdel", b) for b in a.other_diffdel_batches])
    live_keys = ("diffdel", "tcn") + tuple(f"diffdel_B{b}" for b in a.other_diffdel_batches)       # PMC child passes in this run
    profiled = any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
    live = a.traffic == "live" or (a.traffic == "auto" and not profiled)          # never a profiler inside a profiler
    for key, wl, b in jobs:
        try:
            out[key] = measure_workload(wl, b, a.samples, a.other_steps, 1, check, dev, threads, live=live and key in live_keys)
        except Exception as e:          # a side workload must never take the headline line down with it (e.g. out of memory
            out[key] = {"error": f"{type(e).__name__}: {e}"[:500]}      # on a box that is shared or smaller than expected)
            torch.cuda.empty_cache()
    if a.cli_segments > 0:
        try:
            out["cli"] = cli_workload(dev, check, n_seg=a.cli_segments)
        except Exception as e:
            out["cli"] = {"error": f"{type(e).__name__}: {e}"[:500]}
        torch.cuda.empty_cache()
    return out


def live_traffic(kernel_regex, extra_args=(), per_forward=None):
    """HBM bytes per launch of the headline kernel, measured in this run: two child processes
    `rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python3 bench.py --steps 5 ...` (separate passes, kernel trace
    only, the interpreter directly behind `--`: the combination MI355X_MICROARCH.md prescribes), the median over the
    full-size launches, FETCH_SIZE doubled (gfx950 reports half of a wide coalesced streaming read), KB of 1024 bytes.
    `per_forward = (regex of a kernel launched a known number of times per forward, that number)`: a workload whose forward is
    SEVERAL launches (the TCN: first block + matrix-pipe blocks per stream chunk) -- every matching launch summed, over the
    number of forwards seen.
    -> (bytes, source note) or (None, reason)."""
    import csv
    import glob
    import re
    import shutil
    import tempfile
    prof = shutil.which("rocprofv3")
    if prof is None:
        return None, "rocprofv3 not on the PATH"
    rx = re.compile(kernel_regex)
    med = {}
    try:
        with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
            for c in ("FETCH_SIZE", "WRITE_SIZE"):
                d = os.path.join(tmp, c)
                cmd = [prof, "--kernel-trace", "--pmc", c, "--output-format", "csv", "-d", d, "-o", "pmc", "--", sys.executable,
                       os.path.abspath(__file__), "--steps", "5", "--warmup", "1", "--no-cpu-baseline", "--no-extra", "--other", "off",
                       "--traffic", "off", "--detail-file", os.path.join(tmp, c + "_detail.json")] + list(extra_args)
                env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK",
                                                                      "ROLE_RANK", "MASTER_ADDR", "MASTER_PORT", "NTM_DIST_FORCE_INIT")
                       and not k.startswith("TORCHELASTIC")}
                env["TMPDIR"] = "/tmp"
                # the profiler's python grandchild holds the GPU: on a timeout the whole process GROUP must go before the
                # parent measures anything else (killing rocprofv3 alone would leave it launching 4096 x 65536 kernels)
                proc = subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                                        start_new_session=True)
                try:
                    proc.communicate(timeout=240)
                except subprocess.TimeoutExpired:
                    import signal
                    try:
                        os.killpg(proc.pid, signal.SIGKILL)       # the group this call created (pgid == pid), nothing else
                    except ProcessLookupError:
                        pass
                    proc.communicate()
                    return None, f"{c} pass timed out (process group killed)"
                r = proc
                vals, n_once = [], 0
                rx_once = re.compile(per_forward[0]) if per_forward else None
                for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                    for row in csv.DictReader(open(f)):
                        if row["Counter_Name"] == c and rx.search(row["Kernel_Name"]):
                            vals.append(float(row["Counter_Value"]))
                            n_once += 1 if rx_once is not None and rx_once.search(row["Kernel_Name"]) else 0
                if per_forward:
                    if r.returncode != 0 or not n_once or n_once % per_forward[1]:
                        return None, f"{c} pass failed (exit {r.returncode}, {len(vals)} launches seen, {n_once} of the once-per-chunk kernel)"
                    med[c] = sum(vals) / (n_once // per_forward[1])
                    continue
                big = sorted(v for v in vals if v > 0.5 * max(vals)) if vals else []
                if r.returncode != 0 or not big:
                    return None, f"{c} pass failed (exit {r.returncode}, {len(vals)} launches seen)"
                med[c] = big[len(big) // 2]
    except Exception as e:                                   # a profiler hiccup must never cost the bench line
        return None, f"{type(e).__name__}: {e}"[:200]
    how = "every launch of the forward summed, per forward" if per_forward else "median of the full-size launches"
    return (2.0 * med["FETCH_SIZE"] + med["WRITE_SIZE"]) * 1024.0, (
        f"measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE child passes of this command at 5 steps "
        f"({how}: FETCH {med['FETCH_SIZE']:.0f} KB x 2 (gfx950 correction) + WRITE {med['WRITE_SIZE']:.0f} KB)")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _visible_gpus():
    """Number of GPUs this job may use WITHOUT touching the HIP runtime in the launcher process: the visibility
    variables first, else the KFD topology (GPU nodes carry a non-zero simd_count).  None = could not tell."""
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([t for t in v.split(",") if t.strip() != ""])
    top = "/sys/class/kfd/kfd/topology/nodes"
    try:
        n = 0
        for node in os.listdir(top):
            with open(os.path.join(top, node, "properties")) as f:
                for line in f:
                    if line.startswith("simd_count") and int(line.split()[1]) > 0:
                        n += 1
        return n
    except (OSError, ValueError):
        return None


def spawn_ranks(n, argv, grace=5.0):
    """`python bench.py --gpus N` without a launcher: start N fresh interpreters of this file, one per GPU, with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, relay rank 0's stdout (the JSON line), and
    return non-zero if any rank fails.  After the FIRST failure the other ranks get `grace` seconds to end by
    themselves (a rank whose peer vanished usually dies of a signal inside the collective library), are then
    terminated by PID, and EVERY status is reported; the rank blamed first is one that exited by itself with a
    positive status (the culprit), in preference to ranks killed by a signal (its victims).
    This parent never touches the HIP runtime or torch.cuda (the GPU count comes from the visibility variables or
    the KFD topology), and no process that has is ever re-exec'd: the children are new processes."""
    backend = os.environ.get("NTM_DIST_BACKEND", "nccl")
    have = _visible_gpus()
    if backend == "nccl" and have is not None and have < n and "--launch-check" not in argv:
        print(f"bench.py: --gpus {n} needs {n} visible GPUs for the RCCL run, found {have} "
              f"(NTM_DIST_BACKEND=gloo shares one GPU between ranks for a control-flow check)", file=sys.stderr)
        return 2
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL needs it on this driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
    out0 = []
    import threading
    drain = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)   # rank 0's pipe never fills
    drain.start()
    status, order = {}, []                                       # rank -> exit status, in the order the exits were seen
    deadline = None
    try:
        while len(status) < n:
            for r, p in enumerate(procs):
                if r not in status and p.poll() is not None:
                    status[r] = p.returncode
                    order.append(r)
                    if p.returncode != 0 and deadline is None:
                        deadline = time.monotonic() + grace
            if deadline is not None and time.monotonic() > deadline:
                break
            time.sleep(0.02)
    finally:
        terminated = [r for r, p in enumerate(procs) if p.poll() is None]
        for r in terminated:
            procs[r].terminate()
        for r, p in enumerate(procs):
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        drain.join(timeout=10)
    sys.stdout.write("".join(o or "" for o in out0))
    sys.stdout.flush()
    failed = [r for r in order if status[r] != 0]
    if failed:
        culprits = [r for r in failed if status[r] > 0] or failed
        first = culprits[0]
        others = ", ".join(f"rank {r}: {status[r]}" for r in failed if r != first)
        print(f"bench.py: rank {first} exited with status {status[first]}"
              + (f" (then {others})" if others else "")
              + (f"; terminated by the launcher: ranks {terminated}" if terminated else ""), file=sys.stderr)
        return 1
    return 0


def launch_check(a):
    """--launch-check: the N-rank control flow of this file without the data path (runs on CPU over gloo in the
    `not gpu` tests): rank environment -> process group -> shard ranges -> one SUM all-reduce -> rank 0 prints."""
    from ntm_amd import distributed as D
    rank, world, local = D.init_from_env("gloo")
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    total = a.total_batch if a.scaling == "strong" else a.batch * world
    lo, hi = D.shard_range(total, rank, world) if a.scaling == "strong" else (rank * a.batch, (rank + 1) * a.batch)
    v = torch.tensor([float(hi - lo), float(rank), 1.0], dtype=torch.float64)
    if a.fail_rank == rank and a.fail_early:
        os._exit(3)                 # a rank that dies BEFORE the collective: its peers are left inside the all-reduce
    if world > 1:
        torch.distributed.all_reduce(v)
    D.barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()       # before any rank leaves: no peer dies inside gloo's teardown
    if rank == 0:
        print(json.dumps({"launch_check": True, "world": world, "segments_total": int(v[0]), "rank_sum": int(v[1]),
                          "ranks": int(v[2]), "scaling": a.scaling}))
    if a.fail_rank == rank:
        sys.exit(3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4096, help="segments per GPU (weak scaling)")
    ap.add_argument("--samples", type=int, default=65536, help="samples per segment")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --batch segments per GPU (N = 8: BASELINE configs[4]); strong: --total-batch segments over all GPUs")
    ap.add_argument("--total-batch", type=int, default=32768, help="segments of the whole job under --scaling strong")
    ap.add_argument("--variant", default="auto", choices=["auto", "mfma2", "mfma", "valu", "f16x3", "bf16x3"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", default="full", choices=["full", "small"],
                    help="cpu_baseline: full = 16 x 8192 best of 3 + 64 x 8192 (about 10 s); small = 16 x 8192 once (tests)")
    ap.add_argument("--no-extra", action="store_true", help="skip the opt-in f16x3 kernel leg")
    ap.add_argument("--other", default="auto", choices=["auto", "on", "off"],
                    help="attach `other_workloads` (configs[2], [3] and the per-GPU shapes of configs[4]) to the line: "
                         "auto = only for the default single-GPU workload (4096 x 65536)")
    ap.add_argument("--esr", default="fused", choices=["fused", "pass"],
                    help="the loss leg of a step: fused = the ESR sums are accumulated inside the recurrent launch "
                         "(RNN.forward_esr / ntm_gru_forward_esr); pass = forward launch, then the streaming ESR pass on a side "
                         "stream under the next step's launch (rounds 1-2)")
    ap.add_argument("--traffic", default="auto", choices=["auto", "live", "file", "off"],
                    help="roofline.traffic of the headline kernel: live = two rocprofv3 --pmc child passes in this run (auto: when "
                         "rocprofv3 is available), file = the newest profiles/*pmc_traffic_mfma2*.json, off = null")
    ap.add_argument("--delay-mode", default="auto", choices=["auto", "two_pass", "fused"],
                    help="--workload diffdel: the fused DiffDelRNN step (auto: where the matrix-pipe kernel runs) or GRU launch + delay pass")
    ap.add_argument("--other-steps", type=int, default=3)
    ap.add_argument("--other-gru-batches", type=lambda v: [int(t) for t in v.split(",") if t], default=[8192, 16384, 32768])
    ap.add_argument("--other-tcn-batches", type=lambda v: [int(t) for t in v.split(",") if t], default=[8192, 16384],
                    help="TCN legs at the per-GPU shapes of configs[4] (they did not fit before the forward was chunked by streams)")
    ap.add_argument("--other-diffdel-batches", type=lambda v: [int(t) for t in v.split(",") if t], default=[8192],
                    help="DiffDelGRU legs beside configs[2]'s 4096 (8192: the YPN = 4 form of the fused launch, PMC traffic measured live)")
    ap.add_argument("--cli-segments", type=int, default=128,
                    help="other_workloads.cli: the evaluation command end to end on this many 10-second segments (0 = skip)")
    ap.add_argument("--workload", default="gru", choices=["gru", "diffdel", "tcn"],
                    help="gru = BASELINE configs[1] (the headline metric); diffdel = configs[2]; tcn = configs[3]")
    ap.add_argument("--detail-file", default=os.path.join(ROOT, "bench_detail.json"),
                    help="where rank 0 writes the full record; stdout's last line is the compact record (< 4 KB) that names it")
    ap.add_argument("--launch-check", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--fail-rank", type=int, default=-1, help=argparse.SUPPRESS)
    ap.add_argument("--fail-early", action="store_true", help=argparse.SUPPRESS)
    a = ap.parse_args()

    # N > 1 and nobody gave us a rank: this process becomes the launcher (before anything touches the GPU)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a.gpus, sys.argv[1:]))
    if a.launch_check:
        return launch_check(a)

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
    T = a.samples
    if a.scaling == "strong":
        lo, hi = D.shard_range(a.total_batch, rank, world)
        B, first_segment = hi - lo, lo
    else:
        B, first_segment = a.batch, rank * a.batch
    total_segments = a.total_batch if a.scaling == "strong" else a.batch * world
    # which device each rank drives, gathered over the job's own backend (shows that RCCL saw N ranks on N GPUs)
    props = torch.cuda.get_device_properties(local)
    mine = torch.tensor([rank, local, int(getattr(props, "pci_bus_id", -1)), int(getattr(props, "pci_device_id", -1))],
                        dtype=torch.int64, device=dev)
    grouped = torch.distributed.is_initialized()      # world > 1, or one rank with NTM_DIST_FORCE_INIT=1
    if grouped:
        allr = [torch.empty_like(mine) for _ in range(world)]
        torch.distributed.all_gather(allr, mine)
    else:
        allr = [mine]
    rank_devices = [{"rank": int(v[0]), "device": int(v[1]), "pci_bus": int(v[2]), "pci_device": int(v[3])} for v in allr]

    model = ntm_amd.harness.build_model(weights.W_GRU, device=dev)
    model.kernel_variant = a.variant
    x = synth_input(B, T, dev, seed=1234 + first_segment // 4096)    # weak: one seed per 4096-segment block
    gold = None
    if rank == 0 and T == 65536:
        gold = np.load(os.path.join(ROOT, "tests", "golden", "g6_long_65536.npz"))
        x[0, 0] = torch.from_numpy(gold["x"][0, 0]).to(dev)
    INIT_LEN = 1024

    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    side = torch.cuda.Stream(device=dev)       # the loss leg (ESR sums + the job-wide reduction) runs here
    fused_esr = a.esr == "fused" and a.variant == "auto"     # an explicitly chosen kernel variant takes the separate ESR pass

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
        if target is not None and fused_esr:
            # forward + ESR sums in ONE call (ntm_gru_forward_esr: the sums ride in the recurrent launch's y-tile flush)
            y, s = model.forward_esr(x, target, skip=INIT_LEN)
        else:
            y, s = model.forward(x), None
        evs[1].record()
        pend = None
        if target is not None:
            side.wait_event(evs[1])
            y.record_stream(side)
            with torch.cuda.stream(side):       # the per-segment scalars (and, with --esr pass, the 2 GB ESR pass itself)
                if s is None:
                    s = esr_sums(y, target, skip=INIT_LEN)
                else:
                    s.record_stream(side)
                n = T - INIT_LEN
                pend = D.local_loss_sums((s[:, 0] / n) / (s[:, 1] / n + ESR_EPS), s)
                last["sums"] = s
        return y, pend

    def make_target(y0):
        """A target that is NOT the output (round 3 used y0 itself: every timed step then summed exact zeros and a bug in the
        fused accumulation could not have moved the line): 0.9 y0 + 0.02 x, built once, resident."""
        t = y0 * 0.9
        t.add_(x, alpha=0.02)
        return t

    target, y_first = None, None
    last = {}                               # per-stream sums of the most recent loss leg (a reference, no extra GPU work)
    for _ in range(max(a.warmup, 0)):
        y, pend = one_pass(target)
        if pend is not None:
            torch.cuda.current_stream().wait_stream(side)
            D.reduce_many([pend])
        if target is None:
            y_first = y.clone()
            target = make_target(y_first)
    if target is None:                      # --warmup 0: still need the first-pass output and the target
        y_first = one_pass(None)[0].clone()
        target = make_target(y_first)
    # allocator pre-warm (untimed, whatever --warmup is): a timed step allocates its output and its sums while the previous
    # step's are still alive; two passes with the loss leg put every such block into torch's caching allocator, so that no
    # timed step carries a hipMalloc (with --warmup 1 the first timed step did)
    for _ in range(2):
        y, pend = one_pass(target)
        torch.cuda.current_stream().wait_stream(side)
        D.reduce_many([pend])
    if a.fail_rank == rank:                 # test hook: a rank that dies after warm-up, its peers left in the barrier
        torch.cuda.synchronize()
        os._exit(3)

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
    # every timed step must produce the SAME loss scalars: compared per rank, bit for bit, on the local vectors (the job-wide
    # values go through an all-reduce whose ring / tree adds different elements of the buffer in different rank orders, so
    # two rows that are equal on every rank may differ in the last bit after it), then combined over the ranks
    steps_identical = D.max_over_ranks(0.0 if all(torch.equal(p, pends[0]) for p in pends) else 1.0, dev) == 0.0
    kern_ms = [e0.elapsed_time(e1) for e0, e1 in step_evs]
    # the per-GPU roofline is the SLOWEST rank's mean launch duration; the other ranks' means ride along for the record
    kern_ms_rank = float(np.mean(kern_ms))
    kmine = torch.tensor([kern_ms_rank, float(B)], dtype=torch.float64, device=dev)
    if grouped:
        kall = [torch.empty_like(kmine) for _ in range(world)]
        torch.distributed.all_gather(kall, kmine)
    else:
        kall = [kmine]
    kern_by_rank = [{"rank": r, "segments": int(v[1]), "kernel_ms": float(v[0])} for r, v in enumerate(kall)]
    sums_last = last["sums"].clone() if "sums" in last else None
    deterministic = bool(torch.equal(y, y_first))

    # ---- the same step WITHOUT the warm-start cache (what predict() does on every call in the reference, code/model.py:230:
    #      1024 zero samples, B = 1, recomputed by a launch of its own): 3 steps, same brackets, reported beside ms_per_step
    model.warm_cache = False
    one_pass(target)
    D.barrier()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    pends_nc = [one_pass(target)[1] for _ in range(3)]
    torch.cuda.current_stream().wait_stream(side)
    res_nc = D.reduce_many(pends_nc)
    torch.cuda.synchronize()
    D.barrier()
    ms_no_cache = 1e3 * D.max_over_ranks(time.perf_counter() - t1, dev) / 3
    model.warm_cache = True
    no_cache_same = (D.max_over_ranks(0.0 if all(torch.equal(p, pends[0]) for p in pends_nc) else 1.0, dev) == 0.0) if pends else None
    del res_nc

    # ---- opt-in kernel variant, reported beside the headline (never part of `value`): the f16x3 GEMV
    #      engine, checked over the whole batch against the exact-fp32 pass
    extra = {}
    if a.variant in ("auto", "mfma2") and not a.no_extra and world == 1:
        model.kernel_variant = "f16x3"
        ms2 = []
        for i in range(2 + 5):
            y2, _ = one_pass(None)
            torch.cuda.synchronize()
            if i >= 2:
                ms2.append(ev0.elapsed_time(ev1))
        diff = (y2 - y_first).abs().max().item()
        s2 = esr_sums(y2, y_first, skip=INIT_LEN).sum(dim=0)
        k2 = float(np.mean(ms2)) / 1e3
        extra["f16x3"] = {
            "what": "same kernel with W.h evaluated as three fp16 hi/lo MFMA products, fp32 accumulate (opt-in)",
            "kernel_ms": 1e3 * k2, "samples_per_s_kernel": B * T / k2,
            "max_abs_diff_vs_exact_fp32_whole_batch": diff,
            "esr_vs_exact_fp32_whole_batch": float(s2[0] / (s2[1] + ESR_EPS * B * (T - INIT_LEN)))}
        # yardstick: two EXACT fp32 kernels that only differ in summation order (MFMA K order)
        model.kernel_variant = "mfma"
        y3, _ = one_pass(None)
        extra["f16x3"]["yardstick_max_abs_diff_between_two_exact_fp32_kernels"] = (y3 - y_first).abs().max().item()
        del y3
        if gold is not None:
            extra["f16x3"]["stream0_vs_reference_max_abs"] = float(
                np.abs(gold["y"][0, 0] - y2[0, 0].cpu().numpy()).max())
        # the bf16x3 engine (round 6, opt-in): W and h as three bf16 pieces each -- the fp32 operands exactly -- and W.h as eight
        # of their nine partial products on the bf16 matrix pipe, fp32 accumulate; executed-MFMA flops against the bf16 peak AND
        # the algorithmic 25 088 flop/sample against the fp32 peak, whole batch against the exact-fp32 pass
        model.kernel_variant = "bf16x3"
        ms3 = []
        for i in range(2 + 5):
            y4, _ = one_pass(None)
            torch.cuda.synchronize()
            if i >= 2:
                ms3.append(ev0.elapsed_time(ev1))
        k3 = float(np.mean(ms3)) / 1e3
        s4 = esr_sums(y4, y_first, skip=INIT_LEN).sum(dim=0)
        mfma_flop = 8 * 2 * 12288                        # eight products of the 192 x 64 GEMV, 2 flop per multiply-add
        extra["bf16x3"] = {
            "what": "same kernel with W and h as three bf16 pieces each (24 bits: the fp32 operands exactly), W.h = 8 of the 9 partial "
                    "products (W_3.h_3 <= 2^-32 |W||h| dropped) on v_mfma_f32_16x16x32_bf16, fp32 accumulate; hand-scheduled step (opt-in)",
            "kernel_ms": 1e3 * k3, "samples_per_s_kernel": B * T / k3, "speedup_vs_exact_fp32_kernel": kern_ms_rank / (1e3 * k3),
            "cycles_per_step_at_2p4_GHz": k3 / T * 2.4e9,
            "roofline_executed": {"bound": "mfma", "achieved": mfma_flop * B * T / k3 / 1e12, "peak": 2500.0, "unit": "TFLOP/s",
                                  "frac": mfma_flop * B * T / k3 / 1e12 / 2500.0, "flop_per_sample": mfma_flop,
                                  "what": "executed bf16 MFMA flops against the dense bf16 peak"},
            "roofline_algorithmic": {"bound": "mfma", "achieved": FLOP_PER_SAMPLE * B * T / k3 / 1e12, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                                     "frac": FLOP_PER_SAMPLE * B * T / k3 / 1e12 / PEAK_FP32_TFLOPS, "flop_per_sample": FLOP_PER_SAMPLE,
                                     "what": "the algorithmic flops of the fp32 recurrence against the fp32 peak (may exceed 1: "
                                             "the engine does not run on the fp32 pipe)"},
            "max_abs_diff_vs_exact_fp32_whole_batch": (y4 - y_first).abs().max().item(),
            "esr_vs_exact_fp32_whole_batch": float(s4[0] / (s4[1] + ESR_EPS * B * (T - INIT_LEN))),
            "parity_table": "profiles/r06_*_full_batch_parity.jsonl, profiles/r06_*_checkpoint_parity.jsonl (|hip - f64| beside the exact engine's)"}
        if gold is not None:
            extra["bf16x3"]["stream0_vs_reference_max_abs"] = float(np.abs(gold["y"][0, 0] - y4[0, 0].cpu().numpy()).max())
        del y4
        if (B, T) == (4096, 65536):
            # from 8192 streams up two stream groups share a CU (250 registers: two waves per SIMD) and one group's LDS exchange runs
            # under the other's MFMAs: the engine's rate at the per-GPU shape of configs[4]'s 4-GPU strong-scaling leg, exact engine beside it
            x8 = synth_input(8192, T, dev, seed=99)
            at8 = {}
            for vname in ("bf16x3", "mfma2"):
                model.kernel_variant = vname
                ms8 = []
                for i in range(1 + 3):
                    model.initialize_hidden()
                    model.warm_start()
                    model.hidden = model.hidden.expand(1, 8192, 64).contiguous()
                    ev0.record(); y8 = model.forward(x8); ev1.record(); torch.cuda.synchronize()
                    if i:
                        ms8.append(ev0.elapsed_time(ev1))
                    del y8
                at8[vname] = float(np.mean(ms8))
            extra["bf16x3"]["at_8192_streams"] = {"kernel_ms": at8["bf16x3"], "samples_per_s_kernel": 8192.0 * T / (at8["bf16x3"] * 1e-3),
                                                  "exact_fp32_kernel_ms": at8["mfma2"], "speedup_vs_exact_fp32_kernel": at8["mfma2"] / at8["bf16x3"]}
            del x8
            torch.cuda.empty_cache()
        model.kernel_variant = a.variant
        # the loss dict that follows the path in code/test-model.py:250-254 (never part of `value` beyond the ESR
        # sums the timed step already contains): ESR, DCPreESR and MultiSTFT over the whole batch, f16x3 output
        # against the exact-fp32 output
        lm = {}
        for name, fn in (("ESR", lambda: esr_sums(y2, y_first, skip=INIT_LEN)),
                         ("DCPreESR", lambda: esr_dcpre_sums(y2, y_first, skip=INIT_LEN)),
                         ("MultiSTFT", lambda: MRSTFTLoss().per_segment(y2, y_first, skip=INIT_LEN))):
            for _ in range(2):
                ev0.record(); r = fn(); ev1.record(); torch.cuda.synchronize()
            lm[name + "_ms"] = ev0.elapsed_time(ev1)
        lm["MultiSTFT_f16x3_vs_exact_mean"] = float(r.mean())
        extra["loss_pass"] = lm
        del y2
        # the two time-domain losses out of ONE launch (RNN.forward_losses / ntm_gru_forward_losses: ESR + DCPreESR sums in the
        # recurrent launch's flush) against the timed step's launch + the streaming DCPreESR pass behind it; never part of `value`
        def timed_loss_leg(fn, n=4):
            ms_ = []
            for i in range(n + 1):
                model.initialize_hidden()
                model.warm_start()
                model.hidden = model.hidden.expand(1, B, 64).contiguous()
                ev0.record(); r_ = fn(); ev1.record(); torch.cuda.synchronize()
                if i:
                    ms_.append(ev0.elapsed_time(ev1))
            return float(np.mean(ms_)), r_
        ms_l, (_, s_l, d_l) = timed_loss_leg(lambda: model.forward_losses(x, target, INIT_LEN))
        ms_p, (s_p, d_p) = timed_loss_leg(lambda: (lambda ys: (ys[1], esr_dcpre_sums(ys[0], target, INIT_LEN)))(model.forward_esr(x, target, INIT_LEN)))
        extra["esr_dcpre_fused"] = {
            "what": "forward + ESR + DCPreESR sums in one launch (gru_mfma2_kernel<ESR, DCP>) vs the ESR launch + the streaming DCPreESR pass",
            "one_launch_ms": ms_l, "esr_launch_plus_dcpre_pass_ms": ms_p, "saved_ms": ms_p - ms_l,
            "esr_sums_identical": bool(torch.equal(s_l, s_p)),
            "dcpre_sums_max_rel_diff": float(((d_l - d_p).abs() / d_p.abs().clamp_min(1e-300)).max())}
    # achievable HBM rate on this box (state it beside the 8 TB/s vendor peak): device copy of the
    # input batch, read + write bytes over the event time
    cp = torch.empty_like(x)
    for _ in range(3):
        ev0.record(); cp.copy_(x); ev1.record(); torch.cuda.synchronize()
    hbm_copy_gbs = 2.0 * x.numel() * 4 / (ev0.elapsed_time(ev1) * 1e-3) / 1e9
    del cp
    if grouped:
        D.barrier()
        torch.distributed.destroy_process_group()
    if rank != 0:
        return
    total_samples = float(total_segments) * T * a.steps
    # N > 1: the figure of the rank that bounds the job (largest launch duration per segment it owns), rank 0's when N = 1
    slow = max(kern_by_rank, key=lambda d: d["kernel_ms"])
    kern_s = kern_ms_rank / 1e3
    # exact-fp32 kernels: algorithmic flops against the fp32 matrix peak.  --variant f16x3 (opt-in): the MFMA
    # flops it actually executes (three fp16 products per W.h term) against the dense fp16 MFMA peak.
    flop_per_sample, peak_tflops, dtype = FLOP_PER_SAMPLE, PEAK_FP32_TFLOPS, "f32"
    if a.variant == "f16x3":
        flop_per_sample, peak_tflops, dtype = 3 * 2 * 12288 + 2 * (192 + 64), 2500.0, "f16x3 products, f32 accumulate"
    if a.variant == "bf16x3":
        flop_per_sample, peak_tflops, dtype = 8 * 2 * 12288 + 2 * (192 + 64), 2500.0, "bf16x3 x bf16x3 products (operand-exact), f32 accumulate"
    tflops = flop_per_sample * B * T / kern_s / 1e12
    # algorithmic bytes of the dominant launch: x in + y out, + the target it reads when the loss leg rides in it
    bytes_per_sample = BYTES_PER_SAMPLE + (4 if fused_esr else 0)
    hbm_gbs = bytes_per_sample * B * T / kern_s / 1e9
    checks = {"target": "0.9 * first-pass output + 0.02 * x (not the output: the sums are non-zero)",
              "job_esr": res["mean_segment_loss"] if res else None, "job_sum_err2": res["sum_err2"] if res else None,
              "job_sum_tgt2": res["sum_tgt2"] if res else None, "segments": res["segments"] if res else None,
              "every_timed_step_same_loss": steps_identical, "no_warm_cache_steps_same_loss": no_cache_same,
              "last_output_equals_first_pass_bitwise": deterministic}
    if gold is not None:
        yg = y[0, 0].cpu().numpy()
        e = gold["y"][0, 0] - yg
        checks["stream0_vs_reference_max_abs"] = float(np.abs(e).max())
        checks["stream0_vs_reference_esr"] = float((e[INIT_LEN:] ** 2).mean() /
                                                   ((gold["y"][0, 0][INIT_LEN:] ** 2).mean() + ESR_EPS))
    # HBM bytes per launch from the PMC counters, collected as MI355X_MICROARCH.md prescribes (separate rocprofv3 --pmc
    # passes, gfx950 FETCH_SIZE correction), for the launch THIS rank timed (B streams x T samples).  `--traffic live` (what
    # `auto` does when rocprofv3 is on the PATH): the two passes run NOW, as single-GPU child processes of this line's own
    # command at this rank's batch and 5 steps (N > 1: after the process group is gone, on rank 0's GPU -- the launch of a
    # rank does not depend on N), so the number belongs to this box and this binary; otherwise, or if a pass fails, the
    # newest profiles/*pmc_traffic* file of the same kernel and shape.
    traffic, traffic_source = None, None
    many = (B + 15) // 16 > torch.cuda.get_device_properties(local).multi_processor_count      # launch_gru_mfma2: YPN 4 / 16
    if T == 65536 and B > 1024 and a.variant in ("auto", "mfma2") and a.traffic != "off":
        profiled = any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
        rx = r"gru_mfma2_kernel<true, false, 0, 0, " + ("4" if many else "16") + ", false, " + ("true, false>" if fused_esr else "false, false>")
        if a.traffic == "live" or (a.traffic == "auto" and not profiled):     # never a profiler inside a profiler
            traffic, traffic_source = live_traffic(rx, ["--esr", a.esr, "--variant", a.variant, "--batch", str(B),
                                                        "--samples", str(a.samples)])
        if traffic is None:
            import glob
            tag = "" if B == 4096 else f"_B{B}"
            pats = [f"*pmc_traffic_mfma2_esr{tag}.json"] if fused_esr else [f"*pmc_traffic_mfma2{tag}.json", f"*pmc_traffic_gru_B{B}.json"]
            files = sorted(f for pat in pats for f in glob.glob(os.path.join(ROOT, "profiles", pat)))
            if files:
                why = f"; live passes: {traffic_source}" if traffic_source else ""
                traffic = json.load(open(files[-1]))["hbm_bytes_per_launch_corrected"]
                traffic_source = ("profiles/" + os.path.basename(files[-1]) + " (rocprofv3 --pmc passes of this rank's launch as a "
                                  "single-GPU command; not re-measured in this run" + why + ")")
    # roofline of the dominant launch: the per-GPU figure (this rank's launch: B streams) and, for N > 1, the job's --
    # all ranks' algorithmic flops over the SLOWEST rank's mean launch duration, against N x the peak
    roofline = {"bound": "mfma", "achieved": tflops, "peak": peak_tflops, "unit": "TFLOP/s",
                "frac": tflops / peak_tflops, "traffic": traffic, "traffic_source": traffic_source,
                "algorithmic_bytes": float(bytes_per_sample) * B * T,
                "kernel": {"auto": "gru_mfma2_kernel", "mfma2": "gru_mfma2_kernel", "mfma": "gru_mfma_kernel",
                           "valu": "gru_valu_kernel", "f16x3": "gru_mfma2_kernel<f16x3>",
                           "bf16x3": "gru_mfma2_kernel<bf16x3>"}[a.variant]
                          + ("<ESR: forward + loss sums>" if fused_esr else ""),
                "kernel_ms": 1e3 * kern_s, "flop_per_sample": flop_per_sample, "per": "GPU (rank 0's launch)",
                "segments_in_launch": B,
                "hbm": {"achieved": hbm_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": hbm_gbs / PEAK_HBM_GBS, "bytes_per_sample": bytes_per_sample,
                        "copy_kernel_measured": hbm_copy_gbs}}
    agg_tflops = flop_per_sample * float(total_segments) * T / (slow["kernel_ms"] * 1e-3) / 1e12
    roofline["aggregate"] = {"n_gpus": world, "achieved": agg_tflops, "peak": world * peak_tflops, "unit": "TFLOP/s",
                             "frac": agg_tflops / (world * peak_tflops),
                             "what": "all ranks' algorithmic flops / the slowest rank's mean launch duration, against N x the per-GPU peak",
                             "slowest_rank": slow["rank"], "kernel_ms_by_rank": kern_by_rank,
                             "hbm": {"achieved": bytes_per_sample * float(total_segments) * T / (slow["kernel_ms"] * 1e-3) / 1e9,
                                     "peak": world * PEAK_HBM_GBS, "unit": "GB/s",
                                     "frac": bytes_per_sample * float(total_segments) * T / (slow["kernel_ms"] * 1e-3) / 1e9 / (world * PEAK_HBM_GBS)}}
    per_gpu = f"{B}/GPU" if a.scaling == "weak" else f"{a.total_batch} total, {B} on rank 0"
    out = {
        # BASELINE.json's metric with the point of its "1/2/4/8 GPU" axis this line is
        "metric": f"audio samples/sec (44.1 kHz) GRU-HS[64], batch={a.batch if a.scaling == 'weak' else B}x{T}"
                  + (f", {world} GPU" if world == 1 and a.scaling == "weak" else f" per GPU, {world} GPU ({a.scaling} scaling, {total_segments} segments)"),
        "value": total_samples / elapsed, "unit": "samples/s", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps, "ms_per_step_no_warm_cache": ms_no_cache,
        "value_no_warm_cache": float(total_segments) * T / (ms_no_cache * 1e-3), "higher_is_better": True,
        "scaling": a.scaling, "vs_baseline": None, "dtype": dtype, "data": "synthetic",
        # (the driver keeps 120 characters of this string)
        "config": {"workload": f"GRU-HS[64] {total_segments}x{T} f32 on {world} GPU ({per_gpu}), predict+ESR "
                               + ("fused" if fused_esr else "as a side-stream pass") + ", warm-cache on, 1 all-reduce/job",
                   "workload_detail": "CHOWTAPE weights; a step = RNN.predict over the rank's resident batch (warm-start state of 1024 zero samples kept "
                                      "per parameter version after the first pass, then ONE persistent GRU launch) with the per-stream ESR sums "
                                      + ("accumulated inside the recurrent launch" if fused_esr else "as a streaming pass on a side stream under the next step's launch")
                                      + "; the K steps' loss scalars go through one all-reduce before the closing synchronisation; "
                                        "value_no_warm_cache = the same step with the warm-start launch recomputed every pass (code/model.py:229-230)",
                   "segments_total": total_segments, "segments_rank0": B, "samples_per_segment": T, "kernel": a.variant,
                   "untimed_passes_before_the_timed_region": max(a.warmup, 0) + (0 if a.warmup > 0 else 1) + 2,
                   "parallelism": f"streams sharded over {world} GPU(s), no data-path collective"},
        "backend": ("rccl (torch.distributed nccl)" if backend == "nccl" else backend) if grouped else "none (single process)",
        "rccl_ranks": world if (backend == "nccl" and grouped) else 0, "ranks": world, "rank_devices": rank_devices,
        "realtime_factor": total_samples / elapsed / FS,
        "roofline": roofline,
        "checks": checks,
    }
    info = os.path.join(ROOT, "neural-tape-modeling_amd", "build_info.json")      # written by __graft_entry__.build()
    bi = None
    # (read only: the record is written next to the library by its Makefile, so that this process -- which has initialised the
    # GPU, possibly under a profiler -- never starts compiler tools)
    try:
        import hashlib
        bi = json.load(open(info))
        with open(ntm_amd._lib.LIB_PATH, "rb") as f:
            if hashlib.sha256(f.read()).hexdigest() != bi.get("library_sha256"):
                bi = None
                out["build"] = {"error": "build_info.json describes another libntm.so: run `make -C neural-tape-modeling_amd/csrc`"}
    except (OSError, ValueError) as e:
        bi = None
        out["build"] = {"error": f"build_info.json: {type(e).__name__}: {e}"[:200]}
    if bi is not None:
        want = roofline["kernel"].split("<")[0]
        out["build"] = {"compiler": bi.get("compiler"), "arch": bi.get("arch"), "git_head_at_build": bi.get("git_head_at_build"),
                        "library_sha256": bi.get("library_sha256"),
                        "runtime_hip": torch.version.hip,
                        "kernels": [{k: v for k, v in kk.items() if k != "symbol"} for kk in bi.get("kernels", [])
                                    if want in kk["kernel"] or "gru_lat_kernel" in kk["kernel"]]}
    if extra:
        out["other_kernels"] = extra
    if not a.no_cpu_baseline:
        # N > 1: rank 0 alone, after the process group is gone (the other ranks have left; nothing of this is timed)
        out["cpu_baseline"] = cpu_baseline(weights.W_GRU, small=a.cpu_sample == "small")
        out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        # the CPU sample's OWN shapes on the GPU (BASELINE configs[0]: 16 x 8192, what the reference runs on a CPU; the low-latency
        # kernel at these batch sizes) -- like against like, beside the headline's 4096 x 65536 against 16 x 8192
        same = {}
        for key in out["cpu_baseline"]["shapes"]:
            bs, ts = (int(v) for v in key.split("x"))
            xs_ = synth_input(bs, ts, dev, seed=7)
            ms_ = []
            for i in range(4):
                ev0.record(); model.predict(xs_); ev1.record(); torch.cuda.synchronize()
                if i:
                    ms_.append(ev0.elapsed_time(ev1))
            same[key] = {"value": bs * ts / (float(np.mean(ms_)) * 1e-3), "unit": "samples/s", "ms": float(np.mean(ms_)),
                         "speedup_vs_cpu_same_shape": bs * ts / (float(np.mean(ms_)) * 1e-3) / out["cpu_baseline"]["shapes"][key]["value"]}
        out["cpu_baseline"]["same_shapes_on_the_gpu"] = same
        # CPU leg, outside the timed region: scattered streams of the LAST timed step's output (rank 0's rows) against the C
        # oracle over the whole sequence (the oracle is the checker here, never the thing measured)
        import oracle
        rows = sorted({r for r in (1, 15, 16, 17, B // 2 - 1, B // 2, B - 2, B - 1) if 0 <= r < B})
        w_or = oracle.Weights.from_state_dict({k: v.numpy() for k, v in weights.load_state_dict(weights.W_GRU).items()})
        yo, _ = oracle.gru_predict(w_or, x[rows, 0].cpu().numpy(), threads=out["cpu_baseline"]["cores"])
        yr, tr = y[rows, 0].cpu().numpy(), target[rows, 0].cpu().numpy()
        out["checks"]["streams_vs_oracle"] = {"rows": rows, "rows_of": f"rank 0 (segments {first_segment} .. {first_segment + B - 1} of the job)",
                                              "samples_each": T, "max_abs": float(np.abs(yr - yo).max()), "tolerance": 1e-5}
        if sums_last is not None:
            # the loss leg of the LAST timed step: the sums that rode in the recurrent launch against the oracle's esr_sums of
            # the same rows (same y, same target: the accumulation is what is checked, rel 1e-9), and the per-stream ESR the
            # oracle's OWN output gives against that target beside the device's
            so = oracle.esr_sums(yr, tr, INIT_LEN)
            sg = sums_last[rows].cpu().numpy()
            n = T - INIT_LEN
            esr_dev = (sg[:, 0] / n) / (sg[:, 1] / n + ESR_EPS)
            so2 = oracle.esr_sums(yo, tr, INIT_LEN)
            esr_or = (so2[:, 0] / n) / (so2[:, 1] / n + ESR_EPS)
            out["checks"]["esr_sums_vs_oracle"] = {"rows": rows, "max_rel": float(np.abs(sg / so - 1).max()), "tolerance_rel": 1e-9,
                                                   "esr_device": [float(v) for v in esr_dev],
                                                   "esr_of_oracle_output": [float(v) for v in esr_or],
                                                   "esr_max_rel_diff": float(np.abs(esr_dev / esr_or - 1).max())}
    # ---- the BASELINE configs no other driver-run line covers (configs[2], [3], the per-GPU shapes of configs[4]); after
    #      the headline's timed region and CPU leg, which they leave untouched; never part of `value`
    if world == 1 and (a.other == "on" or (a.other == "auto" and (B, T) == (4096, 65536) and a.variant in ("auto", "mfma2"))):
        del x, y, target, y_first
        torch.cuda.empty_cache()
        out["other_workloads"] = other_workloads(a, dev, check=not a.no_cpu_baseline)
    emit(out, a.detail_file)


if __name__ == "__main__":
    main()
