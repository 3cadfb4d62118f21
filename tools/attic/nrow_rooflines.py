#!/usr/bin/env python3
"""Rooflines of the "next"-row kernels that carried milliseconds only: `ntm_tape_hmag`
(code/tape.py:516-551, fp64 Jiles-Atherton RK4) and `ntm_stft_sums` (the MultiSTFT entry, code/test-model.py:25,253).

Neither is HBM-bound and neither has a GEMM in it, so each gets THREE figures, all from this one command on the MI355X:
  algorithmic   flops the textbook form of the computation needs (stated below) / measured time, against the vector peak
                of its arithmetic type (fp64 78.6 TFLOP/s, fp32 157.3 TFLOP/s; MI355X_MICROARCH.md)
  executed      flops of the vector instructions the kernel's hot loop really issues (counted in the gfx950 disassembly
                of the product source: FMA = 2, packed = x2, everything that is not arithmetic = 0), same peaks
  issue bound   the kernel's OWN bound: cycles its hot loop needs to issue on the SIMDs it can occupy (4 cycles per
                wave64 vector instruction, 16 for the quarter-rate transcendentals) against the measured cycles --
                for the tape kernel that is one wavefront per 64 streams on its own SIMD (4096 streams fill 64 of the
                1024 SIMDs and the recurrence is serial in time: the occupancy bound), for the STFT every SIMD.
A figure below 0.5 of the kernel's own bound is a to-do; the others are the explanation of why the peak fraction is what
it is.  Writes one JSON document (gpurun_out/<tag>_nrow_rooflines.json) and a short text table.

    python3 tools/nrow_rooflines.py [--static-only] [-o out.json]
"""
import json
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(ROOT, "neural-tape-modeling_amd", "csrc")
CLOCK_GHZ = 2.4
SIMDS = 256 * 4
PEAK = {"f64": 78.6, "f32": 157.3}

TRANS = re.compile(r"v_(rcp|rsq|sqrt|log|exp|sin|cos)_")                    # quarter rate
FLOPS = [(re.compile(r"v_pk_fma_f32"), 4), (re.compile(r"v_pk_(mul|add)_f32"), 2),
         (re.compile(r"v_(fma|fmac|mad|mac)_f(32|64)"), 2), (re.compile(r"v_(mul|add|sub|subrev|max|min)_f(32|64)"), 1),
         (re.compile(r"v_div_fmas_f64"), 2), (re.compile(r"v_(rcp|rsq|sqrt|log|exp)_f(32|64)"), 1)]


def hot_loop(src, kernel_regex, defines=()):
    """-> {kernel symbol: instruction mix of its largest loop} for every kernel of `src` matching `kernel_regex`."""
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        asm = os.path.join(tmp, "k.s")
        subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"), "-Wno-unused-function",
                        "-mllvm", "-amdgpu-mfma-vgpr-form", "-S", "--cuda-device-only", "-o", asm, src] + list(defines),
                       check=True, stderr=subprocess.DEVNULL)
        text = open(asm).read()
    out = {}
    for m in re.finditer(r"^(\w+):[^\n]*\n(.*?)\n\s*\.amdhsa_kernel", text, flags=re.S | re.M):
        name, body = m.group(1), m.group(2)
        if not re.search(kernel_regex, name):
            continue
        ins, labels = [], {}
        for ln in body.splitlines():
            lm = re.match(r"\s*(\.LBB\w+):", ln)
            if lm:
                labels[lm.group(1)] = len(ins)
                continue
            t = ln.split(";")[0].strip()
            if t and not t.startswith("."):
                ins.append(t)
        # loops = backward branches; several branches to one header are one loop (the longest span).  The hot loop is the
        # INNERMOST one that still holds >= 80 % of the vector instructions of the largest (outer loops add housekeeping).
        spans = {}
        for i, t in enumerate(ins):
            bm = re.match(r"s_c?branch\w*\s+(\.LBB\w+)", t)
            if bm and labels.get(bm.group(1), i + 1) <= i:
                a = labels[bm.group(1)]
                spans[a] = max(spans.get(a, a), i)
        cands = [(sum(1 for u in ins[a:b + 1] if u.startswith("v_")), b - a + 1, a, b) for a, b in spans.items()]
        most = max(c[0] for c in cands)
        _, _, a, b = min((c for c in cands if c[0] >= 0.8 * most), key=lambda c: c[1])
        seg = ins[a:b + 1]
        ops = Counter(u.split()[0] for u in seg)
        valu = {k: v for k, v in ops.items() if k.startswith("v_")}
        flops = sum(v * next((f for rx, f in FLOPS if rx.match(k)), 0) for k, v in valu.items())
        trans = sum(v for k, v in valu.items() if TRANS.match(k))
        out[name] = {"instructions": len(seg), "valu": sum(valu.values()), "valu_transcendental": trans,
                     "salu": sum(v for k, v in ops.items() if k.startswith("s_") and not k.startswith(("s_waitcnt", "s_nop"))),
                     "lds": sum(v for k, v in ops.items() if k.startswith("ds_")),
                     "vmem": sum(v for k, v in ops.items() if k.startswith(("global_", "buffer_", "flat_"))),
                     "flops_per_lane": flops, "issue_cycles": 4 * (sum(valu.values()) - trans) + 16 * trans,
                     "top_ops": dict(sorted(valu.items(), key=lambda kv: -kv[1])[:12])}
    return out


def measure():
    import torch
    sys.path.insert(0, ROOT)
    import ntm_amd
    from ntm_amd.model import stft_sums, MRSTFT_FFT_SIZES, MRSTFT_HOP_SIZES, MRSTFT_WIN_LENGTHS, STFT_EPS
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]

    def timed(fn, reps=5):
        ms = []
        for i in range(reps + 1):
            ev[0].record(); fn(); ev[1].record(); torch.cuda.synchronize()
            if i:
                ms.append(ev[0].elapsed_time(ev[1]))
        return sum(ms) / len(ms)
    res = {"device": torch.cuda.get_device_name(0)}
    tape = {}
    for B in (4096, 65536):
        N = 16384 if B == 4096 else 4096
        H = 8000.0 * torch.sin(torch.arange(N, device="cuda", dtype=torch.float64)[None, :] * 0.01 + torch.rand(B, 1, device="cuda", dtype=torch.float64))
        tp = ntm_amd.TapeMagnetization(batch_size=B)
        tape[f"{B}x{N}"] = {"B": B, "N": N, "ms": timed(lambda: tp.H_mag(H), 3)}
        del H, tp
    res["tape_hmag"] = tape
    B, T, skip = 4096, 65536, 1024
    g = torch.Generator(device="cuda").manual_seed(1)
    t = 0.3 * torch.randn(B, 1, T, device="cuda", generator=g)
    y = t + 0.02 * torch.randn(B, 1, T, device="cuda", generator=g)
    st = {}
    for n_fft, hop, win in zip(MRSTFT_FFT_SIZES, MRSTFT_HOP_SIZES, MRSTFT_WIN_LENGTHS):
        st[str(n_fft)] = {"n_fft": n_fft, "hop": hop, "win": win, "B": B, "T": T, "skip": skip, "frames": 1 + (T - skip) // hop,
                          "ms": timed(lambda: stft_sums(y, t, skip, n_fft, hop, win, STFT_EPS))}
    res["stft_sums"] = st
    res["mrstft_ms"] = timed(lambda: ntm_amd.MRSTFTLoss().per_segment(y, t, skip))
    return res


def main(argv):
    out = {"peaks_tflops": PEAK, "clock_ghz": CLOCK_GHZ, "simds": SIMDS}
    tape_mix = list(hot_loop(os.path.join(CSRC, "tape_kernels.hip"), r"tape_hmag_kernel").values())[0]
    stft_mix = {int(re.search(r"ILi(\d+)ELi0E", k).group(1)): v
                for k, v in hot_loop(os.path.join(CSRC, "stft_kernels.hip"), r"stft_sums_kernelILi\d+ELi0E").items()}
    out["static"] = {"tape_hmag_kernel (one sample of one wavefront = 64 streams)": tape_mix,
                     "stft_sums_kernel<log2 n_fft, mode 0> (one frame PAIR of one wavefront: the y and the target frame of one hop position through ONE complex FFT)": {str(1 << k): v for k, v in sorted(stft_mix.items())}}
    meas = None if "--static-only" in argv else measure()
    table = []
    if meas:
        out["measured"] = meas
        # ---- tape: textbook count per sample of one stream: 4 RK4 stages x (Langevin L(Q) = coth Q - 1/Q with one exp and two
        # divisions ~ 40 flop, L'(x) = 1/x^2 - coth^2 x + 1 ~ 45 flop with its own exp, ~25 flop of products / sums, one
        # division) + the RK4 combination and the trapezoidal derivative (~15) = about 460 flop
        ALG_TAPE = 460.0
        rows = {}
        for key, m in meas["tape_hmag"].items():
            B, N, sec = m["B"], m["N"], m["ms"] * 1e-3
            waves = (B + 63) // 64
            per_simd = -(-waves // SIMDS)                              # wavefronts that share a SIMD
            cyc_per_sample = sec * CLOCK_GHZ * 1e9 / N / per_simd      # measured cycles one wavefront spends per sample
            r = {"samples_per_s": B * N / sec, "ms": m["ms"],
                 "algorithmic": {"flop_per_sample": ALG_TAPE, "tflops": ALG_TAPE * B * N / sec / 1e12, "frac_of_fp64_peak": ALG_TAPE * B * N / sec / 1e12 / PEAK["f64"]},
                 "executed": {"flop_per_sample": tape_mix["flops_per_lane"], "tflops": tape_mix["flops_per_lane"] * B * N / sec / 1e12,
                              "frac_of_fp64_peak": tape_mix["flops_per_lane"] * B * N / sec / 1e12 / PEAK["f64"]},
                 "issue_bound": {"wavefronts": waves, "simds_occupied": min(waves, SIMDS), "wavefronts_per_simd": per_simd,
                                 "issue_cycles_per_sample": tape_mix["issue_cycles"], "measured_cycles_per_sample_and_wavefront": cyc_per_sample,
                                 "frac": tape_mix["issue_cycles"] / cyc_per_sample,
                                 "what": "one lane per stream, the recurrence is serial in time: a wavefront can at best issue its hot loop back to back"}}
            rows[key] = r
            table.append(f"tape_hmag {key}: {m['ms']:.2f} ms = {B * N / sec / 1e9:.2f} G samples/s | algorithmic {r['algorithmic']['frac_of_fp64_peak']:.4f} "
                         f"executed {r['executed']['frac_of_fp64_peak']:.4f} of the fp64 peak | own (issue / occupancy) bound {r['issue_bound']['frac']:.2f} "
                         f"({waves} wavefronts on {min(waves, SIMDS)} of {SIMDS} SIMDs)")
        out["tape_hmag"] = rows
        rows = {}
        for key, m in meas["stft_sums"].items():
            n, frames, B, sec = m["n_fft"], m["frames"], m["B"], m["ms"] * 1e-3
            pairs = frames        # a "frame pair" = the y frame and the target frame of ONE hop position, packed as z = y + i t
            lg = n.bit_length() - 1
            alg = 5.0 * n * lg + 6.0 * n + 10.0 * (n // 2 + 1)     # complex FFT + window of two signals + |.|^2, sqrt, log, sums of two spectra
            mix = stft_mix[lg]
            ex = mix["flops_per_lane"] * 64.0
            issue = mix["issue_cycles"] * B * pairs / SIMDS / (CLOCK_GHZ * 1e9)
            r = {"ms": m["ms"], "frame_pairs_per_stream (= frames: y and target frame of a hop position)": pairs,
                 "algorithmic": {"flop_per_frame_pair": alg, "tflops": alg * B * pairs / sec / 1e12, "frac_of_fp32_peak": alg * B * pairs / sec / 1e12 / PEAK["f32"]},
                 "executed": {"flop_per_frame_pair": ex, "tflops": ex * B * pairs / sec / 1e12, "frac_of_fp32_peak": ex * B * pairs / sec / 1e12 / PEAK["f32"]},
                 "issue_bound": {"issue_cycles_per_frame_pair": mix["issue_cycles"], "seconds_if_every_simd_only_issued": issue, "frac": issue / sec,
                                 "what": "vector-issue time of the frame-pair loop over all 1024 SIMDs / measured time (VALU busy)"}}
            rows[key] = r
            table.append(f"stft_sums n_fft {n}: {m['ms']:.2f} ms | algorithmic {r['algorithmic']['frac_of_fp32_peak']:.3f} executed "
                         f"{r['executed']['frac_of_fp32_peak']:.3f} of the fp32 peak | own (vector issue) bound {r['issue_bound']['frac']:.2f}")
        out["stft_sums"] = rows
        tot = sum(m["ms"] for m in meas["stft_sums"].values())
        table.append(f"MRSTFTLoss.per_segment (3 resolutions + host arithmetic): {meas['mrstft_ms']:.2f} ms (kernels alone {tot:.2f})")
    out["table"] = table
    text = json.dumps(out, indent=1)
    if "-o" in argv:
        with open(argv[argv.index("-o") + 1], "w") as f:
            f.write(text)
    else:
        print(text)
    for ln in table:
        print(ln, file=sys.stderr)


if __name__ == "__main__":
    main(sys.argv[1:])
