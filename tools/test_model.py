#!/usr/bin/env python3
"""The reference's evaluation CLI (code/test-model.py) on the MI355X engine: the SAME command line --

    python -u tools/test_model.py --MODEL GRU --WEIGHTS "GRU-HS[64]-L[DCPreESR]-DS[...]_BEST" \
        --DATASET ReelToReel_Dataset_MiniPulse100_CHOWTAPE_WOWFLUTTER --SUBSET Test --NO_SHUFFLE \
        --SEGMENT_LENGTH 441000 --ADD_DELAY --COMPUTE_LOSS --SAVE_AUDIO --DESCRIPTIVE_NAME LOSS

is scripts/test-model-loss.sh:57-63 verbatim.  Every flag of code/test-model.py:45-85 parses with the reference's type
and default; what is built behind them is the part of the script that drives the model:

  model construction from the weights-directory names   code/test-model.py:192-247   (all `--WEIGHTS`, nargs='+')
  `--COMPUTE_LOSS`: loss over the dataset                code/test-model.py:296-418   (last model of `--WEIGHTS` and the
                                                                                       `--MODEL` string, as upstream)
  the example prediction on segment `--IDX`              code/test-model.py:420-552   (+ `--SAVE_AUDIO`, :876-893)

batched over segments (`--BATCH_SIZE`, the reference loops with BATCH_SIZE = 1, :115) and shardable over ranks.
Presentation and generator flags (`--SAVE_FIG --PLOT_* --ZOOM --ADD_NOISE --NOISE_TYPE --DATASET_NOISE`, `--DELAY_TYPE
Generated`) are out of scope (matplotlib figures, the diffusion generators, the noise dataset): they parse and are
reported in one "out of scope, ignored" line.

Paths: `--DATASET <name>` is looked up under `--AUDIO_PATH` (default ../audio/, code/test-model.py:107,147) unless
`--DATASET_DIR` names the directory itself; a `--WEIGHTS` entry is `<MODEL_PATH>/<name>/best.pth` when that exists
(default ../weights/, :119,198-199,233), a directory holding best.pth, or one of the 44 exported checkpoints
(ntm_amd.weights.available()).  Delay trajectories come from the `trajectory_<id>_*.npy` side-cars, which the feeder
computes and caches on first use for stereo datasets exactly as DelayAnalyzer does; `--MAX_DELAY` (seconds) overrides
the dataset's measured maximum.  Extra flags of this build: --DATASET_DIR --AUDIO_PATH --MODEL_PATH --RESULTS_PATH
--TEMP_PATH --NO_CACHE --BATCH_SIZE --STREAM_CHUNK --MAX_DELAY --INIT_LEN --KERNEL --SEED --NO_EXAMPLE.
"""
import argparse
import json
import os
import time
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ntm_amd  # noqa: E402
from ntm_amd import distributed as D  # noqa: E402
from ntm_amd.feeder import SegmentFeeder  # noqa: E402
from ntm_amd.model import ESR_EPS, esr_dcpre_sums, esr_sums  # noqa: E402
from ntm_amd.utilities import nextpow2, parse_hidden_size, parse_loss, parse_model  # noqa: E402

OUT_OF_SCOPE = ("SAVE_FIG", "PLOT_SWEEP", "PLOT_TRANSFER", "PLOT_PHASE", "PLOT_DELAY", "ZOOM", "ADD_NOISE")
MODEL_IDS = {                                            # code/test-model.py:206-212
    'GRU-DCPreESR': 'Supervised 1',
    'GRU-ESR': 'Supervised 1',
    'DiffDelGRU-DCPreESR': 'Supervised 2',
    'DiffDelGRU-ESR': 'Supervised 2',
    'DiffDelGRU-LogSpec': 'Adversarial',
}


def none_or_int(argument):
    """ Parse NoneType or int input arguments from CLI (code/test-model.py:38-42) """
    if argument == 'None':
        return None
    return int(argument)


def build_parser():
    p = argparse.ArgumentParser(description="Evaluate a trained tape model over a dataset (code/test-model.py).")
    # GLOBAL (code/test-model.py:48-50)
    p.add_argument('--DESCRIPTIVE_NAME', type=str, default=None)
    p.add_argument('--SAVE_FIG', action='store_true', default=False)
    p.add_argument('--SAVE_AUDIO', action='store_true', default=False)
    # MODEL (:53-61).  The reference's default is a bare str, which `for weight in WEIGHTS` would walk character by
    # character; a one-element list here.  It lacks the -L[...] field, so parse_loss raises on it as upstream would.
    p.add_argument('--MODEL', type=str, default="GRU")
    p.add_argument('--WEIGHTS', nargs='+', default=["GRU-HS[64]-DS[ReelToReel_Dataset_MiniPulse100_CHOWTAPE]"])
    p.add_argument('--ADD_DELAY', action='store_true', default=False)
    p.add_argument('--DELAY_TYPE', type=str, default="Real")
    p.add_argument('--ADD_NOISE', action='store_true', default=False)
    p.add_argument('--NOISE_TYPE', type=str, default="Real")
    # DATASET (:64-77)
    p.add_argument('--DATASET', type=str, default="ReelToReel_Dataset_MiniPulse100_CHOWTAPE")
    p.add_argument('--SUBSET', type=str, default="Val")
    p.add_argument('--FRACTION', type=float, default=1.0)
    p.add_argument('--SEGMENT_LENGTH', type=int, default=None)
    p.add_argument('--NO_SHUFFLE', action='store_true', default=False)
    p.add_argument('--DEMODULATE', action='store_true', default=False)
    p.add_argument('--IDX', type=none_or_int, default=None)
    p.add_argument('--SYNC', type=float, default=0.0)
    p.add_argument('--COMPUTE_LOSS', action='store_true', default=False)
    p.add_argument('--DATASET_NOISE', type=str, default="Silence_AKAI_IPS[7.5]_MAXELL")
    # VISUALIZATIONS (:80-84)
    p.add_argument('--PLOT_SWEEP', action='store_true', default=False)
    p.add_argument('--PLOT_TRANSFER', action='store_true', default=False)
    p.add_argument('--PLOT_PHASE', action='store_true', default=False)
    p.add_argument('--PLOT_DELAY', action='store_true', default=False)
    p.add_argument('--ZOOM', type=float, default=None)
    # this build
    p.add_argument('--DATASET_DIR', type=str, default=None, help="the dataset directory itself (overrides AUDIO_PATH/DATASET)")
    p.add_argument('--AUDIO_PATH', type=str, default="../audio/")
    p.add_argument('--MODEL_PATH', type=str, default="../weights/")
    p.add_argument('--RESULTS_PATH', type=str, default="../results/")
    p.add_argument('--TEMP_PATH', type=str, default=".temp/", help="where the loss results are cached (code/test-model.py:304-320)")
    p.add_argument('--NO_CACHE', action='store_true', default=False, help="neither read nor write the loss cache under TEMP_PATH")
    p.add_argument('--BATCH_SIZE', type=int, default=4096, help="segments per launch (the matrix-pipe kernel wants thousands)")
    p.add_argument('--STREAM_CHUNK', type=int, default=8192, help="time chunk of the host->device pipeline; 0 = whole-batch copies")
    p.add_argument('--MAX_DELAY', type=float, default=0.0, help="seconds (DelayAnalyzer.max_delay of the dataset)")
    p.add_argument('--INIT_LEN', type=int, default=None,
                   help="samples cut from the head of every segment before the losses; default: the reference's "
                        "nextpow2(int(max_delay * fs)) (code/test-model.py:323-324), i.e. 2 for a dataset without delay")
    p.add_argument('--KERNEL', type=str, default="auto")
    p.add_argument('--SEED', type=int, default=None, help="seed of the shuffled draw / the random example (upstream: unseeded)")
    p.add_argument('--NO_EXAMPLE', action='store_true', default=False, help="skip the example prediction after the loss")
    return p


def parse_args(argv=None):
    """Parse + the reference's own argument checks (code/test-model.py:91-99)."""
    a = build_parser().parse_args(argv)
    if isinstance(a.WEIGHTS, str):
        a.WEIGHTS = [a.WEIGHTS]
    assert not (a.PLOT_SWEEP and a.PLOT_TRANSFER and a.PLOT_DELAY), "Choose either PLOT_SWEEP, PLOT_TRANSFER or PLOT_DELAY"
    assert a.DELAY_TYPE.lower() in ["real", "generated", "true"], "Choose 'Real', 'Generated' or 'True' as DELAY_TYPE"
    assert a.NOISE_TYPE.lower() in ["real", "generated"], "Choose 'Real' or 'Generated' as NOISE_TYPE"
    return a


def dataset_path(a):
    """code/test-model.py:147: os.path.join(AUDIO_PATH, DATASET)."""
    return a.DATASET_DIR if a.DATASET_DIR else os.path.join(a.AUDIO_PATH, a.DATASET)


def dataset_name(a):
    return os.path.basename(os.path.normpath(a.DATASET_DIR)) if a.DATASET_DIR else a.DATASET


def resolve_weights(weight, model_path):
    """-> (weights-directory name, state_dict or None = take the exported checkpoint of that name)."""
    for cand in (os.path.join(model_path, weight), weight):
        best = os.path.join(cand, "best.pth")
        if os.path.isfile(best):
            return os.path.basename(os.path.normpath(cand)), torch.load(best, map_location="cpu")   # code/test-model.py:233
    name = os.path.basename(os.path.normpath(weight))
    if name not in ntm_amd.weights.available():
        parse_loss(name)              # a malformed name fails here the way it does upstream (:204)
        raise SystemExit(f"no best.pth under '{os.path.join(model_path, weight)}' and '{name}' is not an exported checkpoint")
    return name, None


def check_model_flag(model_flag, weights):
    """The loss loop dispatches on the `--MODEL` string and runs the LAST entry of `--WEIGHTS` (code/test-model.py:345-353);
    upstream a mismatch ends in a TypeError inside predict().  Here it is an error up front."""
    last = parse_model(os.path.basename(os.path.normpath(weights[-1])))
    if model_flag != last:
        raise SystemExit(f"--MODEL {model_flag} does not match the model type '{last}' of --WEIGHTS {weights[-1]}")


def write_wav16(path, x, fs):
    """`sf.write(path, x, fs)` of code/test-model.py:884-893: libsndfile's default for .wav is 16-bit PCM."""
    from scipy.io import wavfile
    x = np.asarray(x, np.float32).reshape(-1)
    wavfile.write(path, int(fs), np.clip(np.rint(x * 32767.0), -32768, 32767).astype(np.int16))


class StageTimes:
    """Stage times of one loss run (`main(argv, profile={})`; bench.py's `other_workloads.cli`): HIP events around every
    stage of every batch on the compute stream, host clocks around the dataset decode and the whole loop."""

    def __init__(self):
        self.events, self.h2d, self.host = {}, [], {}

    def span(self, name):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.events.setdefault(name, []).append((e0, e1))
        e0.record()
        return e1

    def summary(self):
        torch.cuda.synchronize()
        out = {k + "_ms": sum(a.elapsed_time(b) for a, b in v) for k, v in self.events.items()}
        if self.h2d:
            # what the feeder's per-batch copies were: host-to-device DMA from the pinned set, or -- a device-resident set --
            # device-to-device gathers (`batch_copy_kind`; the h2d_* names are kept for older readers of the profile)
            out["h2d_ms"] = out["batch_copy_ms"] = sum(a.elapsed_time(b) for a, b, _ in self.h2d)
            out["h2d_bytes"] = out["batch_copy_bytes"] = sum(n for _, _, n in self.h2d)
            out["batch_copy_kind"] = "device_to_device_gather" if self.host.get("resident") else "host_to_device"
        out.update(self.host)
        return out


def main(argv=None, profile=None):
    """`profile`: a dict that receives the stage times of the loss run (StageTimes.summary); None = no instrumentation."""
    a = parse_args(argv)
    rank, world, local = D.init_from_env()
    say = print if rank == 0 else (lambda *x, **k: None)
    say("\nArguments:")
    say(a)
    ignored = [f"--{k}" for k in OUT_OF_SCOPE if getattr(a, k)]
    if a.DELAY_TYPE.lower() == "generated":
        raise SystemExit("--DELAY_TYPE Generated needs the diffusion trajectory generator (code/model.py DiffusionGenerator): out of scope")
    if ignored:
        say("out of scope, ignored: " + " ".join(ignored) + "  (figures / noise: see INTEGRATION.md)")
    check_model_flag(a.MODEL, a.WEIGHTS)
    torch.cuda.set_device(local)
    if a.SEED is not None:
        np.random.seed(a.SEED)

    # ---- dataset (code/test-model.py:145-154,178-179)
    say("Dataset (audio)")
    t_decode = time.perf_counter()
    feeder = SegmentFeeder(dataset_path(a), subset=a.SUBSET, length=a.SEGMENT_LENGTH, sync=a.SYNC, demodulate=a.DEMODULATE,
                           fraction=a.FRACTION, shuffle=not a.NO_SHUFFLE, seed=a.SEED)
    fs = feeder.fs
    stages = StageTimes() if profile is not None else None
    if stages is not None:      # WAV decode into pinned host memory + trajectory side-cars (read, or analysed on first use)
        stages.host["decode_s"] = time.perf_counter() - t_decode
        stages.host["segments"], stages.host["segment_length"], stages.host["resident"] = len(feeder), feeder.length, feeder.resident
    if a.MAX_DELAY <= 0 and feeder.max_delay > 0:          # dataset.delay_analyzer.max_delay (code/test-model.py:223,323-324)
        a.MAX_DELAY = feeder.max_delay

    # ---- models (code/test-model.py:192-247)
    models = []
    for weight in a.WEIGHTS:
        name, sd = resolve_weights(weight, a.MODEL_PATH)
        model_type, training_loss = parse_model(name), parse_loss(name)
        model = ntm_amd.harness.build_model(name, max_delay_seconds=a.MAX_DELAY, fs=fs, state_dict=sd)
        model.kernel_variant = a.KERNEL
        md = {"weight": name, "model_type": model_type,
              "model_id": MODEL_IDS.get(f"{model_type}-{training_loss}", f"{model_type}-{training_loss}"), "model": model}
        if a.ADD_DELAY:
            md["delay"] = ntm_amd.TimeVaryingDelayLine(max_delay=int(1.25 * feeder.max_delay * fs))     # :236-240
        say("=" * 25)
        for key, value in md.items():
            say(f"{key.ljust(9)}: {value if key != 'model' else type(value).__name__ + f'(hidden_size={parse_hidden_size(name)})'}")
        say("=" * 25, "\n")
        models.append(md)
    model, name = models[-1]["model"], models[-1]["weight"]          # what the loss loop sees upstream (:345-353)
    delay = models[-1].get("delay")
    is_dd = isinstance(model, ntm_amd.DiffDelRNN)
    # code/test-model.py:323-324: INIT_LEN = nextpow2(int(max_delay * fs)) -- for max_delay == 0 that is 2 (the
    # reference's own "# 2**10" comment there is wrong: nextpow2(0) == 2)
    init_len = a.INIT_LEN if a.INIT_LEN is not None else ntm_amd.harness.init_len(a.MAX_DELAY, fs)

    results = {}
    if a.COMPUTE_LOSS:
        results = compute_loss(a, feeder, model, [m["weight"] for m in models], delay, is_dd, init_len, rank, world, say, stages)
        if stages is not None:
            profile.update(stages.summary())
    else:
        say(f"{len(feeder)} segments of {feeder.length} samples @ {fs} Hz; no loss without --COMPUTE_LOSS")
    if rank == 0 and not a.NO_EXAMPLE:
        example_prediction(a, feeder, models, init_len, say)
    return results


def loss_cache_key(a, feeder, names, init_len):
    """Everything that changes the numbers of the loss loop but is not in upstream's cache file name
    (code/test-model.py:301-304 keys on WEIGHTS / ADD_DELAY / DEMODULATE / ADD_NOISE + dataset basename + subset): stored
    INSIDE the cache file and compared on load, so a run with another segment length, seed, kernel ... recomputes."""
    return {"weights": list(names), "dataset": os.path.abspath(dataset_path(a)), "subset": str(feeder.subset),
            "segment_length": int(feeder.length), "segments": int(len(feeder)), "fraction": float(a.FRACTION),
            "shuffle": not a.NO_SHUFFLE, "seed": a.SEED, "sync": float(a.SYNC), "init_len": int(init_len),
            "max_delay": float(a.MAX_DELAY), "kernel": str(a.KERNEL), "add_delay": bool(a.ADD_DELAY),
            "demodulate": bool(a.DEMODULATE), "add_noise": bool(a.ADD_NOISE)}


def compute_loss(a, feeder, model, names, delay, is_dd, init_len, rank, world, say, stages=None):
    """code/test-model.py:296-418: per-segment losses, mean over segments, cached under TEMP_PATH as upstream.
    `names`: the resolved weight names of --WEIGHTS (a --WEIGHTS entry may be a directory path: its slashes must not reach
    the cache file name)."""
    fs = feeder.fs
    say("\nComputing loss over dataset ...", end="")
    save_path = os.path.join(a.TEMP_PATH, 'loss', f"{dataset_name(a)}", f"{feeder.subset}")
    safe = [str(n).replace(os.sep, "_").replace("/", "_") for n in names]
    save_name = f"{safe}_DELAY[{a.ADD_DELAY}]_DEMODULATE[{a.DEMODULATE}]_NOISE[{a.ADD_NOISE}].npy"
    cached = os.path.join(save_path, save_name)
    cache_key = loss_cache_key(a, feeder, names, init_len)
    results = None
    keyfile = cached[:-4] + ".key.json"
    if os.path.exists(cached) and not a.NO_CACHE:
        # every rank takes the same decision: it depends on the files and the arguments only.  The .npy holds what upstream's
        # holds -- a plain {loss name: float} dict (code/test-model.py:399-403; its stats loop formats every value with
        # '{:.6f}') -- so either program can load the other's cache; the argument record lives in a side-car beside it
        # (<name>.key.json).  A cache without a side-car (upstream's own) carries no record of its arguments: recomputed.
        why, blob, rec = "unreadable", None, None
        try:
            blob = np.load(cached, allow_pickle=True).item()
        except Exception as e:
            why = f"unreadable: {type(e).__name__}"
        try:
            with open(keyfile) as f:
                rec = json.load(f)
        except (OSError, ValueError):
            rec = None
        if isinstance(blob, dict) and isinstance(rec, dict) and rec.get("key") == json.loads(json.dumps(cache_key)):
            say(" Loading pre-computed!")
            results = {k: float(v) for k, v in blob.items() if not k.startswith("_")}
            n_seg = rec.get("segments", len(feeder))
        else:
            if isinstance(blob, dict):
                old = rec.get("key") if isinstance(rec, dict) else None
                why = ("differs in " + ", ".join(sorted(k for k in set(old) | set(cache_key) if old.get(k) != cache_key.get(k)))
                       if isinstance(old, dict) else "no argument record")
            say(f" (cached results belong to other arguments [{why}]: recomputing)", end="")
    if results is None:
        say(" Starting analysis...")
        per = {"ESR": [], "DCPreESR": [], "MultiSTFT": []}
        mrstft = ntm_amd.MRSTFTLoss()
        seg_len = feeder.length - (int(feeder.mean_delay * fs) if a.DEMODULATE else 0)    # demodulation trims the tail
        with_stft = seg_len - init_len > 1024                  # the largest STFT frame needs > 1024 samples

        span = (lambda name: stages.span(name)) if stages is not None else (lambda name: None)
        done = lambda e: e.record() if e is not None else None                       # noqa: E731
        t_loop = time.perf_counter()

        # plain GRU on whole-batch copies (--STREAM_CHUNK 0): predict + the ESR and DCPreESR sums in ONE launch where the
        # matrix-pipe kernel runs (RNN.predict_losses / ntm_gru_forward_losses); needs INIT_LEN to be a multiple of 4
        # (the DiffDelGRU takes the batched path whenever --ADD_DELAY is given, as the loss script does: there too ONE launch)
        fused_losses = (not a.DEMODULATE and a.KERNEL == "auto" and init_len % 4 == 0
                        and ((not is_dd and delay is None and a.STREAM_CHUNK <= 0) or (is_dd and (a.ADD_DELAY or a.STREAM_CHUNK <= 0))))

        def batches():
            if a.DEMODULATE or a.ADD_DELAY or a.STREAM_CHUNK <= 0:
                for xin, tgt, dt, _ in feeder.batches(a.BATCH_SIZE, "cuda", rank, world, timing=None if stages is None else stages.h2d):
                    pre = None
                    e = span("predict+ESR+DCPreESR" if fused_losses else "predict")
                    if is_dd:
                        assert dt is not None, "DiffDelGRU needs trajectory_<id>_*.npy side-cars"
                        if fused_losses:
                            out, _, s_esr, s_dc = model.predict_losses(xin, dt * fs, tgt, skip=init_len)
                            pre = {"ESR": s_esr, "DCPreESR": s_dc}
                        else:
                            out, _ = model.predict(xin, dt * fs)
                    elif fused_losses:
                        out, s_esr, s_dc = model.predict_losses(xin, tgt, skip=init_len)
                        pre = {"ESR": s_esr, "DCPreESR": s_dc}
                    else:
                        out = model.predict(xin)
                    done(e)
                    yield xin, tgt, out, dt, pre
            else:
                # predict straight from the feeder's pinned files, H2D copies pipelined along time under the launches
                lo, hi = D.shard_range(len(feeder), rank, world)
                for b0 in range(lo, hi, a.BATCH_SIZE):
                    e = span("predict_streamed")             # H2D chunks pipelined under the launches: one stage
                    out, xin, tgt = feeder.predict_streamed(model, b0, min(hi, b0 + a.BATCH_SIZE), chunk=a.STREAM_CHUNK)
                    done(e)
                    yield xin, tgt, out, None, None

        if delay is not None and not is_dd:
            assert feeder.max_delay > 0, "--ADD_DELAY needs delay trajectories (stereo dataset or side-cars)"
        for xin, tgt, out, dt, pre in batches():
            if delay is not None and not is_dd:                                   # :355-364 (`ADD_DELAY and MODEL == "GRU"`)
                e = span("apply_delay")
                out = ntm_amd.harness.apply_delay(delay, dt * fs, out)
                done(e)
            n = xin.shape[-1] - init_len
            for key, fn in (("ESR", esr_sums), ("DCPreESR", esr_dcpre_sums)):
                if pre is not None:
                    s = pre[key]
                    per[key].append((s[:, 0] / n) / (s[:, 1] / n + ESR_EPS))
                    continue
                e = span(key)
                s = fn(out, tgt, skip=init_len)
                per[key].append((s[:, 0] / n) / (s[:, 1] / n + ESR_EPS))
                done(e)
            if with_stft:
                e = span("MultiSTFT")
                per["MultiSTFT"].append(mrstft.per_segment(out, tgt, skip=init_len))
                done(e)
        # every rank issues the SAME collectives whatever its shard holds (a rank with no segments -- more ranks than
        # segments -- reduces empty tensors): the key set depends on the global segment length only
        if not with_stft:
            del per["MultiSTFT"]
        res = {k: D.reduce_loss_sums(torch.cat(v) if v else torch.zeros(0, device="cuda", dtype=torch.float64))
               for k, v in per.items()}
        n_seg = res['ESR']['segments']
        results = {k: v["mean_segment_loss"] for k, v in res.items()}
        if stages is not None:                         # (reduce_loss_sums fetched the scalars: the device is idle here)
            stages.host["loss_loop_s"] = time.perf_counter() - t_loop
        if rank == 0 and not a.NO_CACHE:
            try:                                   # a cache that cannot be written must never cost the printed results
                os.makedirs(save_path, exist_ok=True)
                np.save(cached, {k: float(v) for k, v in results.items()})          # upstream's format exactly
                with open(keyfile, "w") as f:
                    json.dump({"key": cache_key, "segments": int(n_seg)}, f, indent=1)
            except OSError as e:
                say(f"(loss cache not written: {e})")
    say()
    say("=" * 5, "Stats:", "=" * 5)
    say(f"Model:      {a.WEIGHTS}")
    say(f"Dataset:    {dataset_name(a)}")
    say(f"Subset:     {a.SUBSET}")
    say(f"ADD_DELAY:  {a.ADD_DELAY}")
    say(f"DEMODULATE: {a.DEMODULATE}")
    say(f"ADD_NOISE:  {a.ADD_NOISE}")
    say(f"Segments:   {n_seg}\n")
    for key, value in results.items():
        say(f"{key.ljust(9)}: {'{:.6f}'.format(value)}")
    say()
    say("=" * 18)
    return results


@torch.no_grad()
def example_prediction(a, feeder, models, init_len, say):
    """code/test-model.py:420-552 without the figures: every model of `--WEIGHTS` on segment `--IDX` (random when None),
    the DiffDelGRU with the trajectory `--DELAY_TYPE` selects, `--ADD_DELAY` for GRU models, per-second losses of the
    example with `--COMPUTE_LOSS` (:528-551), WAV export with `--SAVE_AUDIO` (:876-893)."""
    fs = feeder.fs
    say("\nMaking example prediction ...")
    sample_idx = np.random.randint(0, high=len(feeder)) if a.IDX is None else a.IDX
    input, target, meta = feeder[sample_idx]
    input, target = input[None, :1, :].cuda(), target[None, :1, :].cuda()
    say(f"idx = {sample_idx}")
    say(f"filename: {meta['target_name']}")
    d_traj = None
    if "delay_trajectory" in meta:
        if a.DELAY_TYPE == "True":
            src = meta
        else:                                         # "Real": the trajectory of a random example (:456-464)
            _, __, src = feeder[np.random.randint(0, high=len(feeder))]
        d_traj = (torch.as_tensor(src['delay_trajectory']).float().view(1, 1, -1) * fs).cuda()
    outs = {}
    for md in models:
        model = md['model']
        x_in, tgt = input, target
        if md['model_type'] == "GRU":
            output = model.predict(x_in)
        else:
            assert d_traj is not None, "DiffDelGRU needs trajectory_<id>_*.npy side-cars"
            output, output_pre_d = model.predict(x_in, d_traj)
            if not (a.PLOT_TRANSFER or a.PLOT_SWEEP):
                x_in, tgt, output = x_in[:, :, init_len:], tgt[:, :, init_len:], output[:, :, init_len:]     # :486-496
            else:
                output = output_pre_d                 # :497-498: the transfer / sweep figures look at the nonlinearity alone
        if a.ADD_DELAY and a.MODEL == "GRU" and md['model_type'] == "GRU":
            assert d_traj is not None, "--ADD_DELAY needs delay trajectories (stereo dataset or side-cars)"
            output = ntm_amd.harness.apply_delay(md['delay'], d_traj, output, segment_length=2**12)       # :259-290
            start_delay = int(d_traj[0, 0, 0])                                                           # :512-521
            x_in, tgt, output = x_in[:, :, start_delay:], tgt[:, :, start_delay:], output[:, :, start_delay:]
        outs[md['model_id']] = output
        if a.COMPUTE_LOSS and output.shape[-1] > 0:
            SEG = 44100                                                                                  # :530
            nseg = int(np.ceil(output.shape[-1] / SEG))
            tot = {"ESR": 0.0, "DCPreESR": 0.0}
            for k in range(nseg):
                o, t = output[:, :, k * SEG:(k + 1) * SEG], tgt[:, :, k * SEG:(k + 1) * SEG]
                tot["ESR"] += float(ntm_amd.model.ESRLoss()(o, t))
                tot["DCPreESR"] += float(ntm_amd.model.DCPreESR()(o, t))
            say(f"example loss [{md['model_id']}]: " + ", ".join(f"{k} {v / nseg:.6f}" for k, v in tot.items()))
        md['example'] = (x_in, tgt, output)
    if a.SAVE_AUDIO:
        say("\nSaving audio...")
        basename = dataset_name(a) + (f"_{a.DESCRIPTIVE_NAME}" if a.DESCRIPTIVE_NAME else "")
        os.makedirs(a.RESULTS_PATH, exist_ok=True)
        x_in, tgt, _ = models[-1]['example']
        write_wav16(os.path.join(a.RESULTS_PATH, f"{basename}_input.wav"), x_in.cpu().numpy(), fs)
        write_wav16(os.path.join(a.RESULTS_PATH, f"{basename}_target.wav"), tgt.cpu().numpy(), fs)
        for mid, out in outs.items():
            write_wav16(os.path.join(a.RESULTS_PATH, f"{basename}_prediction_{mid}.wav"), out.cpu().numpy(), fs)
    return outs


if __name__ == "__main__":
    main()
