#!/usr/bin/env python3
"""The `--COMPUTE_LOSS` path of the reference's evaluation CLI (code/test-model.py:45-85, 192-247, 296-418)
on the MI355X engine: same UPPER_CASE flags where they apply, batched over segments, shardable over ranks.

    python tools/test_model.py --DATASET_DIR <dir with Test/input_*.wav, target_*.wav> \
        --WEIGHTS "GRU-HS[64]-L[DCPreESR]-DS[...]_BEST" --SEGMENT_LENGTH 441000 --BATCH_SIZE 64 --COMPUTE_LOSS

Differences: plotting / WAV export / noise are out of scope; delay trajectories (DiffDelGRU, `--DEMODULATE`, the
dataset-derived INIT_LEN) come from the `trajectory_<id>_*.npy` side-cars, which the feeder computes and caches on
first use for stereo datasets exactly as DelayAnalyzer does; mono datasets give `--MAX_DELAY` in seconds;
`--WEIGHTS` names one of the exported checkpoints (ntm_amd.weights.available()) or a directory with best.pth.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ntm_amd  # noqa: E402
from ntm_amd import distributed as D  # noqa: E402
from ntm_amd.feeder import SegmentFeeder  # noqa: E402
from ntm_amd.model import ESR_EPS, esr_dcpre_sums, esr_sums  # noqa: E402


def main(argv=None):
    p = argparse.ArgumentParser(description="Compute the loss of a trained tape model over a dataset.")
    p.add_argument('--DATASET_DIR', type=str, required=True)
    p.add_argument('--SUBSET', type=str, default="test")
    p.add_argument('--WEIGHTS', type=str, required=True)
    p.add_argument('--SEGMENT_LENGTH', type=int, default=None)
    p.add_argument('--SYNC', type=float, default=0.0)
    p.add_argument('--BATCH_SIZE', type=int, default=4096, help="segments per launch (the matrix-pipe kernel wants thousands)")
    p.add_argument('--STREAM_CHUNK', type=int, default=8192, help="time chunk of the host->device pipeline; 0 = whole-batch copies")
    p.add_argument('--MAX_DELAY', type=float, default=0.0, help="seconds (DelayAnalyzer.max_delay of the dataset)")
    p.add_argument('--COMPUTE_LOSS', action='store_true', default=False)
    p.add_argument('--DEMODULATE', action='store_true', default=False)
    p.add_argument('--ADD_DELAY', action='store_true', default=False,
                   help="GRU only: apply the measured delay trajectory to the model output (code/test-model.py:236-240,355-364)")
    p.add_argument('--INIT_LEN', type=int, default=None,
                   help="samples cut from the head of every segment before the losses; default: the reference's "
                        "nextpow2(int(max_delay * fs)) (code/test-model.py:323-324), i.e. 2 for a dataset without delay")
    p.add_argument('--KERNEL', type=str, default="auto")
    a = p.parse_args(argv)

    rank, world, local = D.init_from_env()
    torch.cuda.set_device(local)
    feeder = SegmentFeeder(a.DATASET_DIR, subset=a.SUBSET, length=a.SEGMENT_LENGTH, sync=a.SYNC, demodulate=a.DEMODULATE)
    if a.MAX_DELAY <= 0 and feeder.max_delay > 0:          # dataset.delay_analyzer.max_delay (code/test-model.py:323-324)
        a.MAX_DELAY = feeder.max_delay
    sd = None
    if os.path.isdir(a.WEIGHTS):
        sd = torch.load(os.path.join(a.WEIGHTS, "best.pth"), map_location="cpu")
    name = os.path.basename(os.path.normpath(a.WEIGHTS))
    model = ntm_amd.harness.build_model(name, max_delay_seconds=a.MAX_DELAY, fs=feeder.fs, state_dict=sd)
    model.kernel_variant = a.KERNEL
    is_dd = isinstance(model, ntm_amd.DiffDelRNN)
    # code/test-model.py:323-324: INIT_LEN = nextpow2(int(max_delay * fs)) -- for max_delay == 0 that is 2 (the
    # reference's own "# 2**10" comment there is wrong: nextpow2(0) == 2)
    init_len = a.INIT_LEN if a.INIT_LEN is not None else ntm_amd.harness.init_len(a.MAX_DELAY, feeder.fs)
    if not a.COMPUTE_LOSS:
        print(f"{len(feeder)} segments of {feeder.length} samples @ {feeder.fs} Hz; nothing to do without --COMPUTE_LOSS")
        return {}
    per = {"ESR": [], "DCPreESR": [], "MultiSTFT": []}
    mrstft = ntm_amd.MRSTFTLoss()
    with_stft = feeder.length - init_len > 1024               # the largest STFT frame needs > 1024 samples
    def batches():
        if a.DEMODULATE or a.ADD_DELAY or a.STREAM_CHUNK <= 0:
            for xin, tgt, dt, _ in feeder.batches(a.BATCH_SIZE, "cuda", rank, world):
                if is_dd:
                    assert dt is not None, "DiffDelGRU needs trajectory_<id>_*.npy side-cars"
                    out, _ = model.predict(xin, dt * feeder.fs)
                else:
                    out = model.predict(xin)
                yield xin, tgt, out, dt
        else:
            # predict straight from the feeder's pinned files, H2D copies pipelined along time under the launches
            lo, hi = D.shard_range(len(feeder), rank, world)
            for b0 in range(lo, hi, a.BATCH_SIZE):
                out, xin, tgt = feeder.predict_streamed(model, b0, min(hi, b0 + a.BATCH_SIZE), chunk=a.STREAM_CHUNK)
                yield xin, tgt, out, None

    delay = None
    if a.ADD_DELAY and not is_dd:
        assert feeder.max_delay > 0, "--ADD_DELAY needs delay trajectories (stereo dataset or side-cars)"
        delay = ntm_amd.TimeVaryingDelayLine(max_delay=int(1.25 * feeder.max_delay * feeder.fs))      # code/test-model.py:237-238
    for xin, tgt, out, dt in batches():
        if delay is not None:
            out = ntm_amd.harness.apply_delay(delay, dt * feeder.fs, out)
        n = xin.shape[-1] - init_len
        for key, fn in (("ESR", esr_sums), ("DCPreESR", esr_dcpre_sums)):
            s = fn(out, tgt, skip=init_len)
            per[key].append((s[:, 0] / n) / (s[:, 1] / n + ESR_EPS))
        if with_stft:
            per["MultiSTFT"].append(mrstft.per_segment(out, tgt, skip=init_len))
    # every rank issues the SAME collectives whatever its shard holds (a rank with no segments -- more ranks than
    # segments -- reduces empty tensors): the key set depends on the global segment length only
    if not with_stft:
        del per["MultiSTFT"]
    res = {k: D.reduce_loss_sums(torch.cat(v) if v else torch.zeros(0, device="cuda", dtype=torch.float64))
           for k, v in per.items()}
    if rank == 0:
        print("\n===== Stats: =====")
        print(f"Model:      {name}\nDataset:    {os.path.basename(os.path.normpath(a.DATASET_DIR))}\nSubset:     {a.SUBSET}")
        print(f"Segments:   {res['ESR']['segments']}\n")
        for k, v in res.items():
            print(f"{k.ljust(9)}: {v['mean_segment_loss']:.6f}")
        print("\n==================")
    return {k: v["mean_segment_loss"] for k, v in res.items()}


if __name__ == "__main__":
    main()
