"""Segment feeder: the step immediately BEFORE the hot path -- the file layout and slicing of the
reference's `VADataset` (code/dataset.py:129-293, 348-429) restated for batched evaluation.

What is kept: `<data_dir>/<Subset>/input_<id>_*.wav` + `target_<id>_*.wav` pairs (sorted, ids must match),
the '[' ']' glob escaping (:133), one sample rate across the set, files cut into `num_frames // length`
consecutive segments of `length` samples starting at `int(sync * fs)` (:243-256), `(input, target, meta)`
items with the reference's `meta['input_name']` / `['target_name']` strings (:412-419), a segment longer
than a file raising ValueError (:202-204), channel 0 = audio.
What is NOT kept (out of scope): DelayAnalyzer (pulse-train analysis), demodulation,
fractional sub-sampling, shuffling, half/double storage.  Delay trajectories are taken from
`trajectory_<id>_*.npy` side-cars (seconds, one value per sample) when they exist.
The dataset itself (Zenodo 8026272) is not available here, so this module is checked against synthetic
files only (tests/test_feeder.py).
"""
import glob
import os
import re

import numpy as np
import torch
from scipy.io import wavfile


def read_wav(path):
    """-> (float32 [C, N], fs); integer PCM is scaled by 2^(bits-1) like torchaudio.load(normalize=True)."""
    fs, a = wavfile.read(path)
    if a.ndim == 1:
        a = a[:, None]
    if a.dtype == np.int16:
        a = a.astype(np.float32) / 32768.0
    elif a.dtype == np.int32:
        a = a.astype(np.float32) / 2147483648.0
    elif a.dtype == np.uint8:
        a = (a.astype(np.float32) - 128.0) / 128.0
    else:
        a = a.astype(np.float32)
    return np.ascontiguousarray(a.T), int(fs)


def _file_id(path):
    return int(os.path.basename(path).split("_")[1])


class SegmentFeeder:
    def __init__(self, data_dir, subset="train", length=44100, input_only=False, sync=0.0):
        assert os.path.exists(data_dir), "Can't find chosen data_dir"
        self.data_dir, self.subset, self.length, self.input_only, self.sync = data_dir, subset, length, input_only, sync
        search_dir = re.sub(r'([\[\]])', '[\\1]', data_dir)                      # escape [ and ]
        search_string = "**" if subset == "full" else subset.capitalize()
        self.input_files = sorted(glob.glob(os.path.join(search_dir, search_string, "input_*.wav")))
        assert len(self.input_files) > 0, "No input files found!"
        if not input_only:
            self.target_files = sorted(glob.glob(os.path.join(search_dir, search_string, "target_*.wav")))
            assert len(self.target_files) > 0, "No target files found!"
            assert len(self.target_files) == len(self.input_files), "input / target file counts differ"
        else:
            self.target_files = [''] * len(self.input_files)
        traj = {_file_id(p): p for p in glob.glob(os.path.join(search_dir, search_string, "trajectory_*.npy"))}
        self.fs = None
        self.examples = []
        self._audio = []
        for idx, (ifile, tfile) in enumerate(zip(self.input_files, self.target_files)):
            if not input_only and _file_id(ifile) != _file_id(tfile):
                raise RuntimeError(f"Found non-matching file ids: {_file_id(ifile)} != {_file_id(tfile)}! Check dataset.")
            x, fs = read_wav(ifile)
            self.fs = self.fs or fs
            if fs != self.fs:
                raise RuntimeError("Framerate not constant across dataset.")
            if self.length is None:
                self.length = x.shape[-1]
            num_frames = x.shape[-1]
            if num_frames / self.length < 1:
                raise ValueError(f"Sequence length `{self.length}` is longer than file length `{num_frames}`.")
            t = None
            if not input_only:
                t, _ = read_wav(tfile)
                if x.shape[-1] != t.shape[-1]:
                    raise RuntimeError("Found potentially corrupt file!")
            d = np.load(traj[_file_id(ifile)]).astype(np.float32) if _file_id(ifile) in traj else None
            self._audio.append((x, t, d))
            start = int(self.sync * self.fs)
            for n_chunk in range((num_frames - start) // self.length):
                self.examples.append({"idx": idx, "offset": n_chunk * self.length + start})
        self.minutes = self.length * len(self.examples) / self.fs / 60

    def __len__(self):
        return len(self.examples)

    def __getitem__(self, i):
        ex = self.examples[i]
        x, t, d = self._audio[ex["idx"]]
        o, e = ex["offset"], ex["offset"] + self.length
        name = lambda p: "{0}_[{2}:{3}]{1}".format(*os.path.splitext(os.path.basename(p)), o, e)   # noqa: E731
        meta = {"input_name": name(self.input_files[ex["idx"]])}
        inp = torch.from_numpy(x[:, o:e])
        if self.input_only:
            return inp, meta
        meta["target_name"] = name(self.target_files[ex["idx"]])
        if d is not None:
            meta["delay_trajectory"] = torch.from_numpy(d[o:e])
        return inp, torch.from_numpy(t[:, o:e]), meta

    def batches(self, batch_size, device="cuda", rank=0, world=1):
        """Yield (input (B,1,L), target (B,1,L) | None, d_traj_seconds (B,1,L) | None, metas) on `device`
        for this rank's contiguous shard of the segments; host staging buffers are pinned when possible."""
        from .distributed import shard_range
        lo, hi = shard_range(len(self), rank, world)
        pin = torch.cuda.is_available()
        for b0 in range(lo, hi, batch_size):
            items = [self[i] for i in range(b0, min(hi, b0 + batch_size))]
            stack = lambda k: torch.stack([it[k][:1] for it in items])                # noqa: E731  audio = channel 0
            xin = stack(0)
            tgt = None if self.input_only else stack(1)
            metas = [it[-1] for it in items]
            dt = None
            if all("delay_trajectory" in m for m in metas):
                dt = torch.stack([m["delay_trajectory"] for m in metas]).unsqueeze(1)
            out = []
            for a in (xin, tgt, dt):
                if a is None:
                    out.append(None)
                    continue
                a = a.pin_memory() if pin else a
                out.append(a.to(device, non_blocking=True))
            yield out[0], out[1], out[2], metas
