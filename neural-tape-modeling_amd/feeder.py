"""Segment feeder: the step immediately BEFORE the hot path -- the file layout and slicing of the
reference's `VADataset` (code/dataset.py:129-293, 348-429) restated for batched evaluation.

What is kept: `<data_dir>/<Subset>/input_<id>_*.wav` + `target_<id>_*.wav` pairs (sorted, ids must match),
the '[' ']' glob escaping (:133), one sample rate across the set, files cut into `num_frames // length`
consecutive segments of `length` samples starting at `int(sync * fs)` (:243-256), `(input, target, meta)`
items with the reference's `meta['input_name']` / `['target_name']` strings (:412-419), a segment longer
than a file raising ValueError (:202-204), channel 0 = audio.
Delay trajectories come from the `trajectory_<id>_*.npy` side-cars the reference's DelayAnalyzer caches next to
the audio (code/utilities/utilities.py:269-337: a pickled dict with `delay_trajectory` [seconds, one value per
sample], `input_peaks`, `output_peaks`; a plain array of seconds is accepted too); from them the feeder keeps the
analyser's `max_delay` / `min_delay` / `mean_delay` statistics (:191-193, :296-300 -- `max_delay` is what
code/test-model.py:323-324 turns into INIT_LEN and the model's delay-line length), cuts the pulse indices per
segment (code/dataset.py:262-279) and, with `demodulate=True`, demodulates the targets on the device
(`demodulate()` below = DelayAnalyzer.demodulate, :408-465) and drops the mean delay from the segment ends
(code/dataset.py:395-408).
Stereo pairs without a side-car are analysed on first use like the reference does (`find_pulses` / `analyze_delay`
below = code/utilities/utilities.py:343-406, :466-610, pinned by golden g12) and the side-car is written.
`fraction` / `shuffle` select the examples as `create_fractional_patches` does (code/dataset.py:295-341, one device
configuration): the first `int(len * fraction)` segments, or with `shuffle=True` that many drawn WITH replacement by
`np.random.randint` (the reference's global, unseeded generator; `seed` makes the draw repeatable here).
Where the set lives (round 5): on a HIP device whose memory it fits (always, in practice: 288 GB) the decode IS the
host-to-device copy (read_wav_device) and the set stays resident -- batches are device-to-device gathers; otherwise whole
files sit in pinned host memory and batches are large DMA copies (`resident=False` forces this layout).
What is NOT kept (out of scope): half/double storage, un-preloaded operation.
The dataset itself (Zenodo 8026272) is not available here, so this module is checked against synthetic
files only (tests/test_feeder.py).
"""
import glob
import os
import re

import numpy as np
import torch
from scipy.io import wavfile

from . import _lib
from ._lib import ptr


_PCM = {np.dtype(np.int16): (0.0, 32768.0), np.dtype(np.int32): (0.0, 2147483648.0), np.dtype(np.uint8): (128.0, 128.0)}


def read_wav(path):
    """-> (float32 [C, N] numpy, fs); integer PCM is scaled by 2^(bits-1) like torchaudio.load(normalize=True).  The host-side
    decode (no HIP device, or a dataset that does not fit the device: SegmentFeeder(resident=False))."""
    fs, a = wavfile.read(path)
    if a.ndim == 1:
        a = a[:, None]
    if a.dtype in _PCM:
        off, div = _PCM[a.dtype]
        a = (a.astype(np.float32) - np.float32(off)) / np.float32(div) if off else a.astype(np.float32) / np.float32(div)
    else:
        a = a.astype(np.float32, copy=False)
    return np.ascontiguousarray(a.T), int(fs)


_STAGE_BYTES = 64 << 20
_stage = {}          # device index -> two pinned staging buffers + the events that guard their reuse


_TORCH_DT = {np.dtype(np.int16): torch.int16, np.dtype(np.int32): torch.int32, np.dtype(np.uint8): torch.uint8,
             np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64}


def upload_frames(a, dev):
    """Interleaved frames a [N, C] (numpy, possibly memory-mapped; a dtype of _TORCH_DT) -> float32 [C, N] on `dev`: chunks of
    64 MB through two reused pinned staging buffers (the one host pass), H2D as they are, de-interleave + conversion on the
    device.  Asynchronous on the current stream except for the reuse of a staging buffer."""
    N, C = a.shape
    tdt = _TORCH_DT[a.dtype]
    st = _stage.get(dev.index)
    if st is None:
        st = _stage[dev.index] = {"host": [torch.empty(_STAGE_BYTES, dtype=torch.uint8).pin_memory() for _ in range(2)],
                                  "dev": [torch.empty(_STAGE_BYTES, dtype=torch.uint8, device=dev) for _ in range(2)],
                                  "ev": [torch.cuda.Event(), torch.cuda.Event()]}
    frame = a.dtype.itemsize * C
    per = max(1, _STAGE_BYTES // frame)
    out = torch.empty(C, N, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        for k, f0 in enumerate(range(0, N, per)):
            n, b = min(per, N - f0), k & 1
            st["ev"][b].synchronize()                # this staging pair's previous chunk has left it
            hv = st["host"][b][:n * frame]
            hv.numpy().view(a.dtype).reshape(n, C)[...] = a[f0:f0 + n]
            dv = st["dev"][b][:n * frame]
            dv.copy_(hv, non_blocking=True)
            out[:, f0:f0 + n].copy_(dv.view(tdt).view(n, C).t())      # de-interleave + convert on the device
            st["ev"][b].record()
    return out


def read_wav_device(path, device="cuda"):
    """-> (float32 [C, N] tensor ON THE DEVICE, fs), the same values as read_wav.  The decode IS the host-to-device copy:
    the file is memory-mapped, its interleaved frames go through two pinned 64 MB staging buffers (the one host pass: page
    cache -> pinned) and over PCIe as they are, and the de-interleave, the conversion to fp32 and the PCM scaling run on the
    device -- a 450 MB stereo float32 file takes 14 ms (tools/attic/decode_probe.py) where wavfile.read + the numpy transpose +
    pinning took 115 on the host.  With 288 GB of HBM a whole evaluation set lives on the device; SegmentFeeder falls back
    to the pinned-host layout when it does not fit."""
    dev = torch.device(device)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    try:
        fs, a = wavfile.read(path, mmap=True)
    except (ValueError, OSError):                    # formats scipy cannot map (24-bit PCM: it repacks the samples)
        fs, a = wavfile.read(path)
    if a.ndim == 1:
        a = a[:, None]
    N, C = a.shape
    tdt = _TORCH_DT.get(a.dtype)
    if tdt is None or not a.dtype.isnative:
        x, fs = read_wav(path)
        return torch.from_numpy(x).to(dev), fs
    out = upload_frames(a, dev)
    with torch.cuda.device(dev):
        if a.dtype in _PCM:                          # the same fp32 operations as read_wav: exact (powers of two)
            off, div = _PCM[a.dtype]
            if off:
                out.sub_(off)
            out.div_(div)
    return out, int(fs)


def _file_id(path):
    return int(os.path.basename(path).split("_")[1])


def sidecar_path(ifile):
    """`trajectory<rest of the input file's name>.npy` in the input file's directory (code/utilities/utilities.py:273-275)."""
    return os.path.join(os.path.dirname(ifile),
                        "trajectory" + os.path.splitext(os.path.basename(ifile).split("input")[1])[0] + ".npy")


def load_trajectory(path):
    """A `trajectory_<id>_*.npy` side-car -> dict(delay_trajectory [s], input_peaks, output_peaks).
    The reference writes a pickled dict (code/utilities/utilities.py:327-335, read back at :281-282 with
    allow_pickle=True); a plain float array holding only the trajectory is accepted as well."""
    try:
        a = np.load(path)
    except ValueError:                                   # object array: needs pickle, as in the reference
        a = np.load(path, allow_pickle=True)
    if a.dtype == object:
        d = a.item()
        return {"delay_trajectory": np.asarray(d["delay_trajectory"], np.float64),
                "input_peaks": np.asarray(d["input_peaks"]).reshape(-1).astype(np.int64),
                "output_peaks": np.asarray(d["output_peaks"]).reshape(-1).astype(np.int64)}
    return {"delay_trajectory": np.asarray(a, np.float64), "input_peaks": None, "output_peaks": None}


# ---------------------------------------------------------------------------------------------------------
# Pulse-train analysis (host side, numpy/scipy like the reference): what DelayAnalyzer does the first time it sees a
# stereo dataset, code/utilities/utilities.py:343-406 (analyze_delay) and :466-610 (_get_pulse_indices).  Only
# needed when the `trajectory_<id>_*.npy` side-cars are missing; the result is cached in the reference's own
# side-car format.  Pinned by golden g12 (tools/make_goldens_demod.py calls the reference's methods).
# ---------------------------------------------------------------------------------------------------------
def find_pulses(signal, fs, wiggle=True, rel_threshold=0.01, prominence=0.05):
    """Sample indices of the pilot pulses in `signal` (100 pulses per second nominal), with missing pulses
    re-inserted at the median period and, if `wiggle`, pulses whose spacing jumps by >= 5 samples re-timed.
    -> (indices, {"reconstruction_percentage", "wiggle_percentage"})  (code/utilities/utilities.py:466-610)."""
    import scipy.signal
    sig = np.asarray(signal).reshape(-1)
    nominal = fs / 100.0                                   # expected pulse period in samples
    min_dist, max_width = int(0.9 * nominal), (None, nominal / 2)
    gap_factor, wiggle_jump, fallout, max_rebuilt = 1.5, 5, 0.5, 5
    peak = np.max(sig)
    while True:
        # the first pulse with a fixed threshold on the rectified signal, the others with the adaptive one
        first = scipy.signal.find_peaks(np.pad(np.clip(sig, 0, None), (1, 1), 'minimum'), height=peak * 0.01,
                                        distance=min_dist, prominence=0.01, width=max_width)[0][0] - 1
        skip = first + int(nominal / 2)
        rest, _ = scipy.signal.find_peaks(np.pad(sig[skip:], (0, 1), 'minimum'), height=peak * rel_threshold,
                                          distance=min_dist, prominence=prominence, width=max_width)
        idx = np.concatenate(([first], rest + skip))
        period = int(np.median(np.diff(idx)))
        if not np.isclose(period, nominal, atol=nominal * (gap_factor - 1.0)):
            rel_threshold *= fallout; prominence *= fallout        # implausible period: look harder
            continue
        # re-insert missing pulses, one median period after their predecessor
        rebuilt = 0
        gaps = np.where(np.diff(idx) > gap_factor * period)[0]
        while len(gaps) > 0:
            n = gaps[0]
            while idx[n + 1] - idx[n] > gap_factor * period:
                idx = np.concatenate((idx[:n + 1], [idx[n] + period], idx[n + 1:]))
                n += 1
                rebuilt += 1
            gaps = np.where(np.diff(idx) > gap_factor * period)[0]
        rebuilt_pct = rebuilt / len(idx) * 100
        if rebuilt_pct > max_rebuilt:
            rel_threshold *= fallout; prominence *= fallout        # too many guesses: look harder
            continue
        wiggle_pct = 0.0
        if wiggle:
            bad = np.where(np.diff(idx, n=2) >= wiggle_jump)[0]
            idx[bad + 2] = idx[bad + 1] + period
            wiggle_pct = len(bad) / len(idx) * 100
        return idx, {"reconstruction_percentage": rebuilt_pct, "wiggle_percentage": wiggle_pct}


def analyze_delay(input_pilot, output_pilot, fs, wiggle=True, upsampling="cubic"):
    """Delay trajectory [seconds, one value per sample] between two pilot pulse trains
    (code/utilities/utilities.py:343-406): pulse indices of both, their difference at the output pulse times,
    interpolated to every sample (constant outside the pulses).
    -> (input_peaks, output_peaks, delay_trajectory, input_meta, output_meta)."""
    import scipy.interpolate
    xi, xm = find_pulses(input_pilot, fs, wiggle)
    yi, ym = find_pulses(output_pilot, fs, wiggle)
    xi = xi[:len(yi)]
    yi = yi[:len(xi)]
    d = (yi - xi) / fs
    f = scipy.interpolate.interp1d(yi / fs, d, kind=upsampling, fill_value=(d[0], d[-1]), bounds_error=False)
    n = len(np.asarray(input_pilot).reshape(-1))
    return xi, yi, f(np.arange(0, n / fs, 1 / fs)), xm, ym


def write_sidecar(path, input_peaks, output_peaks, delay_trajectory, input_meta, output_meta):
    """The reference's side-car: a pickled dict saved with np.save (code/utilities/utilities.py:327-335)."""
    np.save(path, {"input_peaks": input_peaks, "input_meta": input_meta, "output_peaks": output_peaks,
                   "output_meta": output_meta, "delay_trajectory": delay_trajectory})


def segment_peaks(input_peaks, output_peaks, offset, end, length):
    """Pulse indices of one segment, relative to its start: code/dataset.py:262-279 (including its use of the
    last pulse INDEX VALUE as a slice bound when the segment runs past the last pulse)."""
    if input_peaks is None or offset > int(np.max(input_peaks)):
        return None, None
    first_idx = np.where(input_peaks >= offset)[0][0]
    last_idx = np.where(input_peaks <= end)[0][-1] if end <= int(np.max(input_peaks)) else input_peaks[-1]
    pin = input_peaks[first_idx:last_idx] - offset
    pout = output_peaks[first_idx:last_idx] - offset
    return pin, pout[pout <= length]


@torch.no_grad()
def demodulate(output, x_idx_pulse, y_idx_pulse):
    """DelayAnalyzer.demodulate (code/utilities/utilities.py:408-465) on the device: `output` (C,N) tensor (the
    reference passes the 2-channel target: audio + pilot pulses), pulse indices as integer arrays.
    Returns a new (C,N) fp32 tensor.  Needs the HIP library; there is no CPU fallback."""
    x_idx = np.asarray(x_idx_pulse).reshape(-1).astype(np.int64)
    y_idx = np.asarray(y_idx_pulse).reshape(-1).astype(np.int64)
    assert output.dim() == 2, "output should be (channels, samples)"
    assert len(x_idx) >= 2 and len(y_idx) >= 2, "need at least two pulses"
    if not output.is_cuda:
        raise _lib.NtmError("demodulate: tensor must live on the GPU (no CPU fallback)")
    x = output.to(torch.float32).contiguous()
    C, N = x.shape
    period = int(np.mean(np.diff(x_idx)))
    shift = int(y_idx[0] - x_idx[0])
    yi = torch.from_numpy(y_idx).to(x.device)
    out = torch.empty_like(x)
    scratch = torch.empty(N, device=x.device, dtype=torch.float64)
    rc = _lib.lib().ntm_demodulate(ptr(x), ptr(out), C, N, ptr(yi), len(y_idx), period, shift, ptr(scratch),
                                   _lib.current_stream())
    _lib.check(rc, "ntm_demodulate")
    return out


def decoded_bytes(path, with_trajectory=False):
    """Bytes `path` occupies once decoded to float32 [C, N] (+ a float32 trajectory of N samples), from its WAV header (the
    file is memory-mapped, nothing is read); a format scipy cannot map (24-bit PCM) is bounded by 4 bytes per byte on disk."""
    try:
        _, a = wavfile.read(path, mmap=True)
        frames, ch = (a.shape[0], 1) if a.ndim == 1 else a.shape
        del a
    except (ValueError, OSError):
        frames, ch = os.path.getsize(path), 4
    return 4 * frames * ch + (4 * frames if with_trajectory else 0)


class SegmentFeeder:
    def __init__(self, data_dir, subset="train", length=44100, input_only=False, sync=0.0, demodulate=False,
                 analyze=True, write_sidecars=True, fraction=1.0, shuffle=False, seed=None, resident=None):
        assert os.path.exists(data_dir), "Can't find chosen data_dir"
        assert not (input_only and demodulate), "Can't demodulate without inputs"       # code/dataset.py:68
        self.data_dir, self.subset, self.length, self.input_only, self.sync = data_dir, subset, length, input_only, sync
        self.demodulate = demodulate
        self.mean_delay, self.max_delay, self.min_delay = 0.0, 0.0, 1e6              # DelayAnalyzer, utilities.py:191-193
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
        self.fs = None
        self.examples = []
        self._audio = []
        # `resident`: the decoded set lives ON THE DEVICE (read_wav_device: the decode is the H2D copy; batches are then
        # device-to-device gathers and predict_streamed one launch).  None = yes when a HIP device is present and the DECODED
        # set -- frames x channels x 4 bytes per file from the WAV headers, + frames x 4 for a trajectory side-car: int16 PCM
        # decodes to 2 x its file size, uint8 to 4 x -- fits half of the device's free memory; False = the pinned-host layout
        # with H2D copies per batch (what a set larger than the device needs).  An automatic choice that still runs out of
        # device memory (another process took it meanwhile) frees what it uploaded and falls back to the pinned-host layout.
        auto = resident is None
        if auto:
            resident = False
            if torch.cuda.is_available():
                need = sum(decoded_bytes(f, with_trajectory=os.path.exists(sidecar_path(f))) for f in self.input_files)
                need += sum(decoded_bytes(t) for t in self.target_files if t)
                resident = need < 0.5 * torch.cuda.mem_get_info()[0]
        self.resident = bool(resident)
        try:
            self._decode_all(input_only, analyze, write_sidecars)
        except torch.cuda.OutOfMemoryError:
            if not (auto and self.resident):
                raise
            self.examples, self._audio, self.fs = [], [], None
            self.mean_delay, self.max_delay, self.min_delay = 0.0, 0.0, 1e6
            _stage.clear()
            torch.cuda.empty_cache()
            self.resident = False
            self._decode_all(input_only, analyze, write_sidecars)
        self._finish(demodulate, fraction, shuffle, seed)

    def _decode_all(self, input_only, analyze, write_sidecars):
        """Decode every file pair (+ side-car) into the layout self.resident names; fills _audio, examples, the delay statistics."""
        for idx, (ifile, tfile) in enumerate(zip(self.input_files, self.target_files)):
            if not input_only and _file_id(ifile) != _file_id(tfile):
                raise RuntimeError(f"Found non-matching file ids: {_file_id(ifile)} != {_file_id(tfile)}! Check dataset.")
            # the side-car (the reference's pickled fp64 trajectory: ~90 ms to unpickle per 450 MB) loads on a worker thread
            # while the audio is decoded (file reads, large copies and the device work all release the GIL)
            sidecar = sidecar_path(ifile)
            side = {}
            loader = None
            if os.path.exists(sidecar):
                import threading

                def _load(path=sidecar, out=side):
                    try:
                        out["d"] = load_trajectory(path)
                    except Exception as e:           # re-raised on the constructing thread below
                        out["error"] = e
                loader = threading.Thread(target=_load, daemon=True)
                loader.start()
            x, fs = read_wav_device(ifile) if self.resident else read_wav(ifile)
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
                t, _ = read_wav_device(tfile) if self.resident else read_wav(tfile)
                if x.shape[-1] != t.shape[-1]:
                    raise RuntimeError("Found potentially corrupt file!")
            # the side-car sits next to its input file under the input's own name (code/utilities/utilities.py:273-275):
            # with subset "full" equal ids in Train/ Val/ Test/ must not pick up each other's trajectories
            d = None
            if loader is not None:
                loader.join()
                if "error" in side:
                    raise side["error"]
                d = side["d"]
            if d is None and analyze and t is not None and x.shape[0] > 1 and t.shape[0] > 1:
                # stereo pair without a side-car: analyse the pilot channels as DelayAnalyzer does on first use
                # (code/utilities/utilities.py:306-335) and cache the result next to the audio in its format
                pilot = lambda a: (a[1].cpu().numpy() if isinstance(a, torch.Tensor) else a[1]).astype(np.float64)   # noqa: E731
                xi, yi, T_delay, xm, ym = analyze_delay(pilot(x), pilot(t), fs)
                if write_sidecars:
                    try:
                        write_sidecar(sidecar, xi, yi, T_delay, xm, ym)
                    except OSError:
                        pass                                   # read-only dataset: keep the analysis in memory
                d = {"delay_trajectory": np.asarray(T_delay, np.float64), "input_peaks": xi.astype(np.int64),
                     "output_peaks": yi.astype(np.int64)}
            if d is not None:                                                          # utilities.py:296-300
                tr = torch.from_numpy(np.ascontiguousarray(d["delay_trajectory"]))
                self.mean_delay += float(np.mean(d["delay_trajectory"]))              # (numpy's pairwise sum, as the reference)
                self.max_delay = max(self.max_delay, float(tr.max()))                 # max / min / the fp32 copy: exact in any
                self.min_delay = min(self.min_delay, float(tr.min()))                 # order, so on all host cores (torch)
            # resident: everything is on the device already.  Otherwise whole files live in PINNED host memory when a HIP
            # device is present: a batch then goes to the device as a few large DMA copies straight from here (no per-batch
            # staging copy on the host)
            if d is not None:
                if self.resident:    # fp64 -> fp32 on the device, through the decode's own staging buffers (exactly as rounded on the host)
                    d["traj_f32"] = upload_frames(np.ascontiguousarray(d["delay_trajectory"]).reshape(-1, 1), x.device)
                else:
                    d["traj_f32"] = self._host(tr.to(torch.float32)[None, :].contiguous().numpy())
            self._audio.append((self._host(x), None if t is None else self._host(t), d))
            start = int(self.sync * self.fs)
            for n_chunk in range((num_frames - start) // self.length):
                self.examples.append({"idx": idx, "offset": n_chunk * self.length + start})

    def _finish(self, demodulate, fraction, shuffle, seed):
        n_traj = sum(1 for a in self._audio if a[2] is not None)
        self.mean_delay = self.mean_delay / n_traj if n_traj else 0.0               # utilities.py:341
        assert not (demodulate and n_traj != len(self._audio)), "Can't demodulate without trajectory side-cars!"
        # code/dataset.py:295-341 (create_fractional_patches, a single class of examples)
        self.fraction, self.shuffle = fraction, shuffle
        n_use = int(len(self.examples) * fraction)
        if n_use <= 0:
            raise ValueError(f"Fraction `{fraction}` set too low. No examples selected.")
        if shuffle:
            rs = np.random if seed is None else np.random.RandomState(seed)
            pick = rs.randint(0, high=len(self.examples), size=n_use)
        else:
            pick = np.arange(n_use)
        self.examples = [self.examples[i] for i in pick]
        self.minutes = self.length * len(self.examples) / self.fs / 60
        if self.resident:
            torch.cuda.synchronize()         # the decode's copies and kernels are done: any stream may read the set now

    @staticmethod
    def _host(a):
        if isinstance(a, torch.Tensor):              # resident: decoded on the device
            return a
        t = torch.from_numpy(a)
        if torch.cuda.is_available():
            try:
                return t.pin_memory()
            except RuntimeError:          # not enough pinnable memory: pageable copies still work
                return t
        return t

    def __len__(self):
        return len(self.examples)

    def __getitem__(self, i):
        ex = self.examples[i]
        x, t, d = self._audio[ex["idx"]]
        o, e = ex["offset"], ex["offset"] + self.length
        name = lambda p: "{0}_[{2}:{3}]{1}".format(*os.path.splitext(os.path.basename(p)), o, e)   # noqa: E731
        meta = {"input_name": name(self.input_files[ex["idx"]])}
        inp = x[:, o:e].cpu()            # items are HOST tensors like the reference's (a no-op for the pinned-host layout)
        if self.input_only:
            return inp, meta
        meta["target_name"] = name(self.target_files[ex["idx"]])
        tgt = t[:, o:e].cpu()
        if d is not None:
            T_delay = torch.from_numpy(d["delay_trajectory"][o:e].astype(np.float32))
            pin, pout = segment_peaks(d["input_peaks"], d["output_peaks"], o, e, self.length)
            if self.demodulate:                                                        # code/dataset.py:395-408
                assert pin is not None, "Can't demodulate without pulse indices!"
                tgt = demodulate(tgt.cuda(), pin, pout).cpu()
                cut = int(self.mean_delay * self.fs)
                inp, tgt, T_delay = inp[:, :-cut], tgt[:, :-cut], T_delay[:-cut]
                pin = pin[pin <= inp.shape[-1]]
                pout = pin                                                             # as upstream ("assume everything works")
            meta["delay_trajectory"] = T_delay
            meta["input_peaks"], meta["output_peaks"] = pin, pout
        return inp, tgt, meta

    def runs(self, b0, b1):
        """Segments b0..b1-1 as runs of consecutive segments of one file: [(first row, n segments, file idx, offset)]."""
        out, k, L = [], b0, self.length
        while k < b1:
            ex = self.examples[k]
            r = k + 1
            while r < b1 and self.examples[r]["idx"] == ex["idx"] and self.examples[r]["offset"] == ex["offset"] + (r - k) * L:
                r += 1
            out.append((k - b0, r - k, ex["idx"], ex["offset"]))
            k = r
        return out

    @torch.no_grad()
    def predict_streamed(self, model, b0, b1, chunk=8192, device="cuda", out_host=None):
        """GRU (or DiffDelGRU: the delay trajectory travels too) predict over segments b0..b1-1 straight from pinned
        host memory, pipelined along TIME: the batch goes to
        the device in chunks of `chunk` samples x all segments (pitched DMA copies, ntm_copy2d_async, on a side stream)
        and the kernel for chunk c runs while chunk c+1 is in flight -- every launch still sees the full batch (the
        matrix-pipe kernel needs thousands of streams per launch; splitting the batch by segments instead would starve
        it).  State is carried between the launches, so the result is bit-identical to one launch.
        `out_host`: optional pinned (B,1,L) fp32 host tensor; each finished output chunk is copied back on a
        second side stream while the next chunk computes (host -> device -> host, all three overlapped; the caller
        synchronises before reading it).
        -> (output (B,1,L), input (B,1,L), target (B,1,L) | None), all on `device`."""
        assert not self.demodulate, "demodulated targets take the per-item path (batches())"
        B, L = b1 - b0, self.length
        lib = _lib.lib()
        from .model import DiffDelRNN
        is_dd = isinstance(model, DiffDelRNN)
        if self.resident:
            return self._predict_resident(model, b0, b1, is_dd, device, out_host)
        x = torch.empty(B, 1, L, device=device, dtype=torch.float32)
        t = None if self.input_only else torch.empty(B, 1, L, device=device, dtype=torch.float32)
        y = torch.empty(B, 1, L, device=device, dtype=torch.float32)
        dtr = None
        if is_dd:        # the delay trajectory [seconds] travels with the audio, chunk by chunk
            assert all(self._audio[self.examples[i]["idx"]][2] is not None for i in range(b0, b1)), \
                "DiffDelGRU needs delay trajectories (stereo dataset or side-cars)"
            dtr = torch.empty(B, 1, L, device=device, dtype=torch.float32)
        runs = self.runs(b0, b1)
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream(device=device)
        side.wait_stream(cur)                                   # the fresh buffers belong to the compute stream

        def send(c0, c1):
            for k0, n, idx, off in runs:
                for which, dst in ((0, x), (1, t), (2, dtr)):
                    if dst is None:
                        continue
                    src = self._audio[idx][which]
                    if which == 2:
                        src = src["traj_f32"]
                    rc = lib.ntm_copy2d_async(dst[k0, 0, c0:].data_ptr(), 4 * L, src[0, off + c0:].data_ptr(), 4 * L,
                                              4 * (c1 - c0), n, 0, side.cuda_stream)
                    _lib.check(rc, "ntm_copy2d_async")
            ev = torch.cuda.Event()
            ev.record(side)
            return ev

        if is_dd:
            model.initialize_hidden(1, model.max_delay)
            model.warm_start()
            if B != 1:
                model.hidden = model.hidden.expand(1, B, model.hidden_size).contiguous()
                model.diffdel.buffer = model.diffdel.buffer.expand(B, 1, -1).contiguous()
            deferred, model.diffdel.defer_check = model.diffdel.defer_check, True   # no host sync per chunk
        else:
            model.initialize_hidden()
            model.warm_start()
            if B != 1:
                model.hidden = model.hidden.expand(1, B, model.hidden_size).contiguous()
        try:     # the deferred-check mode of the delay line is restored whatever the chunk loop does
            bounds = [(c0, min(L, c0 + chunk)) for c0 in range(0, L, chunk)]
            ev = send(*bounds[0])
            back = torch.cuda.Stream(device=device) if out_host is not None else None
            if out_host is not None:
                assert tuple(out_host.shape) == (B, 1, L) and out_host.dtype == torch.float32 and out_host.is_contiguous()
            for i, (c0, c1) in enumerate(bounds):
                nxt = send(*bounds[i + 1]) if i + 1 < len(bounds) else None
                cur.wait_event(ev)
                if is_dd:    # GRU + delay line on the chunk (state of both carried); trajectory seconds -> samples
                    y[:, :, c0:c1] = model.forward(x[:, :, c0:c1], dtr[:, :, c0:c1] * float(self.fs))[0]
                else:
                    model.forward_into(x[:, 0, c0:c1], y[:, 0, c0:c1])
                if back is not None:
                    done = torch.cuda.Event()
                    done.record(cur)
                    back.wait_event(done)
                    rc = lib.ntm_copy2d_async(out_host[0, 0, c0:].data_ptr(), 4 * L, y[0, 0, c0:].data_ptr(), 4 * L,
                                              4 * (c1 - c0), B, 1, back.cuda_stream)
                    _lib.check(rc, "ntm_copy2d_async")
                ev = nxt
            if back is not None:
                cur.wait_stream(back)                               # a sync of the caller's stream covers the copies back
        finally:
            if is_dd:
                model.diffdel.defer_check = deferred
        if is_dd and not deferred:
            model.diffdel.raise_if_violated()                  # the assert of code/model.py:284, once for all chunks
        for a in (x, t, dtr):
            if a is not None:
                a.record_stream(side)
        return y, x, t

    @torch.no_grad()
    def _predict_resident(self, model, b0, b1, is_dd, device, out_host):
        """predict_streamed for a device-resident set: the segments are gathered device-to-device (one copy per run of
        consecutive segments) and the whole sequence is ONE launch -- there is no PCIe transfer left to hide."""
        B, L = b1 - b0, self.length

        def gather(which):
            dst = torch.empty(B, 1, L, device=device, dtype=torch.float32)
            for k0, n, idx, off in self.runs(b0, b1):
                src = self._audio[idx][which]
                if which == 2:
                    src = src["traj_f32"]
                dst[k0:k0 + n, 0].copy_(src[0, off:off + n * L].view(n, L))
            return dst
        x = gather(0)
        t = None if self.input_only else gather(1)
        if is_dd:
            assert all(self._audio[self.examples[i]["idx"]][2] is not None for i in range(b0, b1)), \
                "DiffDelGRU needs delay trajectories (stereo dataset or side-cars)"
            y = model.predict(x, gather(2) * float(self.fs))[0]
        else:
            y = model.predict(x)
        if out_host is not None:
            assert tuple(out_host.shape) == (B, 1, L) and out_host.dtype == torch.float32 and out_host.is_contiguous()
            out_host.copy_(y, non_blocking=True)                # the caller synchronises before reading it
        return y, x, t

    def batches(self, batch_size, device="cuda", rank=0, world=1, prefetch=True, timing=None):
        """Yield (input (B,1,L), target (B,1,L) | None, d_traj_seconds (B,1,L) | None, metas) on `device`
        for this rank's contiguous shard of the segments.  On a HIP device consecutive segments of a file are
        one contiguous run of pinned host memory, so a batch is a handful of large async DMA copies issued on a
        side stream while the caller still computes on the previous batch (`prefetch`); the consumer's stream only
        waits for the copy event of the batch it is handed.  (Demodulated targets take the per-item path.)
        `timing`: optional list; every staged batch appends (start event, end event, bytes) recorded on the copy stream
        around its host-to-device copies (the evaluation CLI's stage times)."""
        from .distributed import shard_range
        lo, hi = shard_range(len(self), rank, world)
        on_gpu = torch.cuda.is_available() and torch.device(device).type == "cuda"
        copy_stream = torch.cuda.Stream(device=device) if (on_gpu and prefetch) else None
        L = self.length

        def fill(dst, which, b0, b1):
            """dst (n,1,L) device tensor <- channel 0 of segments b0..b1-1, one copy per run of consecutive segments."""
            k = b0
            while k < b1:
                ex = self.examples[k]
                r = k + 1
                while r < b1 and self.examples[r]["idx"] == ex["idx"] and \
                        self.examples[r]["offset"] == ex["offset"] + (r - k) * L:
                    r += 1
                src = self._audio[ex["idx"]][which]
                if which == 2:
                    src = src["traj_f32"]
                src = src[0, ex["offset"]:ex["offset"] + (r - k) * L]
                dst[k - b0:r - b0, 0].copy_(src.view(r - k, L), non_blocking=on_gpu)
                k = r

        def meta_of(i):
            ex = self.examples[i]
            o, e = ex["offset"], ex["offset"] + L
            name = lambda p: "{0}_[{2}:{3}]{1}".format(*os.path.splitext(os.path.basename(p)), o, e)   # noqa: E731
            m = {"input_name": name(self.input_files[ex["idx"]])}
            if not self.input_only:
                m["target_name"] = name(self.target_files[ex["idx"]])
            return m

        def stage_fast(b0):
            b1 = min(hi, b0 + batch_size)
            has_d = all(self._audio[self.examples[i]["idx"]][2] is not None for i in range(b0, b1))
            out = [torch.empty(b1 - b0, 1, L, device=device, dtype=torch.float32),
                   None if self.input_only else torch.empty(b1 - b0, 1, L, device=device, dtype=torch.float32),
                   torch.empty(b1 - b0, 1, L, device=device, dtype=torch.float32) if has_d else None]
            for which, dst in enumerate(out):
                if dst is not None:
                    fill(dst, which, b0, b1)
            return out, [meta_of(i) for i in range(b0, b1)], None

        def stage_items(b0):
            items = [self[i] for i in range(b0, min(hi, b0 + batch_size))]
            stack = lambda k: torch.stack([it[k][:1] for it in items])                # noqa: E731  audio = channel 0
            metas = [{k: v for k, v in it[-1].items() if not k.endswith("_peaks")} for it in items]
            dt = None
            if all("delay_trajectory" in m for m in metas):
                dt = torch.stack([m["delay_trajectory"] for m in metas]).unsqueeze(1)
            host = [stack(0), None if self.input_only else stack(1), dt]
            host = [a.pin_memory() if (a is not None and on_gpu) else a for a in host]
            return [None if a is None else a.to(device, non_blocking=on_gpu) for a in host], metas, host

        def stage(b0):
            fn = stage_items if self.demodulate else stage_fast
            if copy_stream is None:
                out, metas, keep = fn(b0)
                return out, metas, None, keep
            with torch.cuda.stream(copy_stream):
                if timing is not None:
                    t0 = torch.cuda.Event(enable_timing=True)
                    t0.record(copy_stream)
                out, metas, keep = fn(b0)
                ev = torch.cuda.Event(enable_timing=timing is not None)
                ev.record(copy_stream)
                if timing is not None:
                    timing.append((t0, ev, sum(a.numel() * 4 for a in out if a is not None)))
            return out, metas, ev, keep          # `keep` holds pinned staging buffers alive until the batch is consumed

        starts = list(range(lo, hi, batch_size))
        nxt = stage(starts[0]) if starts else None
        for k in range(len(starts)):
            out, metas, ev, keep = nxt
            nxt = stage(starts[k + 1]) if k + 1 < len(starts) else None
            if ev is not None:
                cur = torch.cuda.current_stream()
                cur.wait_event(ev)
                for a in out:
                    if a is not None:
                        a.record_stream(cur)
            yield out[0], out[1], out[2], metas
            del keep
