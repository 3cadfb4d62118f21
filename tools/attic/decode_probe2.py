#!/usr/bin/env python3
"""Decode stage of the evaluation command, piece by piece, on the dataset bench.py's other_workloads.cli builds (n_seg x 441000
stereo float32 WAV pair + pickled trajectory side-car): SegmentFeeder construction as a whole, then its parts in isolation."""
import json, os, sys, tempfile, time, shutil, warnings
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, ntm_amd
from ntm_amd import feeder as F
from scipy.io import wavfile
n_seg = int(sys.argv[1]) if len(sys.argv) > 1 else 128
tmp = tempfile.mkdtemp(prefix="ntm_dec2_", dir="/tmp")
ds, host = bench.cli_dataset(tmp, n_seg, 441000, torch.device("cuda", 0))
out = {}
def T(name, fn, reps=3):
    best = None
    for _ in range(reps):
        t0 = time.perf_counter(); r = fn(); dt = 1e3 * (time.perf_counter() - t0)
        best = dt if best is None else min(best, dt)
    out[name + "_ms"] = best
    return r
T("SegmentFeeder_resident", lambda: F.SegmentFeeder(ds, subset="Test", length=441000, shuffle=False), reps=5)
T("SegmentFeeder_pinned_host", lambda: F.SegmentFeeder(ds, subset="Test", length=441000, shuffle=False, resident=False))
ifile = os.path.join(ds, "Test", "input_1_.wav")
side = os.path.join(ds, "Test", "trajectory_1_.npy")
T("read_wav_device", lambda: (F.read_wav_device(ifile), torch.cuda.synchronize()))
T("read_wav_host", lambda: F.read_wav(ifile))
fs, mm = wavfile.read(ifile, mmap=True)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    tm = torch.from_numpy(mm)
dst = torch.empty(2, mm.shape[0])
T("torch_copy_from_mmap_transposed", lambda: dst.copy_(tm.t()))
T("pinned_alloc_450MB", lambda: torch.empty(2, mm.shape[0], pin_memory=True))
raw = T("wavfile_read", lambda: wavfile.read(ifile)[1])
T("numpy_transpose", lambda: np.ascontiguousarray(raw.T))
tr = torch.from_numpy(raw)
T("torch_transpose_from_ram", lambda: dst.copy_(tr.t()))
d = T("load_trajectory", lambda: F.load_trajectory(side))
tj = d["delay_trajectory"]
T("np_mean", lambda: np.mean(tj)); T("np_max", lambda: np.max(tj))
tt = torch.from_numpy(tj)
T("torch_max", lambda: tt.max()); T("torch_to_f32", lambda: tt.to(torch.float32))
T("np_to_f32", lambda: np.ascontiguousarray(tj, np.float32))
T("traj_pin_and_h2d", lambda: (tt.to(torch.float32)[None, :].contiguous().pin_memory().to("cuda", non_blocking=True), torch.cuda.synchronize()))
T("traj_upload_frames", lambda: (F.upload_frames(tj.reshape(-1, 1), torch.device("cuda", 0)), torch.cuda.synchronize()))
print(json.dumps(out))
shutil.rmtree(tmp, ignore_errors=True)
