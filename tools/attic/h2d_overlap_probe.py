"""End-to-end rate from HOST memory: a synthetic dataset in the VADataset layout (64 WAV pairs, 4096 segments of
65 536 samples), SegmentFeeder -> pinned batches -> GRU-HS[64] predict -> ESR sums, with and without the
side-stream prefetch of the next batch.  (The headline `value` of bench.py starts with inputs resident in HBM.)"""
import os, sys, time, tempfile
import numpy as np, torch
from scipy.io import wavfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ntm_amd
from ntm_amd.feeder import SegmentFeeder
from ntm_amd.model import esr_sums

L, per_file, n_files = 65536, 64, int(sys.argv[1]) if len(sys.argv) > 1 else 64
root = tempfile.mkdtemp(prefix="ntm_e2e_", dir="/tmp")
d = os.path.join(root, "Set", "Test"); os.makedirs(d)
rng = np.random.default_rng(0)
t0 = time.time()
for i in range(n_files):
    x = (rng.uniform(-0.5, 0.5, L * per_file) * 32767).astype(np.int16)
    wavfile.write(os.path.join(d, f"input_{i}_.wav"), 44100, x)
    wavfile.write(os.path.join(d, f"target_{i}_.wav"), 44100, np.roll(x, 3))
f = SegmentFeeder(os.path.join(root, "Set"), subset="test", length=L)
print(f"dataset: {len(f)} segments x {L} samples, built + loaded in {time.time() - t0:.1f} s")
m = ntm_amd.harness.build_model(ntm_amd.weights.W_GRU)
for prefetch in (False, True, False, True):
    torch.cuda.synchronize(); t0 = time.time()
    tot = 0.0
    for xin, tgt, _, _ in f.batches(512, "cuda", prefetch=prefetch):
        y = m.predict(xin)
        tot += float(esr_sums(y, tgt, 1024)[:, 0].sum())
    torch.cuda.synchronize(); dt = time.time() - t0
    print(f"prefetch={prefetch}: {dt*1e3:.0f} ms for {len(f)*L/1e6:.0f} M samples = {len(f)*L/dt/1e9:.2f} Gsamples/s end to end from host memory")
for rep in range(3):
    for chunk in (65536, 8192, 4096):
        torch.cuda.synchronize(); t0 = time.time()
        y, xin, tgt = f.predict_streamed(m, 0, len(f), chunk=chunk)
        tot = float(esr_sums(y, tgt, 1024)[:, 0].sum())
        torch.cuda.synchronize(); dt = time.time() - t0
        print(f"streamed, one batch of {len(f)}, time chunks of {chunk}: {dt*1e3:.0f} ms = {len(f)*L/dt/1e9:.2f} Gsamples/s end to end from host memory")
host = torch.empty(len(f), 1, L).pin_memory()
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    y, xin, tgt = f.predict_streamed(m, 0, len(f), chunk=8192, out_host=host)
    torch.cuda.synchronize(); dt = time.time() - t0
    print(f"streamed host -> device -> host (output copied back per chunk): {dt*1e3:.0f} ms = {len(f)*L/dt/1e9:.2f} Gsamples/s")
import shutil; shutil.rmtree(root)
