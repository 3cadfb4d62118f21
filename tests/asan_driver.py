"""Child process of tests/test_oracle.py::test_oracle_under_address_and_ub_sanitizers: every entry point of the C
restatement, through the sanitizer build (NTM_ORACLE_LIB), on small and deliberately awkward shapes (T < D, T = 1,
B = 1, D = 1, delays at both ends of the range, ragged TCN lengths).  Any heap overflow / use of uninitialised
stack / UB aborts the process with a report; the parent checks the exit status and the marker line."""
import numpy as np

import oracle
from helpers import oracle_weights

rng = np.random.default_rng(0)
w = oracle_weights("GRU-HS[64]-L[DCPreESR]-DS[ReelToReel_Dataset_MiniPulse100_CHOWTAPE]_BEST")
wd = oracle_weights("DiffDelGRU-HS[64]-L[DCPreESR]-DS[ReelToReel_Dataset_MiniPulse100_CHOWTAPE_WOWFLUTTER]_BEST")
for B, T in ((1, 1), (3, 17), (2, 300)):
    x = rng.uniform(-0.5, 0.5, (B, T)).astype(np.float32)
    y, h = oracle.gru_forward(w, x)
    y2, _ = oracle.gru_forward(w, x, h, threads=2)
    oracle.gru_predict(w, x)
for H in (8, 16, 32):
    k = 1 / np.sqrt(H)
    wh = oracle.Weights(rng.uniform(-k, k, (3 * H, 1)), rng.uniform(-k, k, (3 * H, H)), rng.uniform(-k, k, 3 * H),
                        rng.uniform(-k, k, 3 * H), rng.uniform(-k, k, (1, H)), rng.uniform(-k, k, 1))
    oracle.gru_forward(wh, rng.uniform(-0.5, 0.5, (3, 50)).astype(np.float32))
for B, T, D in ((1, 1, 1), (2, 5, 37), (3, 100, 37), (2, 37, 37), (1, 40, 8900)):
    x = rng.standard_normal((B, T)).astype(np.float32)
    d = rng.uniform(0, D, (B, T)).astype(np.float32)
    d[0, 0], d[-1, -1] = 0.0, D
    if T > 3:
        d[0, 1], d[0, 2] = -0.5, D - 1e-3
    buf = rng.standard_normal((B, D)).astype(np.float32)
    y, nb = oracle.delay_forward(x, d, buf)
    oracle.delay_forward(x, d, buf, warmup=True)
    try:
        d[0, 0] = D + 1
        oracle.delay_forward(x, d, buf)
        raise SystemExit("range violation not reported")
    except AssertionError:
        pass
x = rng.uniform(-0.5, 0.5, (2, 700)).astype(np.float32)
d = rng.uniform(0, 300, (2, 700)).astype(np.float32)
oracle.diffdel_predict(wd, x, d, 300)
oracle.diffdel_predict_f64(wd, x, d, 300)            # fp64 mode (round 6): GRU and delay line on double arrays
oracle.gru_predict_f64(w, x[:, :33], threads=2)
for T, D in ((1, 1), (5, 37), (37, 37)):
    xd = rng.standard_normal((2, T))
    dd = rng.uniform(0, D, (2, T)).astype(np.float32)
    dd[0, 0], dd[-1, -1] = 0.0, D
    oracle.delay_forward_f64(xd, dd, rng.standard_normal((2, D)))
    oracle.delay_forward_f64(xd, dd, rng.standard_normal((2, D)), warmup=True)
t = (x + 0.1 * rng.standard_normal(x.shape)).astype(np.float32)
for skip in (0, 1, 699, 700):
    oracle.esr_sums(x, t, skip)
    oracle.esr_dcpre_sums(x, t, skip)
# TCN parameters in the packed layout of include/ntm.h (block after block W[Cin][K][C], b[C], alpha[C], R[Cin][C]; out_w, out_b)
C, K, dil = 32, 13, (1, 2, 5, 11)
n_par = (1 * K * C + C + C + 1 * C) + 3 * (C * K * C + C + C + C * C) + C + 1
par = rng.uniform(-0.2, 0.2, n_par).astype(np.float32)
for T in (1, 13, 200):
    oracle.tcn_forward(par, 4, C, K, dil, rng.uniform(-1, 1, (2, T)).astype(np.float32))
oracle.tcn_forward(par, 4, C, K, dil, rng.uniform(-1, 1, (3, 64)).astype(np.float32), threads=2)
Hf = 8000.0 * np.sin(np.arange(200) * 0.05)[None, :].repeat(2, 0)
oracle.tape_hmag(Hf, np.zeros((2, 3)), 1.0 / (44100 * 16))
print("ASAN_DRIVER_OK")
