#!/bin/bash
# low-latency kernel A/B: gru_lat2_kernel (default) against gru_lat_kernel (NTM_LAT_OLD=1), ns per step at small batches
set -u
PY=$(command -v python3)
for B in 1 16 128 256 512 1024; do
    NTM_LAT_OLD=1 $PY tests/quick_bench.py --variant lat --B $B --T 16384 --iters 5 2>&1 | tail -n 1 | sed 's/^/old  /'
    $PY tests/quick_bench.py --variant lat --B $B --T 16384 --iters 5 2>&1 | tail -n 1 | sed 's/^/new  /'
done
