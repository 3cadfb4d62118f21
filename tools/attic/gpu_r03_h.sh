#!/bin/bash
# round 3, evidence for the final binary (whole-tile housekeeping forms, fused DiffDelGRU step): default bench line, kernel
# stats of the default line and of the DiffDelGRU workload, A/B probes, PMC traffic of the two kernels
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
PY=$(command -v python3)
( time timeout 600 $PY bench.py ) > "$OUT/r03_h_bench_default.json" 2> "$OUT/r03_h_bench_default.err"; echo "bench exit $?"; tail -n 4 "$OUT/r03_h_bench_default.err"
timeout 600 $PY bench.py --workload diffdel --steps 10 --warmup 2 > "$OUT/r03_h_bench_diffdel.json" 2> "$OUT/r03_h_bench_diffdel.err"; echo "bench diffdel exit $?"
timeout 600 $PY tools/diffdel_ab_probe.py > "$OUT/r03_h_diffdel_ab_probe.txt" 2>&1; cat "$OUT/r03_h_diffdel_ab_probe.txt"
for i in 1 2 3; do $PY tests/quick_bench.py --variant mfma2 --B 4096 --T 65536 --iters 6 | tail -1; done | tee "$OUT/r03_h_quick_bench_mfma2.txt"
bash tools/profile_all.sh r03_h gru diffdel
ALG_GRU=$((4096*65536*8))
ALG_DD=$((4096*65536*16))
bash tools/pmc_traffic.sh r03_h mfma2 "gru_mfma2_kernel<true, false, 0, 0, 16, false>" $ALG_GRU bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --other off --traffic off | tail -1 | cut -c1-200
bash tools/pmc_traffic.sh r03_h diffdel_fused "gru_mfma2_kernel<true, false, 0, 0, 16, true>" $ALG_DD bench.py --workload diffdel --steps 2 --warmup 1 --no-cpu-baseline | tail -1 | cut -c1-200
