#!/bin/bash
# round 3, fourth GPU pass: evidence for the final fused DiffDelRNN step -- kernel stats, bench lines (auto = fused, and
# two_pass), PMC traffic of the fused kernel and of the headline kernel on this round's binary
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
PY=$(command -v python3)
bash tools/profile_all.sh r03_d diffdel
timeout 600 $PY bench.py --workload diffdel --steps 10 --warmup 2 > "$OUT/r03_d_bench_diffdel.json" 2> "$OUT/r03_d_bench_diffdel.err"; echo "bench diffdel exit $?"
timeout 600 $PY tools/diffdel_ab_probe.py > "$OUT/r03_d_diffdel_ab_probe.txt" 2>&1; cat "$OUT/r03_d_diffdel_ab_probe.txt"
ALG_GRU=$((4096*65536*8))
ALG_DD=$((4096*65536*16))
bash tools/pmc_traffic.sh r03_d mfma2 "gru_mfma2_kernel<true, false, 0, 0, 16, false>" $ALG_GRU bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --other off
bash tools/pmc_traffic.sh r03_d diffdel_fused "gru_mfma2_kernel<true, false, 0, 0, 16, true>" $ALG_DD bench.py --workload diffdel --steps 2 --warmup 1 --no-cpu-baseline
