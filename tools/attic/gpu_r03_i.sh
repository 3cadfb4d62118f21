#!/bin/bash
# round 3, evidence for the FINAL binary (whole-tile housekeeping forms, fused DiffDelGRU step, loss leg fused into the
# recurrent launch): whole GPU suite, smoke, default bench line (+ its --esr pass A/B), kernel stats, probes, PMC traffic
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
PY=$(command -v python3)
( time timeout 3000 $PY -m pytest tests -q -m gpu ) > "$OUT/r03_i_tests.log" 2>&1; echo "tests exit $?"; tail -n 5 "$OUT/r03_i_tests.log"
timeout 600 $PY __graft_entry__.py smoke > "$OUT/r03_i_smoke.log" 2>&1; echo "smoke exit $?"
( time timeout 600 $PY bench.py ) > "$OUT/r03_i_bench_default.json" 2> "$OUT/r03_i_bench_default.err"; echo "bench exit $?"; tail -n 4 "$OUT/r03_i_bench_default.err"
for e in pass fused pass fused; do timeout 600 $PY bench.py --no-cpu-baseline --no-extra --other off --traffic off --esr $e; done > "$OUT/r03_i_bench_esr_ab.jsonl" 2>/dev/null
timeout 600 $PY bench.py --workload diffdel --steps 10 --warmup 2 > "$OUT/r03_i_bench_diffdel.json" 2> "$OUT/r03_i_bench_diffdel.err"; echo "bench diffdel exit $?"
timeout 600 $PY tools/diffdel_ab_probe.py > "$OUT/r03_i_diffdel_ab_probe.txt" 2>&1
timeout 600 $PY tools/esr_fused_probe.py > "$OUT/r03_i_esr_fused_probe.txt" 2>&1; cat "$OUT/r03_i_esr_fused_probe.txt"
bash tools/profile_all.sh r03_i gru diffdel
ALG_ESR=$((4096*65536*12))
ALG_DD=$((4096*65536*16))
bash tools/pmc_traffic.sh r03_i mfma2_esr "gru_mfma2_kernel<true, false, 0, 0, 16, false, true>" $ALG_ESR bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --other off --traffic off | tail -1 | cut -c1-150
bash tools/pmc_traffic.sh r03_i diffdel_fused "gru_mfma2_kernel<true, false, 0, 0, 16, true, false>" $ALG_DD bench.py --workload diffdel --steps 2 --warmup 1 --no-cpu-baseline | tail -1 | cut -c1-150
