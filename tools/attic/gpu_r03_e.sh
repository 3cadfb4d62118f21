#!/bin/bash
# round 3, final pass: the whole GPU suite, smoke, the default bench line, kernel stats of the default line and of the
# DiffDelGRU workload on the final binary
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
PY=$(command -v python3)
( time timeout 3000 $PY -m pytest tests -q -m gpu ) > "$OUT/r03_e_tests.log" 2>&1; echo "tests exit $?"; tail -n 6 "$OUT/r03_e_tests.log"
timeout 600 $PY __graft_entry__.py smoke > "$OUT/r03_e_smoke.log" 2>&1; echo "smoke exit $?"
( time timeout 600 $PY bench.py ) > "$OUT/r03_e_bench_default.json" 2> "$OUT/r03_e_bench_default.err"; echo "bench exit $?"; tail -n 4 "$OUT/r03_e_bench_default.err"
bash tools/profile_all.sh r03_e gru diffdel
