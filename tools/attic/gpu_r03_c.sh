#!/bin/bash
# round 3, third GPU pass: the whole GPU suite on the fused build, default bench line, kernel stats of gru + diffdel
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
PY=$(command -v python3)
( time timeout 3000 $PY -m pytest tests -q -m gpu -x ) > "$OUT/r03_c_tests.log" 2>&1; echo "tests exit $?"; tail -n 8 "$OUT/r03_c_tests.log"
( time timeout 600 $PY bench.py ) > "$OUT/r03_c_bench_default.json" 2> "$OUT/r03_c_bench_default.err"; echo "bench exit $?"; tail -n 4 "$OUT/r03_c_bench_default.err"
bash tools/profile_all.sh r03_c gru diffdel
