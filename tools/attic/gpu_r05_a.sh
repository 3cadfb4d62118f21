#!/bin/bash
# Round 5, first pass on the MI355X: the new tests (any hidden size, CLI cache, chunked TCN call), the round-4 parity table
# with the f16x3 engine added, the N > 1 bench line over gloo (inside test_gpu_round4), then the default bench line.
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
PY=$(command -v python3)
TAG=${1:-r05_a}
rm -f "$OUT/r05_checkpoint_parity.jsonl"
( time timeout 1500 $PY -m pytest tests/test_gpu_round5.py -q -m gpu -x ) > "$OUT/${TAG}_tests_r5.log" 2>&1; echo "tests r5 exit $?"; tail -n 15 "$OUT/${TAG}_tests_r5.log"
( time timeout 2400 $PY -m pytest tests/test_gpu_round4.py -q -m gpu ) > "$OUT/${TAG}_tests_r4.log" 2>&1; echo "tests r4 exit $?"; tail -n 15 "$OUT/${TAG}_tests_r4.log"
( time timeout 900 $PY bench.py ) > "$OUT/${TAG}_bench_default.json" 2> "$OUT/${TAG}_bench_default.err"; echo "bench exit $?"; tail -n 4 "$OUT/${TAG}_bench_default.err"
head -c 1500 "$OUT/${TAG}_bench_default.json"
