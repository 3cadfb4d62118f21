#!/bin/bash
# round 3, first GPU pass: the new tests, the default bench line (with other_workloads), a tracked profile of the
# wave-per-stream laboratory kernel (gru_valu_kernel) -- run on the GPU box from the repo root
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
PY=$(command -v python3)
timeout 1500 $PY -m pytest tests/test_gpu_round3.py -x -q > "$OUT/r03_a_tests.log" 2>&1; echo "tests exit $?"; tail -n 15 "$OUT/r03_a_tests.log"
( time timeout 600 $PY bench.py ) > "$OUT/r03_a_bench_default.json" 2> "$OUT/r03_a_bench_default.err"; echo "bench exit $?"; tail -n 4 "$OUT/r03_a_bench_default.err"; cut -c1-600 "$OUT/r03_a_bench_default.json"
dir=$OUT/r03_a_prof_valu; rm -rf "$dir"
rocprofv3 --kernel-trace --stats --output-format csv -d "$dir" -o valu -- "$PY" tests/quick_bench.py --variant valu --B 4096 --T 65536 --iters 3 > "$OUT/r03_a_valu.log" 2>&1
echo "valu exit $?"; tail -n 2 "$OUT/r03_a_valu.log"
$PY tools/summarize_prof.py "$dir" "$OUT/r03_a_valu_kernel_stats.csv" "tests/quick_bench.py --variant valu --B 4096 --T 65536 --iters 3 (laboratory kernel: one wavefront per 2 streams, no MFMA)"
