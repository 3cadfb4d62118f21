#!/bin/bash
# Round 4, evidence for the final binary: whole GPU suite, smoke, the default bench line, the DiffDelGRU / TCN lines, and the
# rocprofv3 --kernel-trace --stats summaries of the same commands (gpurun_out/r04_e_*; copied into profiles/).
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
PY=$(command -v python3)
TAG=${1:-r04_e}
rm -f "$OUT/r04_checkpoint_parity.jsonl"
( time timeout 3300 $PY -m pytest tests -q -m gpu ) > "$OUT/${TAG}_tests.log" 2>&1; echo "tests exit $?"; tail -n 6 "$OUT/${TAG}_tests.log"
timeout 600 $PY __graft_entry__.py smoke > "$OUT/${TAG}_smoke.log" 2>&1; echo "smoke exit $?"
( time timeout 900 $PY bench.py ) > "$OUT/${TAG}_bench_default.json" 2> "$OUT/${TAG}_bench_default.err"; echo "bench exit $?"; tail -n 4 "$OUT/${TAG}_bench_default.err"
timeout 600 $PY bench.py --workload diffdel --steps 10 --warmup 2 > "$OUT/${TAG}_bench_diffdel.json" 2> "$OUT/${TAG}_bench_diffdel.err"; echo "bench diffdel exit $?"
timeout 600 $PY bench.py --workload tcn --steps 10 --warmup 2 > "$OUT/${TAG}_bench_tcn.json" 2> "$OUT/${TAG}_bench_tcn.err"; echo "bench tcn exit $?"
bash tools/profile_all.sh $TAG gru diffdel tcn
rm -rf "$OUT"/${TAG}_prof_*
