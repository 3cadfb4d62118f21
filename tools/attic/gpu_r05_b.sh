#!/bin/bash
# Round 5, second pass: the round-5 tests, the default bench line with other_workloads.cli, the N-row rooflines, the counter
# traffic of the ESR-fused launch at the per-GPU shapes of configs[4] (file fallback of an N > 1 line), rocprofv3 stats.
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
PY=$(command -v python3)
TAG=${1:-r05_b}
( time timeout 1500 $PY -m pytest tests/test_gpu_round5.py -q -m gpu ) > "$OUT/${TAG}_tests_r5.log" 2>&1; echo "tests r5 exit $?"; tail -n 25 "$OUT/${TAG}_tests_r5.log"
( time timeout 900 $PY bench.py ) > "$OUT/${TAG}_bench_default.json" 2> "$OUT/${TAG}_bench_default.err"; echo "bench exit $?"; tail -n 4 "$OUT/${TAG}_bench_default.err"
$PY - <<'PYEOF'
import json
d = json.loads(open("gpurun_out/r05_b_bench_default.json").read().strip().splitlines()[-1])
print(json.dumps(d["other_workloads"].get("cli"), indent=1)[:4000])
print(d["value"], d["roofline"]["frac"], d["roofline"]["traffic_source"][:60])
PYEOF
timeout 900 $PY tools/nrow_rooflines.py -o "$OUT/${TAG}_nrow_rooflines.json" 2> "$OUT/${TAG}_nrow_rooflines.txt"; echo "nrow exit $?"; cat "$OUT/${TAG}_nrow_rooflines.txt"
for B in 8192 16384 32768; do
    ALG=$((B * 65536 * 12))
    timeout 900 bash tools/pmc_traffic.sh $TAG mfma2_esr_B$B 'gru_mfma2_kernel<true, false, 0, 0, 4, false, true, false>' $ALG bench.py --steps 3 --warmup 1 --batch $B --no-cpu-baseline --no-extra --other off --traffic off
done
bash tools/profile_all.sh $TAG gru diffdel tcn tape losses
rm -rf "$OUT"/${TAG}_prof_* "$OUT"/${TAG}_pmc_*_FETCH_SIZE "$OUT"/${TAG}_pmc_*_WRITE_SIZE
ls -la "$OUT" | grep ${TAG}
