#!/bin/bash
# round 3, second GPU pass: the fused DiffDelRNN step -- parity tests, A/B bench lines, kernel stats
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
PY=$(command -v python3)
timeout 1500 $PY -m pytest tests/test_gpu_round3.py -x -q -k "fused or raw_ctypes or cpp_process" > "$OUT/r03_b_tests.log" 2>&1; echo "tests exit $?"; tail -n 25 "$OUT/r03_b_tests.log"
timeout 600 $PY __graft_entry__.py smoke > "$OUT/r03_b_smoke.log" 2>&1; echo "smoke exit $?"; tail -n 12 "$OUT/r03_b_smoke.log"
for mode in auto two_pass auto two_pass; do
  timeout 600 $PY bench.py --workload diffdel --steps 8 --warmup 2 --no-cpu-baseline --delay-mode $mode >> "$OUT/r03_b_bench_diffdel.jsonl" 2>> "$OUT/r03_b_bench_diffdel.err"
done
$PY - <<'PYEOF'
import json
for ln in open("gpurun_out/r03_b_bench_diffdel.jsonl"):
    o = json.loads(ln)
    r = o["roofline"]
    print(round(o["ms_per_step"], 3), round(o["device_ms_per_step"], 3), round(r["kernel_ms"], 3), r["kernel"][:50], r.get("two_pass"), r.get("delay_line", {}).get("kernel_ms"))
PYEOF
bash tools/profile_all.sh r03_b diffdel
