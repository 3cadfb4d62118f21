#!/bin/bash
# Round 5, third pass: round-5 tests on the consistent tree (DCPreESR in the flush, CLI cache fix), the DCP A/B probe, the
# N-row rooflines with the corrected frame accounting, counter traffic of the ESR launch at the per-GPU shapes of configs[4].
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
PY=$(command -v python3)
TAG=${1:-r05_c}
( time timeout 1500 $PY -m pytest tests/test_gpu_round5.py -q -m gpu ) > "$OUT/${TAG}_tests_r5.log" 2>&1; echo "tests r5 exit $?"; tail -n 25 "$OUT/${TAG}_tests_r5.log" | cut -c1-300
timeout 600 $PY tools/dcp_fused_probe.py > "$OUT/${TAG}_dcp_fused_probe.txt" 2>&1; echo "dcp probe exit $?"; cat "$OUT/${TAG}_dcp_fused_probe.txt"
timeout 600 $PY tools/dcp_fused_probe.py 8192 > "$OUT/${TAG}_dcp_fused_probe_B8192.txt" 2>&1; tail -n 1 "$OUT/${TAG}_dcp_fused_probe_B8192.txt"
timeout 900 $PY tools/nrow_rooflines.py -o "$OUT/${TAG}_nrow_rooflines.json" 2> "$OUT/${TAG}_nrow_rooflines.txt"; echo "nrow exit $?"; cat "$OUT/${TAG}_nrow_rooflines.txt"
for B in 8192 16384 32768; do
    ALG=$((B * 65536 * 12))
    timeout 900 bash tools/pmc_traffic.sh $TAG mfma2_esr_B$B 'gru_mfma2_kernel<true, false, 0, 0, 4, false, true, false>' $ALG bench.py --steps 3 --warmup 1 --batch $B --no-cpu-baseline --no-extra --other off --traffic off | tail -n 1 | cut -c1-300
done
rm -rf "$OUT"/${TAG}_pmc_*_FETCH_SIZE "$OUT"/${TAG}_pmc_*_WRITE_SIZE
