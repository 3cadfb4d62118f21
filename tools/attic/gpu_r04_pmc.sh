#!/bin/bash
# Round 4: PMC traffic of the `other_workloads` legs that carried `traffic: null` in round 3 (gru_B8192 / 16384 / 32768 and the
# TCN forward), each as FETCH_SIZE / WRITE_SIZE passes of its own command.  Writes gpurun_out/r04_*pmc_traffic_*.json.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
PY=$(command -v python3)
for B in 8192 16384 32768; do
    ALG=$((B * 65536 * 8))
    timeout 900 bash tools/pmc_traffic.sh r04_c gru_B$B 'gru_mfma2_kernel' $ALG tools/other_leg.py gru $B 65536 2
done
# TCN: 4096 x 65536 = 18 chunks of 228 streams; tcn_first_d1_kernel runs once per chunk
OUT=$PWD/gpurun_out
for C in FETCH_SIZE WRITE_SIZE; do
    dir=$OUT/r04_c_pmc_tcn_$C
    rm -rf "$dir"
    timeout 900 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$dir" -o pmc -- "$PY" tools/other_leg.py tcn 4096 65536 2 > "$OUT/r04_c_pmc_tcn_$C.log" 2>&1
    echo "== pmc tcn $C: exit $?"
done
"$PY" tools/summarize_pmc_forward.py "$OUT/r04_c_pmc_tcn" 'tcn_' 'tcn_first_d1_kernel' 18 $((4096 * 65536 * 8)) $((4096 * 65536)) \
    "$OUT/r04_c_pmc_traffic_tcn_forward.json" "tools/other_leg.py tcn 4096 65536 2"
rm -rf gpurun_out/r04_c_pmc_*_FETCH_SIZE gpurun_out/r04_c_pmc_*_WRITE_SIZE
