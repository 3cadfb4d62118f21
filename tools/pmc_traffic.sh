#!/bin/bash
# HBM traffic of named kernels from the PMC counters, collected as MI355X_MICROARCH.md (HBM section) prescribes: FETCH_SIZE
# and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (they do not fit one pass), with --kernel-trace only, the python
# interpreter directly after `--`.  tools/summarize_pmc.py applies the gfx950 correction (FETCH_SIZE x 2 for wide
# coalesced streaming reads) and writes gpurun_out/<tag>_pmc_traffic_<name>.json.
#   bash tools/pmc_traffic.sh <tag> <name> <kernel regex> <algorithmic bytes per launch> <python args...>
set -u
TAG=$1; NAME=$2; RE=$3; ALG=$4; shift 4
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
PY=$(command -v python3)
for C in FETCH_SIZE WRITE_SIZE; do
    dir=$OUT/${TAG}_pmc_${NAME}_$C
    rm -rf "$dir"
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$dir" -o pmc -- "$PY" "$@" > "$OUT/${TAG}_pmc_${NAME}_$C.log" 2>&1
    echo "== pmc $NAME $C: exit $?"
done
"$PY" tools/summarize_pmc.py "$OUT/${TAG}_pmc_${NAME}" "$RE" "$ALG" "$OUT/${TAG}_pmc_traffic_${NAME}.json" "$*"
