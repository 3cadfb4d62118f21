#!/bin/bash
# SQ / GRBM counters of named kernels (issue mix, waits, busy cycles) in separate rocprofv3 --pmc passes (a pass holds
# a handful of counters), --kernel-trace only, the python interpreter directly after `--`.
#   bash tools/pmc_sq.sh <tag> <name> <kernel regex> <python args...>   -> gpurun_out/<tag>_pmc_sq_<name>.json
set -u
TAG=$1; NAME=$2; RE=$3; shift 3
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
PY=$(command -v python3)
i=0
for SET in "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
    dir=$OUT/${TAG}_pmc_sq_${NAME}_$i
    rm -rf "$dir"
    rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$dir" -o pmc -- "$PY" "$@" > "$OUT/${TAG}_pmc_sq_${NAME}_$i.log" 2>&1
    echo "== pmc_sq $NAME pass $i: exit $?"
    i=$((i + 1))
done
"$PY" - "$OUT/${TAG}_pmc_sq_${NAME}" "$RE" "$OUT/${TAG}_pmc_sq_${NAME}.json" "$*" <<'PYEOF'
import csv, glob, json, os, re, sys
prefix, regex, dst, cmd = sys.argv[1], re.compile(sys.argv[2]), sys.argv[3], sys.argv[4]
vals, dur = {}, []
for d in sorted(glob.glob(prefix + "_[0-9]")):
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if regex.search(r["Kernel_Name"]):
                vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if regex.search(r["Kernel_Name"]):
                dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
big = lambda v: [x for x in v if x > 0.5 * max(v)] if v and max(v) > 0 else v
out = {"kernel_regex": sys.argv[2], "command": "rocprofv3 --kernel-trace --pmc <set> --output-format csv -- python3 " + cmd + " (three passes)",
       "per_launch_mean": {k: sum(big(v)) / len(big(v)) for k, v in vals.items()},
       "kernel_ns_mean": sum(big(dur)) / len(big(dur)) if dur else None}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out))
PYEOF
