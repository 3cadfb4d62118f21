#!/bin/bash
# rocprofv3 --kernel-trace --stats of every BASELINE workload and side kernel, one directory each under
# gpurun_out/<tag>_*; tools/summarize_prof.py condenses each into gpurun_out/<tag>_<name>_kernel_stats.csv (the files
# that get copied into profiles/).  Run on the GPU box from the repo root:  bash tools/profile_all.sh r02_a [names...]
# (rocprofv3 gets the python interpreter itself after `--`: no env/bash hop may sit between them on this pool.)
set -u
TAG=${1:-r02}
shift || true
NAMES=${*:-"gru diffdel tcn tape losses demod"}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
PY=$(command -v python3)

prof() {   # prof <name> <args...>
    local name=$1; shift
    local dir=$OUT/${TAG}_prof_$name
    rm -rf "$dir"
    rocprofv3 --kernel-trace --stats --output-format csv -d "$dir" -o "$name" -- "$PY" "$@" > "$OUT/${TAG}_${name}.log" 2>&1
    echo "== $name: exit $?"
    tail -n 3 "$OUT/${TAG}_${name}.log" | cut -c1-2000
    "$PY" tools/summarize_prof.py "$dir" "$OUT/${TAG}_${name}_kernel_stats.csv" "$name: $*" || echo "summary failed for $name"
}

for n in $NAMES; do
    case $n in
        gru)     prof gru bench.py --steps 5 --warmup 2 --no-cpu-baseline --traffic file ;;
        diffdel) prof diffdel bench.py --workload diffdel --steps 5 --warmup 2 --no-cpu-baseline ;;
        tcn)     prof tcn bench.py --workload tcn --steps 3 --warmup 1 --no-cpu-baseline ;;
        tape)    prof tape tools/tape_probe.py ;;
        losses)  prof losses tools/loss_probe.py ;;
        demod)   prof demod tools/demod_probe.py ;;
        cli)     prof cli tools/other_leg.py cli 128 ;;
    esac
done
