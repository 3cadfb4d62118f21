#!/bin/bash
# usage: pmc_multi.sh <tag> <python args...> ; counter sets below, one pass each
TAG=$1; shift
export TMPDIR=/tmp
PY=$(command -v python3)
i=0
for SET in "SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" "SQ_BUSY_CU_CYCLES SQ_INSTS_BRANCH SQ_INST_CYCLES_VMEM_WR SQ_VMEM_WR_TA_DATA_FIFO_FULL"; do
  rocprofv3 --kernel-trace --pmc $SET --output-format csv -d gpurun_out/${TAG}_$i -o pmc -- "$PY" "$@" > gpurun_out/${TAG}_$i.log 2>&1; echo "pass $i exit $?"; i=$((i+1))
done
"$PY" - "$TAG" <<'PYEOF'
import csv,glob,collections,sys,json
tag=sys.argv[1]
vals=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"gpurun_out/{tag}_[0-9]/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "ntm::" in k: vals[k.split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out={k:{c:(sum(v)/len(v)) for c,v in d.items() if max(v)>0} for k,d in vals.items()}
# keep only large launches: mean of values above half the max
for k,d in vals.items():
    for c,v in d.items():
        big=[x for x in v if x>0.5*max(v)] if max(v)>0 else v
        out[k][c]=sum(big)/len(big) if big else 0
json.dump(out,open(f"gpurun_out/{tag}.json","w"),indent=1)
print(json.dumps(out))
PYEOF
