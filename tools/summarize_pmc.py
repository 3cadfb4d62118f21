#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE passes of tools/pmc_traffic.sh -> one small JSON per kernel.
usage: summarize_pmc.py <dir prefix (…_FETCH_SIZE / …_WRITE_SIZE are appended)> <kernel regex> <algorithmic bytes> <out.json> [command]"""
import csv
import glob
import json
import os
import re
import sys

prefix, regex, alg, dst = sys.argv[1], re.compile(sys.argv[2]), float(sys.argv[3]), sys.argv[4]
cmd = sys.argv[5] if len(sys.argv) > 5 else ""
out = {"kernel_regex": sys.argv[2], "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- python3 " + cmd
       + " (two separate passes)",
       "note": "gfx950: FETCH_SIZE reports 1/2 of the bytes of a wide coalesced streaming read (MI355X_MICROARCH.md, HBM "
               "section) -> doubled; WRITE_SIZE is exact for 16-byte-per-lane streaming stores.  Units: KB of 1024 B.  "
               "Infinity-Cache hits are counted by these counters, not excluded."}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(os.path.join(prefix + "_" + c, "**", "*counter_collection.csv"), recursive=True)
    vals = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c and regex.search(r["Kernel_Name"]):
                vals.setdefault(r["Kernel_Name"][:80], []).append(float(r["Counter_Value"]))
    out[c + "_KB_per_launch"] = {k: v for k, v in vals.items()}
    allv = [x for v in vals.values() for x in v]
    # the largest launches are the workload; small ones (warm-start with B = 1) are left out of the mean
    big = [x for x in allv if x > 0.5 * max(allv)] if allv else []
    # median: a launch that ran while another kernel was still on a side stream (bench.py overlaps the ESR pass of the
    # previous step) also counts that kernel's bytes in its window
    out[c + "_KB_median_of_full_size_launches"] = sorted(big)[len(big) // 2] if big else None
f, w = out["FETCH_SIZE_KB_median_of_full_size_launches"], out["WRITE_SIZE_KB_median_of_full_size_launches"]
if f is not None and w is not None:
    out["hbm_bytes_per_launch_corrected"] = (2.0 * f + w) * 1024.0
    out["algorithmic_bytes_per_launch"] = alg
    out["traffic_over_algorithmic"] = out["hbm_bytes_per_launch_corrected"] / alg
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if not k.endswith("per_launch") or k.startswith("hbm") or k.startswith("alg")}))
