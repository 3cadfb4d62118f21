#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE passes of tools/pmc_traffic.sh for a workload whose forward is SEVERAL launches (the TCN: first
block + three matrix-pipe blocks per stream chunk): HBM bytes per FORWARD = all matching launches summed / number of forwards.
usage: summarize_pmc_forward.py <dir prefix> <kernel regex> <regex of the kernel launched ONCE per chunk> <chunks per forward>
                                <algorithmic bytes per forward> <samples per forward> <out.json> [command]"""
import csv
import glob
import json
import os
import re
import sys

prefix, rx, rx_first, chunks, alg, samples, dst = (sys.argv[1], re.compile(sys.argv[2]), re.compile(sys.argv[3]), int(sys.argv[4]),
                                                    float(sys.argv[5]), float(sys.argv[6]), sys.argv[7])
cmd = sys.argv[8] if len(sys.argv) > 8 else ""
out = {"kernel_regex": sys.argv[2], "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- python3 " + cmd
       + " (two separate passes)",
       "note": "gfx950: FETCH_SIZE x 2 (wide coalesced streaming reads report half), WRITE_SIZE as is; KB of 1024 B; per FORWARD = "
               "sum over every matching launch / forwards, forwards = launches of the once-per-chunk kernel / chunks per forward."}
tot = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    per, n_first = {}, 0
    for f in glob.glob(os.path.join(prefix + "_" + c, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c or not rx.search(r["Kernel_Name"]):
                continue
            k = r["Kernel_Name"][:60]
            per[k] = per.get(k, 0.0) + float(r["Counter_Value"])
            n_first += 1 if rx_first.search(r["Kernel_Name"]) else 0
    forwards = n_first / chunks
    out[c + "_KB_per_forward_by_kernel"] = {k: v / forwards for k, v in per.items()}
    out[c + "_forwards_seen"] = forwards
    tot[c] = sum(per.values()) / forwards
out["hbm_bytes_per_forward_corrected"] = (2.0 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024.0
out["algorithmic_bytes_per_forward"] = alg
out["traffic_over_algorithmic"] = out["hbm_bytes_per_forward_corrected"] / alg
out["bytes_per_sample_moved"] = out["hbm_bytes_per_forward_corrected"] / samples
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if "by_kernel" not in k}))
