#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats output directory into a small committed summary.
usage: summarize_prof.py <rocprof_dir> <out.csv> [note] [kernel-name regex for the per-dispatch lines]"""
import csv
import glob
import os
import re
import sys

src, dst = sys.argv[1], sys.argv[2]
note = sys.argv[3] if len(sys.argv) > 3 else ""
pat = re.compile(sys.argv[4] if len(sys.argv) > 4 else r"gru_|delay_|tcn_|tape_|esr_|stft_|demod")
f = sorted(glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True))[0]
rows = list(csv.DictReader(open(f)))
with open(dst, "w", newline="") as o:
    if note:
        o.write(f"# {note}\n")
    o.write(f"# source: rocprofv3 --kernel-trace --stats ({os.path.basename(f)})\n")
    w = csv.writer(o)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows[:12]:
        w.writerow([r["Name"][:90], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                    r["MinNs"], r["MaxNs"]])
# per-dispatch durations of the GRU kernel from the trace
t = sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True))
if t:
    d = [r for r in csv.DictReader(open(t[0])) if pat.search(r["Kernel_Name"])][:60]
    with open(dst, "a") as o:
        o.write("# per-dispatch kernel launches (ns, first 60): name, grid, workgroup, vgpr, lds, duration\n")
        for r in d:
            o.write("# %s, grid=%s, wg=%s, vgpr=%s, accum_vgpr=%s, lds=%s, dur=%d\n" % (
                r["Kernel_Name"][:60], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")),
                r.get("VGPR_Count", ""), r.get("Accum_VGPR_Count", ""), r.get("LDS_Block_Size", ""),
                int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
