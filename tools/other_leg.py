#!/usr/bin/env python3
"""One `other_workloads` leg of bench.py as a command of its own (what rocprofv3 wraps for the per-leg PMC passes):
    python3 tools/other_leg.py <gru|diffdel|tcn> <B> [T] [steps]      -> bench.measure_workload(...) as one JSON line
    python3 tools/other_leg.py cli <segments>                          -> bench.cli_workload(...) (the evaluation command end to end)"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

if __name__ == "__main__":
    wl, B = sys.argv[1], int(sys.argv[2])
    if wl == "cli":
        print(json.dumps(bench.cli_workload(torch.device("cuda", 0), False, n_seg=B)))
        sys.exit(0)
    T = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
    print(json.dumps(bench.measure_workload(wl, B, T, steps, 1, False, torch.device("cuda", 0))))
