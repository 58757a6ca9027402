#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output (one dir per pass) for the hot kernel: per-dispatch mean of every counter."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
kernel_key = sys.argv[2] if len(sys.argv) > 2 else "remap_views_kernel"
vals = defaultdict(list)
for f in sorted(glob.glob(os.path.join(root, "p*", "**", "*counter_collection.csv"), recursive=True)):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if kernel_key not in row.get("Kernel_Name", "") or "rest_kernel" in row.get("Kernel_Name", ""):
                continue
            vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(vals):
    v = vals[k]
    print("%-40s n=%-3d mean=%.6g min=%.6g max=%.6g" % (k, len(v), sum(v) / len(v), min(v), max(v)))
