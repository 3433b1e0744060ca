#!/usr/bin/env python3
"""Keeps the rows of a rocprofv3 counter_collection.csv that belong to this repository's kernels (salt::...) and the columns that say
something, so that the evidence fits in git: usage tools/pmc_trim.py in.csv out.csv"""
import csv, sys
src, dst = sys.argv[1], sys.argv[2]
cols = ["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value", "Grid_Size", "Workgroup_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count",
        "LDS_Block_Size", "Scratch_Size"]
with open(src, newline="") as f, open(dst, "w", newline="") as g:
    r = csv.DictReader(f)
    w = csv.writer(g)
    w.writerow(cols)
    for row in r:
        if "salt::" not in row["Kernel_Name"]:
            continue
        row["Kernel_Name"] = row["Kernel_Name"][:60]
        w.writerow([row[c] for c in cols])
