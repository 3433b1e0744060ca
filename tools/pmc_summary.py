#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc passes -> pmc_summary.json + profiles/hbm_traffic.json entries.

usage: pmc_summary.py <out.json> <workload> <counter_collection.csv> [...]
Per kernel and counter: mean over dispatches of the counter value (summed over the rows rocprofv3 prints for one
dispatch), skipping the first two dispatches of each kernel (bench warm-up).  hbm bytes per launch =
(FETCH_SIZE + WRITE_SIZE) x 1024, raw (see profiles/hbm_traffic.json:_note)."""
import csv
import json
import sys
from collections import defaultdict

KERNELS = ("k_pack", "k_seed", "k_light", "k_heavy_pe", "k_heavy", "k_gapfin", "k_gap", "k_cigar", "k_pair", "k_sw", "k_pe_final")


def short(name):
    if "salt::k_light2<" in name or "salt::k_light2(" in name:     # the two-reads-per-wave variant of k_light; bench.py reports both as k_light
        return "k_light"
    for k in KERNELS:
        if "salt::%s(" % k in name or "salt::%s<" % k in name:       # plain and templated kernels
            return k
    return None


def main():
    out_path, workload, files = sys.argv[1], sys.argv[2], sys.argv[3:]
    per = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))      # kernel -> counter -> dispatch -> value
    for fn in files:
        with open(fn, newline="") as f:
            for row in csv.DictReader(f):
                k = short(row["Kernel_Name"])
                if k:
                    per[k][row["Counter_Name"]][int(row["Dispatch_Id"])] += float(row["Counter_Value"])
    summary = {}
    for k, ctrs in per.items():
        summary[k] = {}
        for c, by in ctrs.items():
            vals = [by[d] for d in sorted(by)][2:] or [by[d] for d in sorted(by)]
            summary[k][c] = sum(vals) / len(vals)
    traffic = {k: int((v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0)) * 1024) for k, v in summary.items()
               if "FETCH_SIZE" in v and "WRITE_SIZE" in v}
    json.dump({"workload": workload, "per_kernel_mean": summary, "hbm_bytes_per_launch": traffic}, open(out_path, "w"), indent=1)
    print(json.dumps(traffic))


if __name__ == "__main__":
    main()
