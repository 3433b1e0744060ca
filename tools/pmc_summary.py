#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc passes -> pmc_summary_<workload>.json (what bench.py's roofline block reads).

usage: pmc_summary.py <out.json> <workload> <kernel_stats.csv> <counter_collection.csv> [...]
       pmc_summary.py --gather <counter_collection.csv> <loads per kernel>
Per kernel and counter: mean over dispatches of the counter value (summed over the rows rocprofv3 prints for one dispatch),
skipping the first two dispatches of each kernel (bench warm-up).
  hbm_bytes_per_launch = (FETCH_SIZE + WRITE_SIZE) x 1024, raw: for the gather shapes of these kernels tools/ubench/gather shows
      what FETCH_SIZE reports per random load (gather_rate.txt), so no x2 streaming correction is applied.
  issue = per kernel: VALU / SALU / VMEM / LDS wave-instructions per launch and valu_issue_frac = VALU x 2 cycles / (1024 SIMDs x
      duration x 2.4 GHz): the share of the chip's vector issue slots the kernel used (MI355X_MICROARCH.md: a wave64 VALU instruction
      occupies its SIMD-32 for 2 cycles; 2.4 GHz is the peak clock, so the fraction is a lower bound).
Profile runs are made with `bench.py --no-counters`: no launch of a counter-collecting kernel variant is among the dispatches."""
import csv
import json
import sys
from collections import defaultdict

KERNELS = ("k_pack", "k_seed_walk", "k_seed", "k_light2", "k_light", "k_queue_pack", "k_heavy_pe_big", "k_heavy_pe", "k_heavy_big", "k_heavy", "k_gapfin", "k_gap", "k_cigar",
           "k_pair", "k_swtb", "k_swf1", "k_swf", "k_swr", "k_pe_final", "k_fq_count", "k_fq_lines", "k_fq_parse", "k_fq_codes", "k_sam_len", "k_sam_write", "k_heads")
CLOCK_GHZ = 2.4          # MI355X_MICROARCH.md: peak engine clock; issue fractions are quoted against it (a lower bound of the true fraction)


def short(name):
    """Every kernel under its own name: k_light2 (two reads per wave, what the timed steps run) and k_light (one read per wave: paired-end
    mates, long reads, steps with the access counters on) are different kernels and are never folded together."""
    for k in KERNELS:
        if "salt::%s(" % k in name or "salt::%s<" % k in name:       # plain and templated kernels
            return k
    return None


def read_pmc(files):
    per = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))      # kernel -> counter -> dispatch -> value
    for fn in files:
        with open(fn, newline="") as f:
            for row in csv.DictReader(f):
                k = short(row["Kernel_Name"]) or row["Kernel_Name"]
                per[k][row["Counter_Name"]][int(row["Dispatch_Id"])] += float(row["Counter_Value"])
    return per


def gather(fn, loads):
    per = read_pmc([fn])
    for k, ctrs in per.items():
        if "k_gather" not in k:
            continue
        by = ctrs.get("FETCH_SIZE", {})
        vals = [by[d] for d in sorted(by)]
        if vals:
            w = k.split("<")[1].split(">")[0] if "<" in k else "?"
            if w.strip() == "1024":                      # runs of 64 consecutive 16-byte rows: 16 sectors per run, one run per 64 lane-loads
                print("FETCH_SIZE per 64-byte sector of a random 1 KB run of 16-byte rows: %.1f bytes (mean of %d launches)" % (sum(vals) / len(vals) * 1024 / (loads / 4), len(vals)))
                continue
            n = loads / 4 if w.strip() == "64" else loads
            print("FETCH_SIZE per random %s-byte record: %.1f bytes (mean of %d launches)" % (w, sum(vals) / len(vals) * 1024 / n, len(vals)))


def main():
    if sys.argv[1] == "--gather":
        return gather(sys.argv[2], float(sys.argv[3]))
    out_path, workload, stats, files = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4:]
    dur = {}
    with open(stats, newline="") as f:
        for row in csv.DictReader(f):
            k = short(row["Name"])
            if k:
                dur[k] = float(row["AverageNs"]) / 1e6            # ms
    per = read_pmc(files)
    summary = {}
    for k, ctrs in per.items():
        if k not in KERNELS:
            continue
        summary[k] = {}
        for c, by in ctrs.items():
            vals = [by[d] for d in sorted(by)][2:] or [by[d] for d in sorted(by)]
            summary[k][c] = sum(vals) / len(vals)
    traffic = {k: int((v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0)) * 1024) for k, v in summary.items()
               if "FETCH_SIZE" in v and "WRITE_SIZE" in v}
    issue = {}
    for k, v in summary.items():
        if "SQ_INSTS_VALU" not in v or k not in dur:
            continue
        ms = dur[k]
        ghz = CLOCK_GHZ
        issue[k] = {"valu_wave_insts": int(v["SQ_INSTS_VALU"]), "salu_wave_insts": int(v.get("SQ_INSTS_SALU", 0)), "vmem_rd_wave_insts": int(v.get("SQ_INSTS_VMEM_RD", 0)),
                    "lds_wave_insts": int(v.get("SQ_INSTS_LDS", 0)), "waves": int(v.get("SQ_WAVES", 0)), "kernel_ms_rocprof": round(ms, 4),
                    "wait_any_frac": round(v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], 3) if v.get("SQ_WAIT_ANY") and v.get("SQ_WAVE_CYCLES") else None,
                    "valu_issue_frac": round(v["SQ_INSTS_VALU"] * 2 / (1024 * ms / 1e3 * ghz * 1e9), 4)}
    json.dump({"workload": workload, "kernel_ms_rocprof": dur, "per_kernel_mean": summary, "hbm_bytes_per_launch": traffic, "issue": issue}, open(out_path, "w"), indent=1)
    print(json.dumps({"traffic": traffic, "issue": {k: v["valu_issue_frac"] for k, v in issue.items()}}))


if __name__ == "__main__":
    main()
