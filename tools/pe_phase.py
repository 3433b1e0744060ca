#!/usr/bin/env python3
"""k_heavy_pe phase shares on PE pairs of the chr21 workload (diagnostic counters; GPU box)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import salt_amd
from salt_amd import workload
w = workload.prepare("chr21", os.environ.get("SALT_BENCH_CACHE", "/tmp/salt_bench_cache"))
n_pairs, L = 100000, 150
seqs, offs, _, _ = workload.make_pairs(w["genome"], w["snp_pos"], w["snp_mask"], n_pairs, L, seed=3, insert_mean=400, insert_sd=50, damaged=0.03, orphan=0.01)
idx = salt_amd.Index.reload(w["prefix"], rebuild_lkt=False)
aln = salt_amd.GpuAligner(idx, max_reads=2 * n_pairs, max_bases=2 * n_pairs * L)
opt, _ = salt_amd.AlnOpt.from_argv(["-p", "-a", "250", "-b", "550"], idx.l_seed)
opt.collect_counters = 1
aln.alnpe_core1(opt, idx, seqs, offs)
c = aln.counters()
print({k: v for k, v in c.items() if v and not k.startswith("lt_")})
tot = sum(v for k, v in c.items() if k.startswith("t_"))
for k, v in c.items():
    if k.startswith("t_"):
        print("%-10s %6.1f %%   %8.1f kcycles/heavy read" % (k, 100.0 * v / tot, v / max(c["heavy_reads"], 1) / 1e3))
t, r = c["max_heavy"] >> 32, c["max_heavy"] & 0xFFFFFFFF
print("slowest heavy mate: %.1f us (read %d)" % (t / 100.0, r), "mean us/heavy read", c["x3"] / max(c["heavy_reads"], 1) / 100.0)
