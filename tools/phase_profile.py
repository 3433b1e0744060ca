#!/usr/bin/env python3
"""k_heavy phase shares on the chr21 workload (diagnostic counters; run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import salt_amd
from salt_amd import workload
cache = os.environ.get("SALT_BENCH_CACHE", "/tmp/salt_bench_cache")
w = workload.prepare(sys.argv[1] if len(sys.argv) > 1 else "chr21", cache)
seqs, offs, _, _ = workload.make_reads(w["genome"], w["snp_pos"], w["snp_mask"], 200000, 100, seed=1)
idx = salt_amd.Index.reload(w["prefix"], rebuild_lkt=False)
aln = salt_amd.GpuAligner(idx, max_reads=200000, max_bases=200000 * 100)
opt = salt_amd.AlnOpt(l_seed=w["k"], collect_counters=1)
aln.alnse_core1(opt, seqs, offs)
c = aln.counters()
print(c)
tot = sum(v for k, v in c.items() if k.startswith("t_"))
for k, v in c.items():
    if k.startswith("t_"):
        print("%-10s %6.1f %%   %8.1f kcycles/heavy read" % (k, 100.0 * v / tot, v / max(c["heavy_reads"], 1) / 1e3))
lt = {k: v for k, v in c.items() if k.startswith("lt_") and k != "lt_samples"}
tot = sum(lt.values())
for k, v in lt.items():
    print("%-10s %6.1f %%   %8.0f ticks/light read (s_memtime, every 128th read)" % (k, 100.0 * v / max(tot, 1), v / max(c["lt_samples"], 1)))
