#!/usr/bin/env python3
"""Which read is the slowest in k_heavy / k_gapfin on the chr21 workload, and what is it (diagnostic; GPU box)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import salt_amd
from salt_amd import workload
cache = os.environ.get("SALT_BENCH_CACHE", "/tmp/salt_bench_cache")
w = workload.prepare("chr21", cache)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
seqs, offs, _, _ = workload.make_reads(w["genome"], w["snp_pos"], w["snp_mask"], n, 100, seed=1)
idx = salt_amd.Index.reload(w["prefix"], rebuild_lkt=False)
aln = salt_amd.GpuAligner(idx, max_reads=n, max_bases=n * 100)
opt = salt_amd.AlnOpt(l_seed=w["k"], collect_counters=1)
res = aln.alnse_core1(opt, seqs, offs)
c = aln.counters()
for k in ("max_heavy", "max_gapfin"):
    t, r = c[k] >> 32, c[k] & 0xFFFFFFFF
    row = res[r]
    print("%s: %.1f us  read %d  pos %d strand %d n_diff %d is_gap %d n_hits %s hit_gap %s n_cigar %d hit_n_cigar %s" % (
        k, t / 100.0, r, row["pos"], row["strand"], row["n_diff"], row["is_gap"], row["n_hits"].tolist(),
        [[int(h["is_gap"]) for h in row["hits"][s][:row["n_hits"][s]]] for s in range(2)], row["n_cigar"], row["hit_n_cigar"].tolist()))
print("heavy reads", c["heavy_reads"], "mean us/read", c["x3"] / max(c["heavy_reads"], 1) / 100.0)
for k in ("max_heavy", "max_gapfin"):
    r = c[k] & 0xFFFFFFFF
    one = salt_amd.GpuAligner(idx, max_reads=64, max_bases=6400)
    s1, o1 = seqs[offs[r]:offs[r + 1]], np.array([0, 100], dtype=np.uint32)
    one.alnse_core1(opt, s1, o1); one.counters()
    one.alnse_core1(opt, s1, o1)
    c1 = one.counters()
    print(k, "alone:", {kk: v for kk, v in c1.items() if (kk.startswith("t_") or kk.startswith("x") or kk in ("sa_c", "sa_r", "verify", "lv", "loci")) and v})
    one.close()
