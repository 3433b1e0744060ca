#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import numpy as np
import salt_amd, oracle_py
from salt_amd import workload
w = workload.prepare("chr21", os.environ.get("SALT_BENCH_CACHE", "/tmp/salt_bench_cache"))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
seqs, offs, _, _ = workload.make_reads(w["genome"], w["snp_pos"], w["snp_mask"], n, 100, seed=1)
idx = salt_amd.Index.reload(w["prefix"], rebuild_lkt=False)
aln = salt_amd.GpuAligner(idx, max_reads=n, max_bases=n * 100)
opt = salt_amd.AlnOpt(l_seed=w["k"])
res = aln.alnse_core1(opt, seqs, offs)
ora = oracle_py.Oracle(w["prefix"])
want = ora.align(ora.opt(), seqs, offs, n_threads=64)
bad = oracle_py.compare(res, want)
print("mismatching", len(bad))
for i in bad[:12]:
    print(int(i), [(f, res[f][i].tolist(), want[f][i].tolist()) for f in ("pos", "strand", "n_diff", "is_gap", "mapq", "b0", "b1", "n_hits")])
res2 = aln.alnse_core1(opt, seqs, offs)
bad2 = oracle_py.compare(res2, want)
print("second run mismatching", len(bad2), "same set:", set(bad.tolist()) == set(bad2.tolist()), "overlap", len(set(bad.tolist()) & set(bad2.tolist())))
