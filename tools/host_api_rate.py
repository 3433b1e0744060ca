#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point salt_gpu_align_se on the chr21 workload (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import salt_amd
from salt_amd import workload
w = workload.prepare(sys.argv[1] if len(sys.argv) > 1 else "chr21", os.environ.get("SALT_BENCH_CACHE", "/tmp/salt_bench_cache"))
n = 1_000_000
seqs, offs, _, _ = workload.make_reads(w["genome"], w["snp_pos"], w["snp_mask"], n, 100, seed=1)
idx = salt_amd.Index.reload(w["prefix"], rebuild_lkt=False)
aln = salt_amd.GpuAligner(idx, max_reads=n, max_bases=n * 100)
opt = salt_amd.AlnOpt(l_seed=w["k"])
aln.alnse_core1(opt, seqs, offs)
t0 = time.perf_counter()
K = 5
for _ in range(K):
    res = aln.alnse_core1(opt, seqs, offs)
dt = (time.perf_counter() - t0) / K
print("salt_gpu_align_se: %d reads in %.1f ms = %.1f Mreads/s (host buffers in, %d-byte rows out; PCIe-inclusive)" % (n, dt * 1e3, n / dt / 1e6, res.dtype.itemsize))
