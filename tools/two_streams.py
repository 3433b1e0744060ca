#!/usr/bin/env python3
"""Experiment: K batches through ONE workspace/stream vs alternating over N workspaces on N streams (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
torch.cuda.set_device(0)
import salt_amd
from salt_amd import workload
w = workload.prepare("chr21", os.environ.get("SALT_BENCH_CACHE", "/tmp/salt_bench_cache"))
n, L = 1_000_000, 100
seqs, offs, _, _ = workload.make_reads(w["genome"], w["snp_pos"], w["snp_mask"], n, L, seed=1)
idx = salt_amd.Index.reload(w["prefix"], rebuild_lkt=False)
aln0 = salt_amd.GpuAligner(idx, max_reads=n, max_bases=n * L)
opt = salt_amd.AlnOpt(l_seed=w["k"])
dev = torch.device("cuda:0")
d_seqs = torch.from_numpy(seqs).to(dev); d_offs = torch.from_numpy(offs.view(np.int32)).to(dev)
for NS in (1, 2, 3, 4, 5, 6, 8):
    alns = [aln0] + [aln0.fork() for _ in range(NS - 1)]
    streams = [torch.cuda.Stream() for _ in range(NS)]
    res = [torch.zeros(n * salt_amd.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev) for _ in range(NS)]
    K = 12
    for r in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(K):
            s = i % NS
            alns[s].align_resident(opt, n, L, d_seqs.data_ptr(), d_offs.data_ptr(), res[s].data_ptr(), streams[s].cuda_stream)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    same = all(torch.equal(res[0], r_) for r_ in res[1:])
    print("%d stream(s): %.3f ms per batch = %.1f Mreads/s, results equal: %s" % (NS, dt / K * 1e3, n * K / dt / 1e6, same))
    for a in alns[1:]:
        a.close()
