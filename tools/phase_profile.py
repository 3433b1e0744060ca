#!/usr/bin/env python3
"""k_heavy / k_light phase shares (diagnostic shader-clock counters; run on the GPU box).  usage: phase_profile.py [workload] [n_reads]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
torch.cuda.init()
import salt_amd
from salt_amd import workload
cache = os.environ.get("SALT_BENCH_CACHE", "/tmp/salt_bench_cache")
name = sys.argv[1] if len(sys.argv) > 1 else "chr21"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
dev = torch.device("cuda", 0)
g, p, m = workload.generate_device(name, dev)
w = workload.prepare(name, cache, gpu_device=0, arrays=(g, p, m))
site = workload.make_site_map(g.numel(), p, m)
seqs, offs, _, _ = workload.make_reads_hash(g, site, n, 100, seed=1, batch=0)
torch.cuda.empty_cache()
idx = salt_amd.Index.reload(w["prefix"], rebuild_lkt=False)
aln = salt_amd.GpuAligner(idx, max_reads=n, max_bases=n * 100)
opt = salt_amd.AlnOpt(l_seed=w["k"], collect_counters=1)
d_res = torch.zeros(n * salt_amd.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
aln.align_resident(opt, n, 100, seqs.data_ptr(), offs.data_ptr(), d_res.data_ptr(), st)
torch.cuda.synchronize()
aln.counters()
aln.align_resident(opt, n, 100, seqs.data_ptr(), offs.data_ptr(), d_res.data_ptr(), st)
torch.cuda.synchronize()
c = aln.counters()
print({k: v for k, v in c.items() if not k.startswith("t_") and not k.startswith("lt_")})
tot = sum(v for k, v in c.items() if k.startswith("t_"))
for k, v in c.items():
    if k.startswith("t_"):
        print("%-10s %6.1f %%   %8.1f kcycles/heavy read" % (k, 100.0 * v / max(tot, 1), v / max(c["heavy_reads"], 1) / 1e3))
lt = {k: v for k, v in c.items() if k.startswith("lt_") and k != "lt_samples"}
tot = sum(lt.values())
for k, v in lt.items():
    print("%-10s %6.1f %%   %8.0f ticks/light read (s_memtime, every 128th read)" % (k, 100.0 * v / max(tot, 1), v / max(c["lt_samples"], 1)))
