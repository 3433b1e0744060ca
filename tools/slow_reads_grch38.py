#!/usr/bin/env python3
"""The slowest read of k_heavy on the GRCh38-scale workload: what it is and what it costs alone (diagnostic; GPU box).
usage: slow_reads_grch38.py [workload] [n_reads]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
torch.cuda.init()
import salt_amd
from salt_amd import workload
cache = os.environ.get("SALT_BENCH_CACHE", "/tmp/salt_bench_cache")
name = sys.argv[1] if len(sys.argv) > 1 else "grch38"
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
h_seqs = seqs.cpu().numpy(); h_offs = offs.cpu().numpy()
excluded = []
for rnd in range(6):
    aln.align_resident(opt, n, 100, seqs.data_ptr(), offs.data_ptr(), d_res.data_ptr(), st); torch.cuda.synchronize(); aln.counters()
    aln.align_resident(opt, n, 100, seqs.data_ptr(), offs.data_ptr(), d_res.data_ptr(), st); torch.cuda.synchronize()
    c = aln.counters()
    res = d_res.cpu().numpy().view(salt_amd.RESULT_DTYPE)
    t, r = c["max_heavy"] >> 32, c["max_heavy"] & 0xFFFFFFFF
    row = res[r]
    s = h_seqs[h_offs[r]:h_offs[r + 1]]
    print("round %d: slowest read %d: %.1f us (mean %.1f us over %d heavy reads)  pos %d strand %d n_diff %d is_gap %d n_hits %s  seq %s" % (
        rnd, r, t / 100.0, c["x3"] / max(c["heavy_reads"], 1) / 100.0, c["heavy_reads"], row["pos"], row["strand"], row["n_diff"], row["is_gap"], row["n_hits"].tolist(),
        "".join("ACGTN"[min(int(b), 4)] for b in s)), flush=True)
    # the same read alone
    one_s = torch.from_numpy(np.ascontiguousarray(s)).to(dev); one_o = offs[:2].clone()
    one = salt_amd.GpuAligner(idx, max_reads=64, max_bases=6400)
    d1 = torch.zeros(64 * salt_amd.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    one.align_resident(opt, 1, 100, one_s.data_ptr(), one_o.data_ptr(), d1.data_ptr(), st); torch.cuda.synchronize(); one.counters()
    one.align_resident(opt, 1, 100, one_s.data_ptr(), one_o.data_ptr(), d1.data_ptr(), st); torch.cuda.synchronize()
    c1 = one.counters()
    print("   alone:", {k: v for k, v in c1.items() if (k.startswith("t_") or k.startswith("x") or k in ("sa_c", "sa_r", "verify", "lv", "loci", "heavy_reads")) and v}, flush=True)
    one.close()
    # take it out of the batch (replace it by its neighbour) and look for the next one
    excluded.append(int(r))
    q = (r + 1) % n
    seqs[int(h_offs[r]):int(h_offs[r + 1])] = seqs[int(h_offs[q]):int(h_offs[q + 1])]
