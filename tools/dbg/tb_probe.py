#!/usr/bin/env python3
"""k_sw slot time of the paired-end path on the chr21-scale workload with the traceback's band limited (diagnostics library:
SALT_GPU_TB_MAXBW=n counts wider bands as overflow and skips them, SALT_GPU_NO_TB skips the traceback) -- where k_swtb's time goes.
usage (GPU box, diagnostics library in place of the release one): tb_probe.py [pairs]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
torch.cuda.set_device(0)
import salt_amd
from salt_amd import workload
cache = os.environ.get("SALT_BENCH_CACHE", "/tmp/salt_bench_cache")
w = workload.prepare("chr21", cache)
n_pairs, L = int(sys.argv[1]) if len(sys.argv) > 1 else 500000, 150
seqs, offs, _, _ = workload.make_pairs(w["genome"], w["snp_pos"], w["snp_mask"], n_pairs, L, seed=3, insert_mean=400, insert_sd=50, damaged=0.03, orphan=0.01)
idx = salt_amd.Index.reload(w["prefix"], rebuild_lkt=False)
aln = salt_amd.GpuAligner(idx, max_reads=2 * n_pairs, max_bases=2 * n_pairs * L)
opt, _ = salt_amd.AlnOpt.from_argv(["-p", "-a", "250", "-b", "550"], idx.l_seed)
dev = torch.device("cuda:0")
d_seqs = torch.from_numpy(seqs).to(dev); d_offs = torch.from_numpy(offs.view(np.int32)).to(dev)
d_res = torch.zeros(2 * n_pairs * salt_amd.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    aln.align_pe_resident(opt, idx, n_pairs, L, d_seqs.data_ptr(), d_offs.data_ptr(), d_res.data_ptr(), st)
torch.cuda.synchronize()
aln.timing(True)
for _ in range(4):
    aln.align_pe_resident(opt, idx, n_pairs, L, d_seqs.data_ptr(), d_offs.data_ptr(), d_res.data_ptr(), st)
torch.cuda.synchronize()
ms, n = aln.kernel_ms()
pc = aln.pe_counts()
print("TB_MAXBW=%s NO_TB=%s: k_sw %.3f ms per call (%d calls), requests %d, overflowed %d" % (os.environ.get("SALT_GPU_TB_MAXBW"), os.environ.get("SALT_GPU_NO_TB"),
      ms.get("k_sw", 0.0) / max(n, 1), n, pc[0], pc[4]))
