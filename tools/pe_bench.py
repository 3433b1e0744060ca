#!/usr/bin/env python3
"""Paired-end path on the chr21-scale workload (SURVEY 8d config 4 shape: 2 x 150 bp, insert N(400,50), -p -a 250 -b 550):
GPU rows vs the CPU oracle on a sample, and the PCIe-inclusive rate of salt_gpu_align_pe.  Run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import numpy as np
import torch
torch.cuda.set_device(0)                                   # torch first: it must see the device before any other HIP user
import salt_amd
from salt_amd import workload
import oracle_py

cache = os.environ.get("SALT_BENCH_CACHE", "/tmp/salt_bench_cache")
w = workload.prepare(os.environ.get("SALT_PE_WORKLOAD", "chr21"), cache)
n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
n_check = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
L = 150
seqs, offs, _, _ = workload.make_pairs(w["genome"], w["snp_pos"], w["snp_mask"], n_pairs, L, seed=3, insert_mean=400, insert_sd=50,
                                       damaged=0.03, orphan=0.01)
idx = salt_amd.Index.reload(w["prefix"], rebuild_lkt=False)
aln = salt_amd.GpuAligner(idx, max_reads=2 * n_pairs, max_bases=2 * n_pairs * L)
opt, _ = salt_amd.AlnOpt.from_argv(["-p", "-a", "250", "-b", "550"], idx.l_seed)
aln.alnpe_core1(opt, idx, seqs[:offs[2000]], offs[:2001])                       # warm-up (pac upload, scratch)
t0 = time.perf_counter()
res = aln.alnpe_core1(opt, idx, seqs, offs)
dt = time.perf_counter() - t0
print("GPU PE: %d pairs (2 x %d bp) in %.3f s = %.2f Mreads/s (host buffers in, results out; PCIe-inclusive)" % (n_pairs, L, dt, 2 * n_pairs / dt / 1e6))
print("mapped mates %.2f %%, rescued (soft-clipped or SW-scored) mates %d" % (100.0 * (res["pos"] != 0xFFFFFFFF).mean(),
      int(((res["seq_start"] != 0) | (res["seq_end"] != L - 1)).sum())))
# kernels only: inputs and results resident in HBM
dev = torch.device("cuda:0")
d_seqs = torch.from_numpy(seqs).to(dev); d_offs = torch.from_numpy(offs.view(np.int32)).to(dev)
d_res = torch.zeros(2 * n_pairs * salt_amd.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    aln.align_pe_resident(opt, idx, n_pairs, L, d_seqs.data_ptr(), d_offs.data_ptr(), d_res.data_ptr(), st)
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 5
for _ in range(K):
    aln.align_pe_resident(opt, idx, n_pairs, L, d_seqs.data_ptr(), d_offs.data_ptr(), d_res.data_ptr(), st)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
# the same batches dealt to 4 workspaces on 4 streams (as bench.py does for SE); SALT_PE_STREAMS=1 skips it (kernel profiles)
NS = int(os.environ.get("SALT_PE_STREAMS", "4"))
forks = [aln] + [aln.fork() for _ in range(NS - 1)]
streams = [torch.cuda.Stream() for _ in range(NS)]
d_ress = [d_res] + [torch.zeros_like(d_res) for _ in range(NS - 1)]
for rep in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(3 * K):
        forks[i % NS].align_pe_resident(opt, idx, n_pairs, L, d_seqs.data_ptr(), d_offs.data_ptr(), d_ress[i % NS].data_ptr(), streams[i % NS].cuda_stream)
    torch.cuda.synchronize()
    dt3 = (time.perf_counter() - t0) / (3 * K)
print("GPU PE resident, %d streams: %.3f ms per %d pairs = %.1f Mreads/s (mates/s), equal results: %s" % (
      NS, dt3 * 1e3, n_pairs, 2 * n_pairs / dt3 / 1e6, all(torch.equal(d_ress[0], x) for x in d_ress[1:])))
for f in forks[1:]:
    f.close()
same = np.array_equal(d_res.cpu().numpy().view(salt_amd.RESULT_DTYPE)[["pos", "strand", "mapq"]], res[["pos", "strand", "mapq"]])
pc = aln.pe_counts()
print("rescue requests %d (%.1f %% of the mates), overflowed %d" % (pc[0], 100.0 * pc[0] / (2 * n_pairs), pc[4]))
if os.environ.get("SALT_PE_RAW"):
    print("pe counters", [int(x) for x in pc])
if os.environ.get("SALT_GPU_SW_SKIP_TB") in ("4", "8"):
    print("k_sw columns per request %.1f, lazy-F stripe steps per column %.1f" % (pc[5] / max(pc[0], 1), pc[6] / max(pc[5], 1)))
elif pc[5]:
    print("k_sw phase clocks per request (us): forward %.1f, reverse %.1f, traceback %.1f" % tuple(x / max(pc[0], 1) / 100.0 for x in pc[5:8]))
print("GPU PE resident: %.3f ms per %d pairs = %.1f Mreads/s (mates/s), rows equal to the host-call rows: %s" % (dt * 1e3, n_pairs, 2 * n_pairs / dt / 1e6, same))
ora = oracle_py.Oracle(w["prefix"])
oo = ora.opt()
cores = min(os.cpu_count() or 1, 64)
t0 = time.perf_counter()
want = ora.align_pe(oo, seqs[:offs[2 * n_check]], offs[:2 * n_check + 1], opt.min_tlen, opt.max_tlen, n_threads=cores)
dt = time.perf_counter() - t0
bad = oracle_py.compare(res[:2 * n_check], want, pe=True)
print("oracle PE: %d pairs in %.2f s on %d threads = %.3f Mreads/s; mismatching mates vs GPU: %d of %d" % (n_check, dt, cores, 2 * n_check / dt / 1e6, len(bad), 2 * n_check))
for i in bad[:5]:
    print("  mate", int(i), [(f, res[f][i].tolist(), want[f][i].tolist()) for f in ("pos", "strand", "n_diff", "is_gap", "mapq", "b0", "b1", "seq_start", "seq_end")])
