#!/usr/bin/env python3
"""GPU-box probe of the index builder + aligner at scale: generate a genome of the grch38 model with `genome_len` bases, index it
with the device suffix sorter, load + attach, align one batch and compare a sample with the oracle.  Prints stage times.
Usage: tools/idx_scale.py <genome_len> <n_snps> [contigs] [n_reads] [n_check]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
os.environ.setdefault("SALT_IDX_VERBOSE", "1")

import numpy as np
import torch
import salt_amd
from salt_amd import workload


def main():
    n, n_snps = int(sys.argv[1]), int(sys.argv[2])
    contigs = int(sys.argv[3]) if len(sys.argv) > 3 else 24
    n_reads = int(sys.argv[4]) if len(sys.argv) > 4 else 1_000_000
    n_check = int(sys.argv[5]) if len(sys.argv) > 5 else 50_000
    dev = torch.device("cuda", 0)
    T = {}
    t0 = time.time()
    g = workload.make_genome_hash(n, 38, dev)
    pos, mask = workload.make_snps_hash(g, n_snps, 144)
    torch.cuda.synchronize()
    T["generate"] = time.time() - t0
    print("[scale] generated %d bases, %d SNPs in %.1f s" % (n, n_snps, T["generate"]), flush=True)
    t0 = time.time()
    contigs_in, groups = workload.as_builder_input(g.cpu().numpy(), pos.cpu().numpy(), mask.cpu().numpy(), contigs)
    T["letters"] = time.time() - t0
    site = workload.make_site_map(n, pos, mask)
    d = os.environ.get("SALT_SCALE_DIR", "/tmp/salt_scale")
    os.makedirs(d, exist_ok=True)
    prefix = os.path.join(d, "idx")
    t0 = time.time()
    salt_amd.idx_build_mem(contigs_in, groups, prefix, 21, gpu_device=0, flags=salt_amd.IDX_NO_LP)
    T["index"] = time.time() - t0
    del contigs_in, groups
    print("[scale] index built in %.1f s; files: %.2f GB" % (T["index"], sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d)) / 1e9), flush=True)
    t0 = time.time()
    idx = salt_amd.Index.reload(prefix, rebuild_lkt=False)
    T["load"] = time.time() - t0
    t0 = time.time()
    aln = salt_amd.GpuAligner(idx, device=0, max_reads=n_reads, max_bases=n_reads * 100)
    torch.cuda.synchronize()
    T["attach"] = time.time() - t0
    print("[scale] load %.1f s, attach %.1f s, image %.2f GiB (compact %.2f GiB)" % (T["load"], T["attach"], aln.image()[1] / 2**30, aln.image_compact()[1] / 2**30), flush=True)
    seqs, offs, start, rev = workload.make_reads_hash(g, site, n_reads, 100, seed=2, batch=0)
    opt = salt_amd.AlnOpt(l_seed=21)
    d_res = torch.zeros(n_reads * salt_amd.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    st = torch.cuda.Stream(device=dev)
    for _ in range(2):
        aln.align_resident(opt, n_reads, 100, seqs.data_ptr(), offs.data_ptr(), d_res.data_ptr(), st.cuda_stream)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(5):
        aln.align_resident(opt, n_reads, 100, seqs.data_ptr(), offs.data_ptr(), d_res.data_ptr(), st.cuda_stream)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 5
    print("[scale] align: %.3f ms per %d reads = %.1f Mreads/s (one stream, same batch)" % (dt * 1e3, n_reads, n_reads / dt / 1e6), flush=True)
    res = d_res.cpu().numpy().view(salt_amd.RESULT_DTYPE)
    mapped = res["pos"] != 0xFFFFFFFF
    s_np, r_np = start.cpu().numpy(), rev.cpu().numpy()
    exact = mapped & (res["pos"].astype(np.int64) == s_np)
    print("[scale] mapped %.4f, at the simulated position %.4f" % (mapped.mean(), exact.mean()), flush=True)
    if n_check:
        import oracle_py
        t0 = time.time()
        ora = oracle_py.Oracle(prefix)
        T["oracle_load"] = time.time() - t0
        oo = ora.opt()
        hs, ho = seqs[:n_check * 100].cpu().numpy(), offs[:n_check + 1].cpu().numpy().view(np.uint32)
        t0 = time.time()
        want = ora.align(oo, hs, ho, n_threads=min(os.cpu_count() or 1, 64))
        T["oracle_align"] = time.time() - t0
        bad = oracle_py.compare(res[:n_check], want)
        print("[scale] oracle: load %.1f s, %d reads in %.1f s (%.3f Mreads/s); MISMATCHES: %d" % (T["oracle_load"], n_check, T["oracle_align"], n_check / T["oracle_align"] / 1e6, len(bad)), flush=True)
        for i in bad[:5]:
            print("   read", i, "gpu", res[i]["pos"], res[i]["strand"], res[i]["n_diff"], "oracle", want[i]["pos"], want[i]["strand"], want[i]["n_diff"])
        ora.close()
    print("[scale] times:", {k: round(v, 2) for k, v in T.items()})
    aln.close(); idx.destroy()


if __name__ == "__main__":
    main()
