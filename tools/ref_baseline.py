#!/usr/bin/env python3
"""The REAL reference binary (oracle/_ref/salt, built from /root/reference by oracle/Makefile and shipped as a binary) on the bench's
GRCh38-scale index and a sample of its reads: its own [alnse_core] clock.  usage: tools/ref_baseline.py <n_reads> [threads]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
torch.cuda.init()
from salt_amd import workload
name = os.environ.get("SALT_E2E_WORKLOAD", "grch38")
n = int(sys.argv[1]); threads = sys.argv[2] if len(sys.argv) > 2 else "64"
cache = os.environ.get("SALT_BENCH_CACHE", "/tmp/salt_bench_cache")
dev = torch.device("cuda", 0)
g, p, m = workload.generate_device(name, dev)
w = workload.prepare(name, cache, gpu_device=0, arrays=(g, p, m))
site = workload.make_site_map(g.numel(), p, m)
fq = os.path.join(w["dir"], "ref_sample.fq")
seqs, _, _, _ = workload.make_reads_hash(g, site, n, 100, seed=77, batch=0)
open(fq, "wb").write(workload.fastq_bytes(seqs.cpu().numpy(), n, 100, first_id=0))
del g, site
torch.cuda.empty_cache()
print(sorted(os.listdir(w["dir"]))[:30])
for exe, threads in [(os.path.join(ROOT, "oracle", "_ref", "salt"), t) for t in threads.split(",")] + [(os.path.join(ROOT, "oracle", "salt_oracle"), threads.split(",")[-1])]:
    extra = []
    t0 = time.time()
    r = subprocess.run([exe, "-d", "-c", "-t", threads, w["prefix"], fq] + extra, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=900)
    dt = time.time() - t0
    tail = [l for l in r.stderr.decode(errors="replace").splitlines() if "total" in l or "rror" in l or "load" in l.lower()]
    print("== %s -t %s: rc %d, process %.1f s" % (os.path.basename(exe), threads, r.returncode, dt))
    for l in tail[-6:]:
        print("   ", l)
