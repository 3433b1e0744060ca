#!/usr/bin/env python3
"""End-to-end experiments with the `salt` binary on the GRCh38-scale workload (GPU box): builds the index once, writes one FASTQ file,
then runs `salt -d -c` for every settings string given on the command line ("ENV=VAL,ENV=VAL"; "" = defaults).
usage: tools/e2e_text.py <n_reads> [settings ...]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
torch.cuda.init()
from salt_amd import workload
name = os.environ.get("SALT_E2E_WORKLOAD", "grch38")
n = int(sys.argv[1]); settings = sys.argv[2:] or [""]
cache = os.environ.get("SALT_BENCH_CACHE", "/tmp/salt_bench_cache")
dev = torch.device("cuda", 0)
g, p, m = workload.generate_device(name, dev)
w = workload.prepare(name, cache, gpu_device=0, arrays=(g, p, m))
site = workload.make_site_map(g.numel(), p, m)
fq = os.path.join(w["dir"], "e2e.fq")
with open(fq, "wb") as f:
    done = 0
    while done < n:
        k = min(1000000, n - done)
        seqs, _, _, _ = workload.make_reads_hash(g, site, k, 100, seed=77, batch=done // 1000000)
        f.write(workload.fastq_bytes(seqs.cpu().numpy(), k, 100, first_id=done))
        done += k
del g, site
torch.cuda.empty_cache()
salt = os.path.join(ROOT, "salt_amd", "bin", "salt")
for s in settings:
    env = dict(os.environ)
    for kv in filter(None, s.split(",")):
        a, b = kv.split("=")
        env[a] = b
    sam = os.path.join(w["dir"], "e2e.sam")
    to_null = env.pop("OUT", "") == "null"
    t0 = time.time()
    with open("/dev/null" if to_null else sam, "wb") as fo:
        r = subprocess.run([salt, "-d", "-c", "-t", env.pop("T", "64"), w["prefix"], fq], stdout=fo, stderr=subprocess.PIPE, env=env)
    tail = [l for l in r.stderr.decode().splitlines() if l.startswith("[salt") or l.startswith("[alnse_core]: total")]
    print("== %s  (rc %d, process %.1f s)" % (s or "defaults", r.returncode, time.time() - t0))
    for l in tail:
        print("   ", l)
    sys.stdout.flush()
