#!/usr/bin/env python3
"""End-to-end paired-end runs of the `salt` binary on the GRCh38-scale workload (GPU box): builds the index once, writes the two FASTQ
files of <n_pairs> pairs of 2 x 150 bases, then runs `salt -d -c -p -a 250 -b 550` for every settings string ("ENV=VAL,ENV=VAL";
"" = defaults; OUT=null writes to /dev/null).   usage: tools/e2e_pe_text.py <n_pairs> [settings ...]"""
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
L = 150
fq = [os.path.join(w["dir"], "e2e_1.fq"), os.path.join(w["dir"], "e2e_2.fq")]
with open(fq[0], "wb") as f1, open(fq[1], "wb") as f2:
    done = 0
    while done < n:
        k = min(500000, n - done)
        seqs, _, _, _, _ = workload.make_pairs_hash(g, site, k, L, seed=78, batch=done // 500000)
        r = seqs.view(k, 2, L).cpu().numpy()
        f1.write(workload.fastq_bytes(r[:, 0, :].reshape(-1), k, L, first_id=done))
        f2.write(workload.fastq_bytes(r[:, 1, :].reshape(-1), k, L, first_id=done))
        done += k
del g, site
torch.cuda.empty_cache()
salt = os.path.join(ROOT, "salt_amd", "bin", "salt")
for s in settings:
    env = dict(os.environ)
    for kv in filter(None, s.split(",")):
        a, b = kv.split("=")
        env[a] = b
    sam = os.path.join(w["dir"], "e2e_pe.sam")
    to_null = env.pop("OUT", "") == "null"
    t0 = time.time()
    with open("/dev/null" if to_null else sam, "wb") as fo:
        cmd = [salt, "-d", "-c", "-p", "-a", "250", "-b", "550", "-t", env.pop("T", "64"), w["prefix"]] + fq
        if env.pop("ROCPROF", ""):                            # kernel-trace statistics of this run under gpurun_out/pe_prof
            cmd = ["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", os.path.join(ROOT, "gpurun_out", "pe_prof"), "-o", "pe", "--"] + cmd
        r = subprocess.run(cmd, stdout=fo, stderr=subprocess.PIPE, env=env)
    tail = [l for l in r.stderr.decode().splitlines() if l.startswith("[salt") or "total" in l]
    print("== %s  (rc %d, process %.1f s)" % (s or "defaults", r.returncode, time.time() - t0))
    for l in tail[-6:]:
        print("   ", l)
    sys.stdout.flush()
