#!/usr/bin/env python3
"""Debug: PE fields GPU vs oracle on the tandem genome for given options (GPU box)."""
import os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import salt_amd, oracle_py
from salt_amd import workload
d = "/tmp/tandem"; os.makedirs(d, exist_ok=True)
genome = workload.make_tandem()
pos, mask = workload.make_snps(genome, 600, seed=5)
fa, snp, prefix = d + "/g.fa", d + "/s.txt", d + "/idx"
workload.write_fasta(fa, "tandem", genome); workload.write_snps(snp, "tandem", genome, pos, mask)
subprocess.run([ROOT + "/salt_amd/bin/salt-idx", "-k", "21", fa, snp, prefix], check=True, stderr=subprocess.DEVNULL)
seqs, offs, _, _ = workload.make_pairs(genome, pos, mask, 100, 150, seed=9, insert_mean=400, insert_sd=40)
idx = salt_amd.Index.reload(prefix)
opt, _ = salt_amd.AlnOpt.from_argv(sys.argv[1].split(), idx.l_seed)
aln = salt_amd.GpuAligner(idx, device=0, max_reads=200, max_bases=200 * 150)
res = aln.alnpe_core1(opt, idx, seqs, offs)
ora = oracle_py.Oracle(prefix)
oo = ora.opt(l_overlap=opt.l_overlap, max_seed=opt.max_seed, max_locate=opt.max_locate, seed_only_ref=opt.seed_only_ref)
want = ora.align_pe(oo, seqs, offs, opt.min_tlen, opt.max_tlen, n_threads=8)
bad = oracle_py.compare(res, want, pe=True)
print("bad", bad.tolist())
for i in bad[:6]:
    for nm, r in (("gpu", res[i]), ("ora", want[i])):
        print(i, nm, "pos", r["pos"], "str", r["strand"], "nd", r["n_diff"], "gap", r["is_gap"], "mapq", r["mapq"], "b0", r["b0"], "b1", r["b1"], "nh", r["n_hits"].tolist(),
              "hits0", [(int(h["pos"]), int(h["n_diff"])) for h in r["hits"][0][:r["n_hits"][0]]], "hits1", [(int(h["pos"]), int(h["n_diff"])) for h in r["hits"][1][:r["n_hits"][1]]],
              "ss", r["seq_start"], r["seq_end"])
# the SE stage alone with PE semantics is not exposed; show SE semantics for reference
