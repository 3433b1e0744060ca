#!/usr/bin/env python3
"""GPU CLI vs CPU oracle on the tandem-repeat genome for one option row; both SAMs land under gpurun_out/ (GPU box).
usage: tools/tandem_diff.py "<options>" """
import os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np
from salt_amd import workload
d = "/tmp/tandem"; os.makedirs(d, exist_ok=True)
genome = workload.make_tandem()
pos, mask = workload.make_snps(genome, 600, seed=5)
fa, snp, prefix = d + "/g.fa", d + "/s.txt", d + "/idx"
workload.write_fasta(fa, "tandem", genome); workload.write_snps(snp, "tandem", genome, pos, mask)
subprocess.run([ROOT + "/salt_amd/bin/salt-idx", "-k", "21", fa, snp, prefix], check=True, stderr=subprocess.DEVNULL)
seqs, offs, _, _ = workload.make_reads(genome, pos, mask, 400, 100, seed=13)
workload.write_fastq(d + "/se.fq", seqs, offs)
ps, po, _, _ = workload.make_pairs(genome, pos, mask, 100, 150, seed=9, insert_mean=400, insert_sd=40)
o1 = np.arange(101, dtype=np.uint32) * 150
workload.write_fastq(d + "/p1.fq", np.concatenate([ps[po[2 * i]:po[2 * i + 1]] for i in range(100)]), o1)
workload.write_fastq(d + "/p2.fq", np.concatenate([ps[po[2 * i + 1]:po[2 * i + 2]] for i in range(100)]), o1)
args = sys.argv[1].split()
files = [d + "/p1.fq", d + "/p2.fq"] if "-p" in args else [d + "/se.fq"]
strip = lambda out: b"".join(l for l in out.splitlines(keepends=True) if not l.startswith(b"@PG"))
out = ROOT + "/gpurun_out/"
open(out + "tandem_gpu.sam", "wb").write(strip(subprocess.run([ROOT + "/salt_amd/bin/salt"] + args + [prefix] + files, capture_output=True).stdout))
open(out + "tandem_ora.sam", "wb").write(strip(subprocess.run([ROOT + "/oracle/salt_oracle"] + args + [prefix] + files, capture_output=True).stdout))
