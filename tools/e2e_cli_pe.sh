#!/bin/bash
# Paired-end drop-in check at scale: `salt -p -d -c` on 100 000 pairs (2 x 150 bp) of the chr21 workload, SAM compared byte for
# byte with the CPU oracle's (3 batches of 50 000 pairs through the reader / workers / ordered writer).  Run on the GPU box.
set -e
cd "$(dirname "$0")/.."
export SALT_BENCH_CACHE=${SALT_BENCH_CACHE:-/tmp/salt_bench_cache}
read -r P F1 F2 < <(python3 - <<'PY'
import sys, os; sys.path.insert(0, '.')
import numpy as np
from salt_amd import workload
w = workload.prepare("chr21", os.environ["SALT_BENCH_CACHE"])
n, L = 120000, 150
seqs, offs, _, _ = workload.make_pairs(w["genome"], w["snp_pos"], w["snp_mask"], n, L, seed=3, insert_mean=400, insert_sd=50, damaged=0.03, orphan=0.01)
o1 = np.arange(n + 1, dtype=np.uint32) * L
r = seqs.reshape(2 * n, L)
f1, f2 = os.path.join(w["dir"], "pe_1.fq"), os.path.join(w["dir"], "pe_2.fq")
workload.write_fastq(f1, r[0::2].reshape(-1), o1); workload.write_fastq(f2, r[1::2].reshape(-1), o1)
print(w["prefix"], f1, f2)
PY
)
./salt_amd/bin/salt -p -d -c -a 250 -b 550 -t 16 $P $F1 $F2 2> gpurun_out/e2e_pe.log | grep -v '^@PG' > /tmp/pe_gpu.sam
tail -1 gpurun_out/e2e_pe.log
./oracle/salt_oracle -p -d -c -a 250 -b 550 $P $F1 $F2 2>/dev/null | grep -v '^@PG' > /tmp/pe_ora.sam
cmp /tmp/pe_gpu.sam /tmp/pe_ora.sam && echo "PE E2E SAM identical to the CPU oracle on $(grep -c -v '^@' /tmp/pe_ora.sam) lines"
