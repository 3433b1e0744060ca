#!/bin/bash
# end-to-end (PCIe + host I/O inclusive) rate of the C++ CLI on the chr21-scale workload; run on the GPU box
set -e
cd "$(dirname "$0")/.."
CACHE=${SALT_BENCH_CACHE:-/tmp/salt_bench_cache}
python3 - <<'PY'
import sys, os; sys.path.insert(0, '.')
from salt_amd import workload
cache = os.environ.get("SALT_BENCH_CACHE", "/tmp/salt_bench_cache")
w = workload.prepare("chr21", cache)
fq = os.path.join(w["dir"], "reads_1M.fq")
if not os.path.exists(fq):
    seqs, offs, _, _ = workload.make_reads(w["genome"], w["snp_pos"], w["snp_mask"], 1_000_000, 100, seed=1)
    workload.write_fastq(fq, seqs, offs)
print(w["prefix"], fq)
PY
P=$CACHE/salt_chr21_g40000000_s190000_k21
for t in 1 16; do
  ./salt_amd/bin/salt -d -c -t $t $P/idx $P/reads_1M.fq 2> gpurun_out/e2e_t$t.log > /tmp/e2e.sam
  tail -1 gpurun_out/e2e_t$t.log
done
./oracle/salt_oracle -d -c -t 16 $P/idx $P/reads_1M.fq 2> gpurun_out/e2e_oracle.log | grep -v '^@PG' > /tmp/ora.sam
tail -1 gpurun_out/e2e_oracle.log
grep -v '^@PG' /tmp/e2e.sam | cmp - /tmp/ora.sam && echo "E2E SAM identical to the CPU oracle on 1M reads"
