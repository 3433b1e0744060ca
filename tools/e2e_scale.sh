#!/bin/bash
# end-to-end CLI rate vs host threads on 4M reads of the chr21 workload (run on the GPU box; output discarded to /dev/null)
set -e
cd "$(dirname "$0")/.."
CACHE=${SALT_BENCH_CACHE:-/tmp/salt_bench_cache}
python3 - <<'PY'
import sys, os; sys.path.insert(0, '.')
from salt_amd import workload
cache = os.environ.get("SALT_BENCH_CACHE", "/tmp/salt_bench_cache")
w = workload.prepare("chr21", cache)
fq = os.path.join(w["dir"], "reads_4M.fq")
if not os.path.exists(fq):
    seqs, offs, _, _ = workload.make_reads(w["genome"], w["snp_pos"], w["snp_mask"], 4_000_000, 100, seed=5)
    workload.write_fastq(fq, seqs, offs)
PY
P=$CACHE/salt_chr21_g40000000_s190000_k21
nproc
for t in 8 16 32 64; do
  ./salt_amd/bin/salt -d -c -t $t $P/idx $P/reads_4M.fq 2> gpurun_out/e2e4_t$t.log > /dev/null
  tail -2 gpurun_out/e2e4_t$t.log
done
