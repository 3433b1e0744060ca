#!/bin/bash
# end-to-end (PCIe + host I/O inclusive) rate of the C++ CLI on a workload.py config (default chr21) and a byte compare of
# its SAM with the CPU oracle's; run on the GPU box:  tools/e2e_cli.sh [config]
set -e
cd "$(dirname "$0")/.."
CFG=${1:-chr21}
export SALT_BENCH_CACHE=${SALT_BENCH_CACHE:-/tmp/salt_bench_cache}
read -r P FQ < <(python3 - "$CFG" <<'PY'
import sys, os; sys.path.insert(0, '.')
from salt_amd import workload
w = workload.prepare(sys.argv[1], os.environ["SALT_BENCH_CACHE"])
fq = os.path.join(w["dir"], "reads_1M.fq")
if not os.path.exists(fq):
    seqs, offs, _, _ = workload.make_reads(w["genome"], w["snp_pos"], w["snp_mask"], 1_000_000, 100, seed=1)
    workload.write_fastq(fq, seqs, offs)
print(w["prefix"], fq)
PY
)
echo "index $P reads $FQ"
for t in 1 16; do
  ./salt_amd/bin/salt -d -c -t $t $P $FQ 2> gpurun_out/e2e_${CFG}_t$t.log > /tmp/e2e.sam
  tail -1 gpurun_out/e2e_${CFG}_t$t.log
done
./oracle/salt_oracle -d -c -t 16 $P $FQ 2> gpurun_out/e2e_${CFG}_oracle.log | grep -v '^@PG' > /tmp/ora.sam
tail -1 gpurun_out/e2e_${CFG}_oracle.log
grep -v '^@PG' /tmp/e2e.sam | cmp - /tmp/ora.sam && echo "E2E SAM identical to the CPU oracle on 1M reads ($CFG)"
grep -c -v '^@' /tmp/ora.sam; cut -f3 /tmp/ora.sam | grep -v '^@' | sort | uniq -c | sort -k1,1nr | head -12
