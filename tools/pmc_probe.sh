#!/bin/bash
# Extra PMC passes for kernel analysis (each group in its own run); summary -> gpurun_out/probe/pmc_summary.json
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/probe
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --pmc $grp --output-format csv -d "$OUT/p$i" -o pmc -- python3 "$ROOT/bench.py" --no-cpu --streams 1 --steps 3 --warmup 2 > /dev/null 2>> "$OUT/log.txt" || echo "group $i failed: $grp"
  find "$OUT/p$i" -name '*counter_collection.csv' -exec cp {} "$OUT/pmc_${i}_counter_collection.csv" \;
  rm -rf "$OUT/p$i"
  echo "[probe] group $i done: $grp"
done <<'GROUPS'
TA_BUSY_avr TA_TOTAL_WAVEFRONTS_sum
TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum
TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum
TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum
TCC_EA0_RDREQ_32B_sum TCC_REQ_sum
SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA
SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES
SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_LDS
GROUPS
python3 "$ROOT/tools/pmc_summary.py" "$OUT/pmc_summary.json" chr21 "$OUT"/pmc_*_counter_collection.csv > /dev/null
python3 - "$OUT/pmc_summary.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))["per_kernel_mean"]
for k in ("k_pack", "k_seed", "k_light", "k_heavy", "k_gap", "k_gapfin", "k_cigar"):
    if k in d:
        print(k, {c: (round(v, 1) if v < 1e4 else int(v)) for c, v in sorted(d[k].items())})
PY
