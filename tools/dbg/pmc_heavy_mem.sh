#!/bin/bash
# what k_heavy waits for at 8 and 12 one-wave blocks per CU: L1 -> L2 read latency, translation misses, address-unit stalls, scalar memory
R=$PWD; OUT=$R/gpurun_out/heavy_mem; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
Q="--workload grch38 --no-cpu --no-counters --e2e-reads 0 --streams 1 --steps 4 --warmup 2 --batches 4 --mode se"
for n in 8 12; do
 for pass in "a1:TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES GRBM_GUI_ACTIVE" "a2:TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_REQUEST TCP_TOTAL_CACHE_ACCESSES TCP_CACHE_MISS" "a3:TA_ADDR_STALLED_BY_TC_CYCLES TA_TA_BUSY TCP_TCR_TCP_STALL_CYCLES TCP_TCP_LATENCY" "b:SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" "c:TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS TCP_UTCL1_STALL_INFLIGHT_MAX TCP_UTCL1_STALL_MULTI_MISS TCP_READ_TAGCONFLICT_STALL_CYCLES"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  echo "pass $n $name" >> $OUT/progress.txt
  SALT_GPU_HEAVY_PER_CU=$n timeout -k 10 150 rocprofv3 --pmc $ctrs --output-format csv -d $OUT/p -o pmc -- python3 $R/bench.py $Q > /dev/null 2>> $OUT/log.txt || echo "pass $n $name FAILED"
  f=$(find $OUT/p -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then (head -1 $f; grep "salt::k_heavy(" $f) > $OUT/pmc_${n}_$name.csv; fi; rm -rf $OUT/p
 done
done
python3 - $OUT <<'P'
import csv, sys, collections, glob, os
for n in (8, 12):
    m={}
    for f in glob.glob(os.path.join(sys.argv[1], 'pmc_%d_*.csv'%n)):
        d=collections.defaultdict(list)
        for r in csv.DictReader(open(f)): d[r['Counter_Name']].append(float(r['Counter_Value']))
        m.update({k: sum(v)/len(v) for k,v in d.items()})
    print("== %d blocks per CU"%n)
    for k in sorted(m): print("   %-44s %.4g"%(k, m[k]))
    if m.get('TCP_TCC_READ_REQ'): print("   L1->L2 read latency %.0f cycles"%(m['TCP_TCC_READ_REQ_LATENCY']/m['TCP_TCC_READ_REQ']))
P
