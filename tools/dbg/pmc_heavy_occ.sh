#!/bin/bash
# resident waves of k_heavy at 8 / 12 / 16 one-wave blocks per CU: SQ_WAVE_CYCLES (quad-cycles, all XCDs) x 4 / (GRBM_GUI_ACTIVE / 8)
R=$PWD; OUT=$R/gpurun_out/heavy_occ; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
Q="--workload grch38 --no-cpu --no-counters --e2e-reads 0 --streams 1 --steps 4 --warmup 2 --batches 4 --mode se"
for n in 8 12; do
  SALT_GPU_HEAVY_PER_CU=$n rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD --output-format csv -d $OUT/p$n -o pmc -- python3 $R/bench.py $Q > $OUT/bench_$n.json 2>> $OUT/log.txt || echo "pass $n FAILED"
  f=$(find $OUT/p$n -name '*counter_collection.csv' | head -1)
  (head -1 $f; grep "salt::k_heavy(" $f) > $OUT/pmc_$n.csv; rm -rf $OUT/p$n
  python3 - $OUT/pmc_$n.csv $n <<'P'
import csv, sys, collections
d=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])): d[r['Counter_Name']].append(float(r['Counter_Value']))
m={k: sum(v)/len(v) for k,v in d.items()}
print("per CU %s: waves launched %.0f, kernel %.0f cycles, mean resident waves %.0f" % (sys.argv[2], m['SQ_WAVES'], m['GRBM_GUI_ACTIVE']/8, m['SQ_WAVE_CYCLES']*4/(m['GRBM_GUI_ACTIVE']/8)))
P
done
