R=$PWD; OUT=$R/gpurun_out/pk_pmc; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
Q="--workload grch38 --no-cpu --no-counters --e2e-reads 0 --mode pe --streams 1 --steps 4 --warmup 2 --batches 4 --pe-steps 4 --pe-batches 2"
for pass in "sq:SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS GRBM_GUI_ACTIVE" "sq2:SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "sq3:SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU" "sq4:SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  rocprofv3 --pmc $ctrs --output-format csv -d $OUT/$name -o pmc -- python3 $R/bench.py $Q > /dev/null 2>> $OUT/log.txt || echo "pass $name FAILED"
  f=$(find $OUT/$name -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then (head -1 $f; grep "k_sw" $f) > $OUT/pmc_$name.csv; fi
  rm -rf $OUT/$name
  echo "pass $name done"
done
