#!/bin/bash
# Regenerates the evidence under gpurun_out/<tag>/ on the GPU box (copy what is to be judged into profiles/<tag>/):
#   bench line, rocprofv3 kernel-trace stats of the same command, PMC passes (each in its own run), traffic + issue summary,
#   FETCH_SIZE calibration for the aligner's gather shapes (tools/ubench/gather).
# usage: tools/profile_round.sh <tag> <workload> [extra bench args]      SALT_PROFILE_MODE=pe: the same passes over the paired-end leg (bench.py --mode pe)
set -e
TAG=${1:-r03}; WL=${2:-grch38}; shift || true; shift || true
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
if [ "${SALT_PROFILE_MODE:-se}" != pe ]; then
python3 bench.py --workload "$WL" "$@" > "$OUT/bench_${WL}_1gpu.json" 2> "$OUT/bench.log"
tail -1 "$OUT/bench_${WL}_1gpu.json" | cut -c1-300
fi
cd /tmp && export TMPDIR=/tmp
MODE=${SALT_PROFILE_MODE:-se}
SUF=""; PRE=""; [ "$MODE" = pe ] && SUF="_pe" && PRE="pe_"
Q="--workload $WL --no-cpu --no-counters --e2e-reads 0 --mode $MODE"
# kernel-trace stats twice: --streams 1 (every kernel alone on the GPU: what the roofline block of the bench line is built from)
# and the default command (batches overlapping on 4 streams: what "kernel_ms" of the bench line shows)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o kt -- python3 "$ROOT/bench.py" $Q --streams 1 "$@" > "$OUT/bench_under_rocprofv3${SUF}_${WL}.json" 2>> "$OUT/bench.log"
find "$OUT/kt" -name '*kernel_stats.csv' -exec cp {} "$OUT/rocprofv3_kernel_stats${SUF}_${WL}.csv" \;
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt3" -o kt -- python3 "$ROOT/bench.py" $Q "$@" > "$OUT/bench_under_rocprofv3_streams${SUF}_${WL}.json" 2>> "$OUT/bench.log"
find "$OUT/kt3" -name '*kernel_stats.csv' -exec cp {} "$OUT/rocprofv3_kernel_stats_streams${SUF}_${WL}.csv" \;
rm -rf "$OUT/kt3" "$OUT/kt"
echo "[profile] kernel stats done"
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "tcc:TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "sq:SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS GRBM_GUI_ACTIVE" "sq2:SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  rocprofv3 --pmc $ctrs --output-format csv -d "$OUT/pmc_$name" -o pmc -- python3 "$ROOT/bench.py" $Q --streams 1 --steps 4 --warmup 2 --batches 4 --pe-steps 4 --pe-batches 2 > /dev/null 2>> "$OUT/bench.log" || echo "[profile] pmc pass $name FAILED"
  find "$OUT/pmc_$name" -name '*counter_collection.csv' -exec cp {} "$OUT/${PRE}pmc_${name}_${WL}.csv" \;
  # keep the product's kernels only (the generator's torch kernels are nine tenths of the rows)
  if [ -f "$OUT/${PRE}pmc_${name}_${WL}.csv" ]; then (head -1 "$OUT/${PRE}pmc_${name}_${WL}.csv"; grep "salt::" "$OUT/${PRE}pmc_${name}_${WL}.csv") > "$OUT/.f.csv" && mv "$OUT/.f.csv" "$OUT/${PRE}pmc_${name}_${WL}.csv"; fi
  rm -rf "$OUT/pmc_$name"
  echo "[profile] pmc pass $name done"
done
python3 "$ROOT/tools/pmc_summary.py" "$OUT/pmc_summary${SUF}_${WL}.json" "$WL" "$OUT/rocprofv3_kernel_stats${SUF}_${WL}.csv" "$OUT"/${PRE}pmc_*_${WL}.csv
# FETCH_SIZE per random load of 4 / 16 / 32 / 64 bytes (what one gather really moves)
if [ "$MODE" != pe ] && [ -x "$ROOT/tools/ubench/gather" ]; then
  "$ROOT/tools/ubench/gather" 16 256 > "$OUT/gather_rate.txt" 2>&1 || true
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_gather" -o pmc -- "$ROOT/tools/ubench/gather" 16 256 > /dev/null 2>> "$OUT/bench.log" || true
  find "$OUT/pmc_gather" -name '*counter_collection.csv' -exec cp {} "$OUT/pmc_gather_calibration.csv" \;
  rm -rf "$OUT/pmc_gather"
  python3 "$ROOT/tools/pmc_summary.py" --gather "$OUT/pmc_gather_calibration.csv" 268435456 >> "$OUT/gather_rate.txt" || true
  cat "$OUT/gather_rate.txt"
fi
