#!/bin/bash
# Regenerates the evidence under gpurun_out/<tag>/ on the GPU box (copy what is to be judged into profiles/<tag>/):
#   bench line, rocprofv3 kernel-trace stats of the same command, PMC passes (each in its own run), traffic summary.
# usage: tools/profile_round.sh <tag> [bench args]
set -e
TAG=${1:-r01}; shift || true
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
python3 bench.py "$@" > "$OUT/bench_chr21_1gpu.json" 2> "$OUT/bench.log"
tail -1 "$OUT/bench_chr21_1gpu.json" | cut -c1-400
cd /tmp && export TMPDIR=/tmp
# kernel-trace stats twice: --streams 1 (every kernel alone on the GPU: what the roofline block of the bench line is built from)
# and the default command (batches overlapping on 4 streams: what "kernel_ms" / roofline.timed_region of the bench line show)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o kt -- python3 "$ROOT/bench.py" --no-cpu --streams 1 "$@" > "$OUT/bench_under_rocprofv3.json" 2>> "$OUT/bench.log"
cp "$OUT"/kt/*kernel_stats.csv "$OUT/rocprofv3_kernel_stats.csv" 2>/dev/null || find "$OUT/kt" -name '*kernel_stats.csv' -exec cp {} "$OUT/rocprofv3_kernel_stats.csv" \;
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt3" -o kt -- python3 "$ROOT/bench.py" --no-cpu "$@" > "$OUT/bench_under_rocprofv3_streams.json" 2>> "$OUT/bench.log"
find "$OUT/kt3" -name '*kernel_stats.csv' -exec cp {} "$OUT/rocprofv3_kernel_stats_streams.csv" \;
rm -rf "$OUT/kt3"
echo "[profile] kernel stats done"
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "tcc:TCC_HIT_sum TCC_MISS_sum" "sq:SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "sq2:SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  rocprofv3 --pmc $ctrs --output-format csv -d "$OUT/pmc_$name" -o pmc -- python3 "$ROOT/bench.py" --no-cpu --streams 1 --steps 3 --warmup 2 > /dev/null 2>> "$OUT/bench.log"
  find "$OUT/pmc_$name" -name '*counter_collection.csv' -exec cp {} "$OUT/pmc_${name}_counter_collection.csv" \;
  echo "[profile] pmc pass $name done"
done
python3 "$ROOT/tools/pmc_summary.py" "$OUT/pmc_summary.json" chr21 "$OUT"/pmc_*_counter_collection.csv
rm -rf "$OUT"/kt "$OUT"/pmc_fetch "$OUT"/pmc_write "$OUT"/pmc_tcc "$OUT"/pmc_sq "$OUT"/pmc_sq2
