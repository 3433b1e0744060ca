#!/bin/bash
# A/B runs of bench.py under extra arguments (GPU box): tools/ab_args.sh "--streams 6" "--streams 8 --batches 16" ...
for s in "$@"; do
  timeout -k 10 400 python bench.py --e2e-reads 0 --cpu-sample 20000 $s > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err || { echo "$s: FAILED"; tail -3 gpurun_out/ab_tmp.err; continue; }
  python - "$s" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab_tmp.json").read().strip().splitlines()[-1])
print("%-40s value %.1f Mreads/s  step %.3f ms  parity %s" % (sys.argv[1] or "defaults", d["value"], d["ms_per_step"], d["parity"]["mismatching_reads"]))
PY
done
