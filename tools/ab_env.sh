#!/bin/bash
# A/B runs of bench.py under environment settings (GPU box): tools/ab_env.sh "A=1 B=2" "C=3" ...  -> one summary line per setting
for s in "$@"; do
  env $s timeout -k 10 400 python bench.py --e2e-reads 0 --cpu-sample 20000 > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err || { echo "$s: FAILED"; tail -3 gpurun_out/ab_tmp.err; continue; }
  python - "$s" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab_tmp.json").read().strip().splitlines()[-1])
print("%-40s value %.1f Mreads/s  step %.3f ms  serialized %s  parity %s" % (sys.argv[1] or "defaults", d["value"], d["ms_per_step"], d.get("kernel_ms_serialized"), d["parity"]["mismatching_reads"]))
PY
done
