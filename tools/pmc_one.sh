#!/bin/bash
# one PMC pass of bench.py (serialized kernels) -> per-kernel means on stdout.  usage: tools/pmc_one.sh "<counters>" [bench args]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CTRS=$1; shift
OUT=$ROOT/gpurun_out/pmc_one; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CTRS --output-format csv -d "$OUT" -o pmc -- python3 "$ROOT/bench.py" --no-cpu --e2e-reads 0 --streams 1 --steps 4 --warmup 2 --batches 4 "$@" > /dev/null 2> "$OUT/err.log"
python3 - "$OUT" <<'PY'
import csv, glob, sys
from collections import defaultdict
per = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))
for fn in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(fn, newline="")):
        k = row["Kernel_Name"]
        if "salt::" not in k: continue
        k = k.split("salt::")[1].split("(")[0].split("<")[0]
        per[k][row["Counter_Name"]][int(row["Dispatch_Id"])] += float(row["Counter_Value"])
for k, c in per.items():
    print(k, {n: round(sum(list(v.values())[2:]) / max(len(v) - 2, 1)) for n, v in c.items()})
PY
