#!/bin/bash
# kernel-trace stats of one short bench run (no CPU baseline); prints the per-kernel table.
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
SPP="${1:-64}"
OUT="$ROOT/gpurun_out/tq"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --spp "$SPP" --no-cpu-baseline > "$OUT/trace.log" 2>&1
echo "trace exit $?"
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    short = n.split("(")[0][-60:]
    print("%-62s calls %5s total %9.3f ms avg %9.1f us  %5.1f%%" % (short, r["Calls"], int(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
