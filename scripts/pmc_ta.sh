#!/bin/bash
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/pmc_ta"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run_pass () {
  name="$1"; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- \
    python3 "$ROOT/bench.py" --steps 1 --warmup 0 --spp 16 --no-cpu-baseline > "$OUT/$name.log" 2>&1
  echo "pass $name exit $?"
}
run_pass ta1 TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE GRBM_TA_BUSY
run_pass ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum
run_pass tcp1 TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum
run_pass tcp2 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum
python3 - "$OUT" <<'PY'
import csv, glob, sys, os, collections
root=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(os.path.join(root,'*','*','*_counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name']
        k='k_trace' if 'k_trace' in n else ('k_shade' if 'k_shade' in n else None)
        if k: agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
for k in agg:
    print(k, {c: '%.4g'%v for c,v in sorted(agg[k].items())})
PY
