#!/bin/bash
# SQ pass (VALU/SALU instruction counts, active lanes, cycles) per kernel on the BDPT dev workload.
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/pmc_sq_bdpt"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE --output-format csv -d "$OUT/sq" -- python3 "$ROOT/scripts/dev_bdpt_profile.py" > "$OUT/sq.log" 2>&1
echo "exit $?"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, re
f = glob.glob(sys.argv[1] + "/sq/*/*_counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    m = re.search(r"(k_\w+(?:<[^>]*>)?)", n)
    k = m.group(1) if m else n[:28]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
for k, a in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0)):
    cyc = a.get("GRBM_GUI_ACTIVE", 0) / 8.0
    v = a.get("SQ_INSTS_VALU", 0)
    print("%-28s disp %3d  VALU %.3e  SALU %.3e  lanes %.1f  cycles %.3e (%.2f ms @2.4GHz)  VALU util %.2f" % (
        k, len(disp[k]), v, a.get("SQ_INSTS_SALU", 0), a.get("SQ_THREAD_CYCLES_VALU", 0) / max(v, 1), cyc, cyc / 2.4e6, v / max(cyc * 256, 1)))
PY
