#!/bin/bash
# Two SQ passes only (instruction counts, busy/wait, lane utilisation); see pmc_profile.sh.
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
SPP="${1:-16}"
OUT="$ROOT/gpurun_out/pmc"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run_pass () {
  name="$1"; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- \
    python3 "$ROOT/bench.py" --steps 1 --warmup 0 --spp "$SPP" --no-cpu-baseline > "$OUT/$name.log" 2>&1
  echo "pass $name exit $?"
}
run_pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run_pass sq3 SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_BRANCH GRBM_GUI_ACTIVE
python3 "$ROOT/scripts/pmc_summarize.py" "$OUT"
