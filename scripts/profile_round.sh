#!/bin/bash
# Round profile: kernel-trace stats of the bench command + FETCH_SIZE / WRITE_SIZE passes (separate runs).
# Usage on the GPU box:  bash scripts/profile_round.sh <tag>
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
TAG="${1:-r01}"
OUT="$ROOT/gpurun_out/$TAG"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-exclusive-step > "$OUT/trace.log" 2>&1
echo "trace exit $?"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_single" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --single-pipeline > "$OUT/trace_single.log" 2>&1
echo "trace (single pipeline) exit $?"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pmc_$C" -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --spp 64 --no-cpu-baseline --single-pipeline > "$OUT/pmc_$C.log" 2>&1
  echo "pmc $C exit $?"
done
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d "$OUT/pmc_TCC" -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --spp 64 --no-cpu-baseline --single-pipeline > "$OUT/pmc_TCC.log" 2>&1
echo "pmc TCC exit $?"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_SQ" -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --spp 64 --no-cpu-baseline --single-pipeline > "$OUT/pmc_SQ.log" 2>&1
echo "pmc SQ exit $?"
