#!/bin/bash
# Round profile of the bench workload.  Usage on the GPU box:  bash scripts/profile_round.sh <tag>
#   trace / trace_single   rocprofv3 --kernel-trace --stats of `bench.py --steps 2 --warmup 1` (two pipelines / one)
#   pass_*                 one 128-spp pass (`--steps 1 --warmup 0 --spp 128 --single-pipeline`, nothing else rendered:
#                          no counting render, no exclusive step), one PMC counter set per run -- counters are never
#                          combined with anything but --kernel-trace
# Results land in gpurun_out/<tag>/; scripts/profile_summarize.py <tag> turns them into profiles/<tag>_*.
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
TAG="${1:-r02}"
OUT="$ROOT/gpurun_out/$TAG"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-exclusive-step > "$OUT/trace.log" 2> "$OUT/trace.err"
echo "trace exit $?"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_single" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --single-pipeline --no-count-step > "$OUT/trace_single.log" 2> "$OUT/trace_single.err"
echo "trace (single pipeline) exit $?"
ONE="--steps 1 --warmup 0 --spp 128 --no-cpu-baseline --single-pipeline --no-count-step"
rocprofv3 --kernel-trace --output-format csv -d "$OUT/pass_timeline" -- python3 "$ROOT/bench.py" $ONE > "$OUT/pass_timeline.log" 2> "$OUT/pass_timeline.err"
echo "pass timeline exit $?"
pmc () {
  name="$1"; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -- python3 "$ROOT/bench.py" $ONE > "$OUT/pmc_$name.log" 2> "$OUT/pmc_$name.err"
  echo "pmc $name exit $?"
}
pmc FETCH_SIZE FETCH_SIZE
pmc WRITE_SIZE WRITE_SIZE
pmc TCC TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pmc SQ SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE
pmc SQ2 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD
# texture-addresser counters: at most two per pass (a longer list in one pass is refused with "error code 38: Request
# exceeds the capabilities of the hardware to collect" -- round 1 mistook that for a crash)
pmc TA1 TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum
pmc TA2 TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum
