#!/bin/bash
# BDPT evidence: rocprofv3 --kernel-trace --stats and one SQ counter pass of the BDPT configurations
# (config 1 = input.txt 256^2 x 4 spp, input.txt at 1024^2 x 8 spp, config 4 = mis_test.txt 1024^2 x 64 spp).
# Usage on the GPU box: bash scripts/profile_bdpt.sh <tag>; then python scripts/profile_summarize_bdpt.py <tag>
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
TAG="${1:-r03_bdpt}"
OUT="$ROOT/gpurun_out/$TAG"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export AB_ONLY=cfg1_bdpt_input_256x256_4spp_spl8,cfg4_bdpt_mis_test_1024x1024_64spp_spl8,cfg4b_bdpt_input_1024x1024_8spp_spl8
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/scripts/bench_configs.py" > "$OUT/trace.log" 2> "$OUT/trace.err"
echo "trace exit $?"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_SQ" -- python3 "$ROOT/scripts/bench_configs.py" > "$OUT/pmc_SQ.log" 2> "$OUT/pmc_SQ.err"
echo "pmc SQ exit $?"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_FETCH_SIZE" -- python3 "$ROOT/scripts/bench_configs.py" > "$OUT/pmc_FETCH_SIZE.log" 2> "$OUT/pmc_FETCH_SIZE.err"
echo "pmc FETCH exit $?"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_WRITE_SIZE" -- python3 "$ROOT/scripts/bench_configs.py" > "$OUT/pmc_WRITE_SIZE.log" 2> "$OUT/pmc_WRITE_SIZE.err"
echo "pmc WRITE exit $?"
