#!/bin/bash
# Collects rocprofv3 PMC counters for the bench workload in separate passes (counters only, with
# --kernel-trace; never combined with sys/hip/hsa tracing).  Run on the GPU box via gpurun:
#   bash scripts/pmc_profile.sh [spp]
# Results: gpurun_out/pmc/<pass>/..._counter_collection.csv ; summarise with scripts/pmc_summarize.py
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
SPP="${1:-16}"
OUT="$ROOT/gpurun_out/pmc"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run_pass () {
  name="$1"; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- \
    python3 "$ROOT/bench.py" --steps 1 --warmup 0 --spp "$SPP" --no-cpu-baseline > "$OUT/$name.log" 2>&1
  echo "pass $name exit $?"
}
run_pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run_pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
run_pass sq3 SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INST_LEVEL_VMEM SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH GRBM_GUI_ACTIVE
run_pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
run_pass fetch FETCH_SIZE
run_pass write WRITE_SIZE
ls -R "$OUT" | head -40
