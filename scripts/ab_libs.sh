#!/bin/bash
# A/B of two builds of libhpt.so (see `make variant`): runs scripts/ab_tuning.py with each, alternating, three rounds.
# usage: bash scripts/ab_libs.sh <variant name> [env assignments for ab_tuning.py ...]
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
V="$1"; shift
for r in 1 2 3; do
  echo "== round $r: default build"; env "$@" AB_VARIANTS=0 AB_ROUNDS=4 python3 "$ROOT/scripts/ab_tuning.py" | tail -1
  echo "== round $r: variant $V";    env "$@" AB_VARIANTS=0 AB_ROUNDS=4 HPT_LIBRARY="$ROOT/path_tracing_amd/csrc/libhpt_$V.so" python3 "$ROOT/scripts/ab_tuning.py" | tail -1
done
