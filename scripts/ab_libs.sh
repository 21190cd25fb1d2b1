#!/bin/bash
# A/B of two builds of libhpt.so (see `make variant`): runs scripts/ab_tuning.py with each, four rounds, the order
# swapped from round to round (the second process of a pair runs ~0.8 % faster than the first whatever it loads -- two
# copies of one build measured 132.1 against 130.9 ms -- so a fixed order favours whichever build comes second).
# usage: bash scripts/ab_libs.sh <variant name> [env assignments for ab_tuning.py ...]
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
V="$1"; shift
run_default () { echo "== round $1: default build"; env "${@:2}" AB_VARIANTS=0 AB_ROUNDS=4 python3 "$ROOT/scripts/ab_tuning.py" | tail -1; }
run_variant () { echo "== round $1: variant $V"; env "${@:2}" AB_VARIANTS=0 AB_ROUNDS=4 HPT_LIBRARY="$ROOT/path_tracing_amd/csrc/libhpt_$V.so" python3 "$ROOT/scripts/ab_tuning.py" | tail -1; }
for r in 1 2 3 4; do
  if [ $((r % 2)) -eq 1 ]; then run_default $r "$@"; run_variant $r "$@"; else run_variant $r "$@"; run_default $r "$@"; fi
done
