#!/bin/bash
# sweep the second-launch (long ray) tuning of the split trace step; usage: tune_long.sh v1 v2 ...
for v in "$@"; do
  echo "HPT_TUNE_LONG=$v"
  HPT_TUNE_LONG=$v AB_SPP=${AB_SPP:-64} AB_ROUNDS=2 AB_VARIANTS=0 timeout -k 10 120 python scripts/ab_bench.py 2>&1 | grep "^variant"
done
