export AB_VARIANTS="0:0"
echo "== single pipeline 64 spp: default = cooperative, bp = plain loop at the hoisted place, bl = cooperative with the old LDS sizes, prev = before"; ABM_ROUNDS=3 python scripts/ab_matrix.py default bp bl prev 2>&1 | tail -5
echo "== two pipelines, 256 spp"; AB_DUAL=1 AB_SPP=256 ABM_ROUNDS=3 python scripts/ab_matrix.py default bp bl prev 2>&1 | tail -5
