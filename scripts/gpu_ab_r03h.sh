export AB_VARIANTS="0:0"
echo "== single pipeline"; python scripts/ab_matrix.py default s5 s6 2>&1 | tail -4
echo "== two pipelines, 256 spp"; AB_DUAL=1 AB_SPP=256 python scripts/ab_matrix.py default s5 s6 2>&1 | tail -4
