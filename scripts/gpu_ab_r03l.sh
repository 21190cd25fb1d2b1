python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
export AB_VARIANTS="0:0"
echo "== single pipeline 64 spp"; python scripts/ab_matrix.py default prev 2>&1 | tail -3
echo "== two pipelines, 256 spp"; AB_DUAL=1 AB_SPP=256 ABM_ROUNDS=4 python scripts/ab_matrix.py default prev 2>&1 | tail -3
echo "== 1M, 4 spp, two pipelines"; AB_DUAL=1 AB_TRIS=1000000 AB_SIZE=4096 AB_SPP=4 python scripts/ab_matrix.py default prev 2>&1 | tail -3
