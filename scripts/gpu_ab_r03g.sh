python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "resume_launch or split" 2>&1 | tail -2
V="0:0,0:0x20000000,4:0x20000000,8:0x20000000"
echo "== sphere 100k"; AB_VARIANTS=$V AB_ROUNDS=3 python scripts/ab_tuning.py 2>&1 | tail -4
echo "== sphere 1M"; AB_VARIANTS=$V AB_ROUNDS=3 AB_TRIS=1000000 AB_SIZE=4096 AB_SPP=4 python scripts/ab_tuning.py 2>&1 | tail -4
echo "== random"; AB_VARIANTS="0:0,12:0,0:0x20000000,6:0x20000000" AB_SCENE=random AB_SPP=16 python scripts/ab_tuning.py 2>&1 | tail -4
echo "== input.txt-like small scene (cornell 36 tris)"; AB_VARIANTS="0:0,0:0x20000000" AB_TRIS=30 AB_SIZE=512 python scripts/ab_tuning.py 2>&1 | tail -2
