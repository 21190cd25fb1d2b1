V="0:0,0:0x20000000,0:0x40000000"
echo "== 2 / 3 / 4 pipelines: config 3 (256 spp)"; AB_DUAL=1 AB_SPP=256 AB_VARIANTS=$V AB_ROUNDS=4 python scripts/ab_tuning.py 2>/dev/null | tail -3
echo "== config 5 shape (4 spp)"; AB_DUAL=1 AB_VARIANTS=$V AB_ROUNDS=4 AB_TRIS=1000000 AB_SIZE=4096 AB_SPP=4 python scripts/ab_tuning.py 2>/dev/null | tail -3
echo "== config 5 shape (16 spp)"; AB_DUAL=1 AB_VARIANTS=$V AB_ROUNDS=3 AB_TRIS=1000000 AB_SIZE=4096 AB_SPP=16 python scripts/ab_tuning.py 2>/dev/null | tail -3
echo "== config 2 (36 tris, 512^2, 64 spp)"; AB_DUAL=1 AB_VARIANTS=$V AB_ROUNDS=4 AB_TRIS=30 AB_SIZE=512 AB_SPP=64 python scripts/ab_tuning.py 2>/dev/null | tail -3
echo "== rank share: 1024^2 32 spp"; AB_DUAL=1 AB_VARIANTS=$V AB_ROUNDS=4 AB_SPP=32 python scripts/ab_tuning.py 2>/dev/null | tail -3
