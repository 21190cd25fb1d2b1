V="0:0,0:0x4500000,0:0x5500000,0:0x6500000,0:0x4700000,0:0x5700000,0:0x6700000,0:0x4900000,0:0x5900000,0:0x6900000"
for lib in libhpt.so libhpt_w8.so; do
echo "== $lib sphere 100k (refill 16/24/32 x node_min 12/16/24)"; HPT_LIBRARY=path_tracing_amd/csrc/$lib AB_VARIANTS=$V AB_ROUNDS=3 python scripts/ab_tuning.py 2>&1 | tail -10
echo "== $lib sphere 1M"; HPT_LIBRARY=path_tracing_amd/csrc/$lib AB_VARIANTS=$V AB_ROUNDS=3 AB_TRIS=1000000 AB_SIZE=4096 AB_SPP=4 python scripts/ab_tuning.py 2>&1 | tail -10
done
echo "== random, budget 6 forced"; AB_VARIANTS="0:0,12:0,12:0x100000,12:0x5700000" AB_SCENE=random AB_SPP=16 python scripts/ab_tuning.py 2>&1 | tail -4
