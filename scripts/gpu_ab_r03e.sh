V="0:0,0:0x100000,0:0x300000,0:0x700000,0:0x900000,0:0xb00000,0:0x2100000,0:0x3100000,0:0x4100000,0:0x5100000,0:0x8100000,0:0x10100000"
echo "== sphere 100k"; AB_VARIANTS=$V AB_ROUNDS=3 python scripts/ab_tuning.py 2>&1 | tail -13
echo "== sphere 1M"; AB_VARIANTS=$V AB_ROUNDS=3 AB_TRIS=1000000 AB_SIZE=4096 AB_SPP=4 python scripts/ab_tuning.py 2>&1 | tail -13
echo "== w8 lib"; AB_VARIANTS="0:0,0:0x100000" AB_ROUNDS=3 HPT_LIBRARY=path_tracing_amd/csrc/libhpt_w8.so python scripts/ab_tuning.py 2>&1 | tail -2
AB_VARIANTS="0:0,0:0x100000" AB_ROUNDS=3 AB_TRIS=1000000 AB_SIZE=4096 AB_SPP=4 HPT_LIBRARY=path_tracing_amd/csrc/libhpt_w8.so python scripts/ab_tuning.py 2>&1 | tail -2
