export HPT_LIBRARY=$PWD/path_tracing_amd/csrc/libhpt_dev.so
mkdir -p gpurun_out/r03j
AB_LEAVES=2,1,3,4 python scripts/sweep_leaf.py 2>/dev/null > gpurun_out/r03j/leaf_100k.log
AB_LEAVES=2,1,3,4 AB_TRIS=1000000 AB_SIZE=4096 AB_SPP=4 python scripts/sweep_leaf.py 2>/dev/null > gpurun_out/r03j/leaf_1m.log
AB_LEAVES=2,1,3,4 AB_SCENE=random AB_SPP=16 python scripts/sweep_leaf.py 2>/dev/null > gpurun_out/r03j/leaf_random.log
cat gpurun_out/r03j/*.log
