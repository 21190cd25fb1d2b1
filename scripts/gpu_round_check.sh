set -x
mkdir -p gpurun_out/r03a
python -m pytest tests -m gpu -x -q > gpurun_out/r03a/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03a/pytest.log
tail -5 gpurun_out/r03a/pytest.log
python bench.py > gpurun_out/r03a/bench.json 2> gpurun_out/r03a/bench.err; echo "bench rc $?"
AB_VARIANTS=0 AB_ROUNDS=3 python scripts/ab_tuning.py > gpurun_out/r03a/ab_sphere100k.log 2>&1
AB_VARIANTS=0 AB_ROUNDS=3 AB_SCENE=random python scripts/ab_tuning.py > gpurun_out/r03a/ab_random100k.log 2>&1
AB_VARIANTS=0 AB_ROUNDS=3 AB_TRIS=1000000 AB_SIZE=4096 AB_SPP=4 python scripts/ab_tuning.py > gpurun_out/r03a/ab_sphere1m.log 2>&1
tail -2 gpurun_out/r03a/ab_*.log
