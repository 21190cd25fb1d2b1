O=gpurun_out/r03d; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "resume_stack or split" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
AB_VARIANTS="0:0,0:0x40000,0:0x100000" python scripts/ab_tuning.py 2>&1 | tail -4
AB_VARIANTS="0:0,0:0x40000,0:0x100000" AB_TRIS=1000000 AB_SIZE=4096 AB_SPP=4 python scripts/ab_tuning.py 2>&1 | tail -4
AB_VARIANTS="0:0,12:0,12:0x100000" AB_SCENE=random AB_SPP=16 python scripts/ab_tuning.py 2>&1 | tail -4
