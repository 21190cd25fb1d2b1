set -x
O=gpurun_out/r03b; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_bvh_walk.py -m gpu -x -q -k "resume_stack or split or work_counts or config3 or no_host_wait" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
export AB_VARIANTS="0:0,0:0x40000"
python scripts/ab_matrix.py default r02 w7 w8 > $O/m_sphere100k.log 2>&1; tail -9 $O/m_sphere100k.log
AB_TRIS=1000000 AB_SIZE=4096 AB_SPP=4 python scripts/ab_matrix.py default r02 w7 w8 > $O/m_sphere1m.log 2>&1; tail -9 $O/m_sphere1m.log
AB_SCENE=random ABM_ROUNDS=2 python scripts/ab_matrix.py default r02 > $O/m_random100k.log 2>&1; tail -5 $O/m_random100k.log
