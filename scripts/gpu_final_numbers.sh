# the round's numbers on one box: bench line, per-config table, scaling projection, first call, host boundary
mkdir -p gpurun_out/r03_final
python bench.py > gpurun_out/r03_final/bench_line.json 2> gpurun_out/r03_final/bench.err; echo "bench rc $?"
python scripts/bench_configs.py > gpurun_out/r03_final/bench_configs.json 2> gpurun_out/r03_final/bench_configs.err; echo "configs rc $?"
python scripts/scaling_projection.py > gpurun_out/r03_final/scaling_projection.json 2> gpurun_out/r03_final/scaling.err; echo "scaling rc $?"
python scripts/first_call_breakdown.py > gpurun_out/r03_final/first_call.json 2> /dev/null; echo "first call rc $?"
python scripts/host_boundary_rate.py > gpurun_out/r03_final/host_boundary.json 2> /dev/null; echo "host boundary rc $?"
python bench.py --fanout --gpus 1 > gpurun_out/r03_final/bench_fanout_1.json 2> /dev/null; echo "fanout rc $?"
python bench.py --gpus 2 --backend gloo --no-cpu-baseline > gpurun_out/r03_final/bench_2ranks_gloo_one_gpu.json 2> /dev/null; echo "2 ranks rc $?"
