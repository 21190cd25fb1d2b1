# usage: bash scripts/gpu_quick.sh <out tag> <command...>   -- runs the command on the GPU box, output under gpurun_out/<tag>.log
TAG=$1; shift
mkdir -p gpurun_out
"$@" > gpurun_out/$TAG.log 2>&1
echo "rc $?" >> gpurun_out/$TAG.log
tail -40 gpurun_out/$TAG.log
