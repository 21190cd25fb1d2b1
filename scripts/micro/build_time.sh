set -e
cd "$(dirname "$0")/../.."
g++ -O2 -std=c++17 -pthread -ffp-contract=off -DHPT_DEV_TUNING -I path_tracing_amd/csrc -o /tmp/build_time scripts/micro/build_time.cpp path_tracing_amd/csrc/scene_build.cpp
python3 - <<'PY'
import sys; sys.path.insert(0, '.')
import numpy as np
from path_tracing_amd import scene_io as S
for n in (100000, 1000000):
    L, sp, tr = S.cornell_with_sphere(n); np.ascontiguousarray(tr).tofile('/tmp/tris_%d.bin' % n)
PY
cat /sys/fs/cgroup/cpu.max 2>/dev/null || true
for t in ${BUILD_THREADS:-default}; do
  for n in 100000 1000000; do echo "== threads $t, $n triangles"; if [ $t = default ]; then /tmp/build_time /tmp/tris_$n.bin 2>&1 | tail -8; else HPT_BUILD_THREADS=$t /tmp/build_time /tmp/tris_$n.bin 2>&1 | tail -8; fi; done
done
