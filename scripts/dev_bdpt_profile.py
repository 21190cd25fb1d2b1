import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np
import path_tracing_amd as hpt
from path_tracing_amd import scene_io as S
import oracle
sc = S.load_scene('/root/repo/tests/golden/scenes/input.txt')
L, sp, tr = S.flatten_for_pt(sc); order = oracle.object_order(sc)
cam = S.make_camera(sc.eye, sc.look_at, sc.view_up, sc.fov, 1024, 1024, tan_in_float=True)
scene = hpt.Scene(L, sp, tr); scene.set_groups(*order)
for i in range(2):
    scene.render_bdpt(cam, 1024, 1024, 4, 4, 4, 8, hpt.make_params(seed=1, flags=hpt.FLAG_TIME_KERNELS))
    st = scene.stats()
print({k: round(st[k], 2) if isinstance(st[k], float) else st[k] for k in ("ms_total", "ms_extend", "ms_shade", "ms_connect", "ms_other", "n_extend", "n_connect")})
