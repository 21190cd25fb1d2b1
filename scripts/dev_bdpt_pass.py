"""BDPT pass-size sweep on input.txt, 1024x1024, 8 spp."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import path_tracing_amd as hpt
from path_tracing_amd import scene_io as S
import oracle
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sc = S.load_scene(os.path.join(root, "tests/golden/scenes/input.txt"))
L, sp, tr = S.flatten_for_pt(sc); order = oracle.object_order(sc)
cam = S.make_camera(sc.eye, sc.look_at, sc.view_up, sc.fov, 1024, 1024, tan_in_float=True)
scene = hpt.Scene(L, sp, tr); scene.set_groups(*order)
ref = None
for spass in (0, 1, 2, 4, 8):
    for i in range(2):
        img = scene.render_bdpt(cam, 1024, 1024, 4, 4, 8, 8, hpt.make_params(seed=1, samples_per_pass=spass, flags=hpt.FLAG_TIME_KERNELS))
        st = scene.stats()
    if ref is None: ref = img
    assert np.array_equal(img, ref)
    print("samples_per_pass %d: %.1f ms (connect %.1f, other %.1f)" % (spass, st["ms_total"], st["ms_connect"], st["ms_other"]), flush=True)
