"""Developer check of the GPU BDPT path vs the cpu_bdpt-estimator oracle."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import path_tracing_amd as hpt
from path_tracing_amd import scene_io as S
import oracle
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def cmp(name, a, b):
    d = np.abs(a - b)
    print("%-46s rmse %.3e maxabs %.3e differing px %d/%d mean %.6f %.6f" % (name, np.sqrt((d**2).mean()), d.max(), int((d.max(axis=-1) > 0).sum()), d.shape[0]*d.shape[1], a.mean(), b.mean()), flush=True)
for name, W, H, spp, spl in (("input", 64, 64, 4, 8), ("input", 40, 24, 2, 3), ("mis_test", 48, 48, 4, 8)):
    sc = S.load_scene(os.path.join(here, "tests/golden/scenes/%s.txt" % name))
    L, sp, tr = S.flatten_for_pt(sc)
    order = oracle.object_order(sc)
    t0 = time.time(); ref, st = oracle.bdpt_render(L, sp, tr, order, sc.eye, sc.look_at, sc.view_up, sc.fov, W, H, 4, 4, spp, spl, seed=5); t1 = time.time()
    cam = S.make_camera(sc.eye, sc.look_at, sc.view_up, sc.fov, W, H, tan_in_float=True)
    with hpt.Scene(L, sp, tr) as scene:
        scene.set_groups(*order)
        img = scene.render_bdpt(cam, W, H, 4, 4, spp, spl, hpt.make_params(seed=5))
        st2 = scene.stats()
        cmp("%s %dx%dx%d spl %d grouped" % (name, W, H, spp, spl), img, ref)
    order1 = oracle.object_order(None, sp, tr)
    ref1, _ = oracle.bdpt_render(L, sp, tr, order1, sc.eye, sc.look_at, sc.view_up, sc.fov, W, H, 4, 4, spp, spl, seed=5)
    with hpt.Scene(L, sp, tr) as scene:
        img1 = scene.render_bdpt(cam, W, H, 4, 4, spp, spl, hpt.make_params(seed=5))
        cmp("   one group (spheres then triangles)", img1, ref1)
    print("   oracle %.2fs; gpu device ms %.2f; grouped vs single-group oracle differ px %d" % (t1 - t0, st2["ms_total"], int((np.abs(ref - ref1).max(axis=-1) > 0).sum())))
