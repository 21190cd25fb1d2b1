"""A/B of the BVH leaf size (HPT_MAX_LEAF, read when a scene is built by a development build: `make variant VARIANT=dev EXTRA=-DHPT_DEV_TUNING`, HPT_LIBRARY=.../libhpt_dev.so) x node-step budget of the first trace launch:
ms per pass, single pipeline.  AB_SCENE: sphere (config 3 / 5 shape) | random (incoherent small triangles) | input"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import path_tracing_amd as hpt
from path_tracing_amd import scene_io as S
W = H = int(os.environ.get("AB_SIZE", "1024")); spp = int(os.environ.get("AB_SPP", "64"))
kind = os.environ.get("AB_SCENE", "sphere"); ntri = int(os.environ.get("AB_TRIS", "100000"))
if kind == "sphere": L, sp, tr = S.cornell_with_sphere(ntri)
elif kind == "random": L, sp, tr = S.cornell_random_triangles(ntri)
else:
    sc = S.load_scene(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/scenes/input.txt")); L, sp, tr = S.flatten_for_pt(sc)
cam = S.make_camera(S.CORNELL_EYE, S.CORNELL_LOOK, S.CORNELL_UP, 50.0, W, H)
leaves = [int(x) for x in os.environ.get("AB_LEAVES", "4,2,3").split(",")]
budgets = [int(x) for x in os.environ.get("AB_BUDGETS", "0").split(",")]
ref = None
print("scene %s %d tris, %dx%d, %d spp" % (kind, len(tr), W, H, spp))
for leaf in leaves:
    os.environ["HPT_MAX_LEAF"] = str(leaf)
    with hpt.Scene(L, sp, tr) as scene:
        for b in budgets:
            ms = []
            for r in range(4):
                p = hpt.make_params(seed=1, flags=hpt.FLAG_TIME_KERNELS | hpt.FLAG_SINGLE_PIPELINE); p.reserved = b << 1
                img = scene.render_pt(cam, W, H, 4, spp, p)
                st = scene.stats()
                if r: ms.append((st["ms_total"], st["ms_extend"] + st["ms_connect"], st["ms_resume"], st["ms_shade"]))
            if ref is None: ref = img
            assert np.array_equal(img, ref)
            m = np.median(np.array(ms), axis=0)
            print("leaf %d budget %2d: pass %.2f ms | first %.2f resume %.2f shade %.2f | nodes %d depth %d long %.3f" % (
                leaf, st["split_budget"], m[0], m[1], m[2], m[3], st["bvh_nodes"], st["bvh_depth"], st["long_rays_last_pass"] / max(st["traced_rays_last_pass"], 1)), flush=True)
