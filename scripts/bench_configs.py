"""Timings of the BASELINE.json configs other than the headline one (bench.py covers configs[2])."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import path_tracing_amd as hpt
from path_tracing_amd import scene_io as S
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {}
def timed(fn, reps=3):
    ms = []
    for _ in range(reps + 1):
        fn(); ms.append(scene.stats()["ms_total"])
    return float(np.median(ms[1:]))
# config 1: cpu_bdpt estimator on input.txt, 256x256, 4 spp
sc = S.load_scene(os.path.join(here, "tests/golden/scenes/input.txt"))
L, sp, tr = S.flatten_for_pt(sc); order = S.object_order(sc)
cam = S.make_camera(sc.eye, sc.look_at, sc.view_up, sc.fov, 256, 256, tan_in_float=True)
scene = hpt.Scene(L, sp, tr); scene.set_groups(*order)
ms = timed(lambda: scene.render_bdpt(cam, 256, 256, 4, 4, 4, 8, hpt.make_params(seed=1)))
out["cfg1_bdpt_input_256x256_4spp_spl8"] = {"ms": ms, "Msamples_per_s": 256 * 256 * 4 / ms / 1e3}
scene.close()
# config 2: PT, diffuse Cornell (36 triangles), 512x512, 64 spp
L2, sp2, tr2 = S.cornell_diffuse()
cam2 = S.make_camera(S.CORNELL_EYE, S.CORNELL_LOOK, S.CORNELL_UP, 50.0, 512, 512)
scene = hpt.Scene(L2, sp2, tr2)
ms = timed(lambda: scene.render_pt(cam2, 512, 512, 4, 64, hpt.make_params(seed=1)))
out["cfg2_pt_cornell_diffuse_512x512_64spp"] = {"ms": ms, "Msamples_per_s": 512 * 512 * 64 / ms / 1e3}
scene.close()
# config 4: BDPT on mis_test.txt, 1024x1024, 64 spp
sc4 = S.load_scene(os.path.join(here, "tests/golden/scenes/mis_test.txt"))
L4, sp4, tr4 = S.flatten_for_pt(sc4); order4 = S.object_order(sc4)
cam4 = S.make_camera(sc4.eye, sc4.look_at, sc4.view_up, sc4.fov, 1024, 1024, tan_in_float=True)
scene = hpt.Scene(L4, sp4, tr4); scene.set_groups(*order4)
ms = timed(lambda: scene.render_bdpt(cam4, 1024, 1024, 4, 4, 64, 8, hpt.make_params(seed=1)), reps=2)
out["cfg4_bdpt_mis_test_1024x1024_64spp_spl8"] = {"ms": ms, "Msamples_per_s": 1024 * 1024 * 64 / ms / 1e3}
# the same estimator on input.txt at config 4's size (a non-degenerate BDPT workload), 8 spp
scene.close()
cam1 = S.make_camera(sc.eye, sc.look_at, sc.view_up, sc.fov, 1024, 1024, tan_in_float=True)
scene = hpt.Scene(L, sp, tr); scene.set_groups(*order)
ms = timed(lambda: scene.render_bdpt(cam1, 1024, 1024, 4, 4, 8, 8, hpt.make_params(seed=1)), reps=2)
out["bdpt_input_1024x1024_8spp_spl8"] = {"ms": ms, "Msamples_per_s": 1024 * 1024 * 8 / ms / 1e3}
scene.close()
# config 5 shape on one GPU at reduced spp: 1M triangles, 4096x4096, 4 spp (PT)
L5, sp5, tr5 = S.cornell_with_sphere(1_000_000)
cam5 = S.make_camera(S.CORNELL_EYE, S.CORNELL_LOOK, S.CORNELL_UP, 50.0, 4096, 4096)
scene = hpt.Scene(L5, sp5, tr5)
st0 = scene.stats()
ms = timed(lambda: scene.render_pt(cam5, 4096, 4096, 4, 4, hpt.make_params(seed=1)), reps=2)
out["cfg5_pt_1Mtri_4096x4096_4spp_1gpu"] = {"ms": ms, "Msamples_per_s": 4096 * 4096 * 4 / ms / 1e3, "triangles": len(tr5),
                                            "ms_bvh_build": st0["ms_bvh_build"], "bvh_depth": st0["bvh_depth"]}
scene.close()
print(json.dumps(out, indent=1))
