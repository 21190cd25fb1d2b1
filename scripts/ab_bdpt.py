"""A/B of builds on the bidirectional path: device ms of config 1 and of input.txt at 1024^2 x 8 spp (median of 5), per library."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, os, json
sys.path.insert(0, %r)
import numpy as np, path_tracing_amd as hpt
from path_tracing_amd import scene_io as S
sc = S.load_scene(os.path.join(%r, "tests/golden/scenes/input.txt"))
L, sp, tr = S.flatten_for_pt(sc)
out = {}
with hpt.Scene(L, sp, tr) as scene:
    scene.set_groups(*S.object_order(sc))
    for name, W, spp in (("cfg1_256", 256, 4), ("input_1024", 1024, 8)):
        cam = S.make_camera(sc.eye, sc.look_at, sc.view_up, sc.fov, W, W, tan_in_float=True)
        ms = []
        for r in range(6):
            img = scene.render_bdpt(cam, W, W, 4, 4, spp, 8, hpt.make_params(seed=1, flags=hpt.FLAG_TIME_KERNELS))
            st = scene.stats(); ms.append((st["ms_total"], st["ms_connect"]))
        m = np.median(np.array(ms[1:]), axis=0)
        out[name] = [round(float(m[0]), 3), round(float(m[1]), 3), float(img.mean())]
print(json.dumps(out))
''' % (ROOT, ROOT)
for lib in sys.argv[1:] or ["default"]:
    env = dict(os.environ)
    if lib != "default": env["HPT_LIBRARY"] = os.path.join(ROOT, "path_tracing_amd", "csrc", "libhpt_%s.so" % lib)
    r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
    print("%-8s %s" % (lib, (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1]), flush=True)
