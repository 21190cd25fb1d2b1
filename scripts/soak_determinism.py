#!/usr/bin/env python3
"""Soak: the same renders over and over, every image compared with the first of its kind (a race between waves, pipelines
or launches shows as a differing image sooner or later; the parity tests see each configuration once or twice).
usage: python scripts/soak_determinism.py [rounds=40]     (GPU; prints one line per case, exit code 1 on a difference)"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import path_tracing_amd as hpt
from path_tracing_amd import scene_io as S

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
golden = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "scenes")
bad = 0


def soak(name, render):
    global bad
    t0 = time.time(); first = None; diffs = 0
    for r in range(rounds):
        h = hashlib.sha256(np.ascontiguousarray(render(r)).tobytes()).hexdigest()
        if first is None: first = h
        diffs += h != first
    bad += diffs
    print("%-58s %3d renders, %d differ, %.1f s" % (name, rounds, diffs, time.time() - t0), flush=True)


cam = lambda W, H: S.make_camera(S.CORNELL_EYE, S.CORNELL_LOOK, S.CORNELL_UP, 50.0, W, H)
with hpt.Scene(*S.cornell_with_sphere(100_000)) as sc:
    soak("PT 100k sphere 512^2 x 32 spp, two pipelines", lambda r: sc.render_pt(cam(512, 512), 512, 512, 4, 32, hpt.make_params(seed=3)))
    soak("PT same, one pipeline", lambda r: sc.render_pt(cam(512, 512), 512, 512, 4, 32, hpt.make_params(seed=3, flags=hpt.FLAG_SINGLE_PIPELINE)))
    soak("PT same, HPT_FLAG_NO_HOST_WAIT", lambda r: sc.render_pt(cam(512, 512), 512, 512, 4, 32, hpt.make_params(seed=3, flags=hpt.FLAG_NO_HOST_WAIT)))
    soak("PT same, small passes (4 spp each) and 8-pixel tiles", lambda r: sc.render_pt(cam(512, 512), 512, 512, 4, 32, hpt.make_params(seed=3, samples_per_pass=4, tile=8)))
with hpt.Scene(*S.cornell_random_triangles(20_000)) as sc:
    soak("PT 20k random triangles 256^2 x 8 spp (deep stacks)", lambda r: sc.render_pt(cam(256, 256), 256, 256, 4, 8, hpt.make_params(seed=5)))
for name in ("input", "mis_test"):
    d = S.load_scene(os.path.join(golden, name + ".txt"))
    L, sp, tr = S.flatten_for_pt(d)
    c = S.make_camera(d.eye, d.look_at, d.view_up, d.fov, 256, 256, tan_in_float=True)
    with hpt.Scene(L, sp, tr) as sc:
        sc.set_groups(*S.object_order(d))
        soak("BDPT %s 256^2 x 4 spp, spl 8" % name, lambda r: sc.render_bdpt(c, 256, 256, 4, 4, 4, 8, hpt.make_params(seed=8)))
        soak("BDPT %s same, HPT_FLAG_NO_HOST_WAIT" % name, lambda r: sc.render_bdpt(c, 256, 256, 4, 4, 4, 8, hpt.make_params(seed=8, flags=hpt.FLAG_NO_HOST_WAIT)))
        soak("PT %s 256^2 x 16 spp" % name, lambda r: sc.render_pt(c, 256, 256, 4, 16, hpt.make_params(seed=8)))
print("soak: %s" % ("OK" if bad == 0 else "%d differing renders" % bad))
sys.exit(1 if bad else 0)
