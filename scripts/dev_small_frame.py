"""Latency of small progressive frames (the reference GUI renders a few spp per frame, src/main.cpp:416)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import path_tracing_amd as hpt
from path_tracing_amd import scene_io as S
sc = S.load_scene(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/scenes/input.txt"))
L, sp, tr = S.flatten_for_pt(sc)
for (W, H, spp) in ((800, 600, 1), (800, 600, 4), (256, 256, 1)):
    cam = S.camera_for(sc, W, H)
    with hpt.Scene(L, sp, tr) as scene:
        ts = []; dev = []
        for f in range(30):
            t0 = time.perf_counter()
            scene.render_pt(cam, W, H, 4, spp, hpt.make_params(seed=1, sample_offset=f * spp))
            ts.append((time.perf_counter() - t0) * 1e3); dev.append(scene.stats()["ms_total"])
        print("%dx%d %d spp: wall %.3f ms per frame (device %.3f ms)" % (W, H, spp, float(np.median(ts[5:])), float(np.median(dev[5:]))), flush=True)
    t0 = time.perf_counter()
    for f in range(10): hpt.pt_render_wrapper(L, sp, tr, cam, W, H, 4, spp, seed=f)
    print("   one-shot wrapper (cached scene): %.3f ms per call" % ((time.perf_counter() - t0) * 100), flush=True)
hpt.wrapper_cache_clear()
