#!/usr/bin/env python3
"""Where the first call of a scene goes (reference: its timer includes upload and allocation, src/main_cli.cpp:207-219):
scene creation (flatten + BVH build on the host, upload, per-triangle frames) / first render (workspace allocation +
render) / a later render, for the config-3 scene and the config-5 shape."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import path_tracing_amd as hpt
from path_tracing_amd import scene_io as S
import torch
torch.cuda.init(); torch.zeros(1, device="cuda")
out = {}
for name, ntri, W, spp in (("cfg3_100k_1024x1024_256spp", 100_000, 1024, 256), ("cfg5_shape_1M_4096x4096_4spp", 1_000_000, 4096, 4)):
    L, sp, tr = S.cornell_with_sphere(ntri)
    cam = S.make_camera(S.CORNELL_EYE, S.CORNELL_LOOK, S.CORNELL_UP, 50.0, W, W)
    t0 = time.perf_counter()
    scene = hpt.Scene(L, sp, tr)
    t1 = time.perf_counter()
    scene.render_pt(cam, W, W, 4, spp, hpt.make_params(seed=1))
    t2 = time.perf_counter()
    scene.render_pt(cam, W, W, 4, spp, hpt.make_params(seed=1))
    t3 = time.perf_counter()
    st = scene.stats()
    out[name] = {"scene_create_ms": (t1 - t0) * 1e3, "of_which_bvh_build_ms": st["ms_bvh_build"], "of_which_upload_and_frames_ms": st["ms_upload"],
                 "first_render_ms": (t2 - t1) * 1e3, "later_render_ms": (t3 - t2) * 1e3, "device_ms_of_a_render": st["ms_total"],
                 "first_call_total_ms": (t2 - t0) * 1e3}
    scene.close()
print(json.dumps(out, indent=1))
