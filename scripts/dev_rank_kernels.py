import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import path_tracing_amd as hpt
from path_tracing_amd import scene_io as S
L, sp, tr = S.cornell_with_sphere(100000)
W = H = 1024
cam = S.make_camera(S.CORNELL_EYE, S.CORNELL_LOOK, S.CORNELL_UP, 50.0, W, H)
scene = hpt.Scene(L, sp, tr)
for G in (1, 8):
    n_local = hpt.local_pixels(W, H, hpt.make_params(seed=1, rank=0, world=G))
    buf = torch.zeros((n_local, 3), dtype=torch.float32, device="cuda")
    for rep in range(2):
        p = hpt.make_params(seed=1, rank=0, world=G, flags=hpt.FLAG_TIME_KERNELS | hpt.FLAG_SINGLE_PIPELINE)
        scene.render_pt_device(cam, W, H, 4, 256, p, buf.data_ptr(), torch.cuda.current_stream().cuda_stream)
        st = scene.stats()
    print("G=%d: total %.2f | first %.2f (n %d) resume %.2f shade %.2f tail-trace %.2f other %.2f | x G/8: %.2f" % (
        G, st["ms_total"], st["ms_extend"], st["n_extend"], st["ms_resume"], st["ms_shade"], st["ms_connect"], st["ms_other"], st["ms_total"] * G / 8))
