"""Experiment: do two independent render pipelines on two HIP streams fill each other's idle issue slots?"""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import path_tracing_amd as hpt
from path_tracing_amd import scene_io as S
L, sp, tr = S.cornell_with_sphere(100000)
W = H = 1024
cam = S.make_camera(S.CORNELL_EYE, S.CORNELL_LOOK, S.CORNELL_UP, 50.0, W, H)
spass = int(os.environ.get("OV_SPASS", "64"))
NP = int(os.environ.get('OV_N', '2'))
scenes = [hpt.Scene(L, sp, tr) for _ in range(NP)]
streams = [torch.cuda.Stream() for _ in range(NP)]
n_local = hpt.local_pixels(W, H, hpt.make_params(seed=1))
bufs = [torch.zeros((n_local, 3), dtype=torch.float32, device="cuda") for _ in range(NP)]
def run(i, spp, off):
    p = hpt.make_params(seed=1, sample_offset=off, samples_per_pass=spass)
    scenes[i].render_pt_device(cam, W, H, 4, spp, p, bufs[i].data_ptr(), streams[i].cuda_stream)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    run(0, 256, 0)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    th = [threading.Thread(target=run, args=(i, 256 // NP, (256 // NP) * i)) for i in range(NP)]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("one pipeline 256 spp: %.1f ms | %d pipelines x %d spp: %.1f ms" % ((t1 - t0) * 1e3, NP, 256 // NP, (t2 - t1) * 1e3), flush=True)
