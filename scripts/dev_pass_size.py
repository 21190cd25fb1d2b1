"""Config 3 render time against the pass size (two pipelines in flight)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import path_tracing_amd as hpt
from path_tracing_amd import scene_io as S
L, sp, tr = S.cornell_with_sphere(100000)
W = H = 1024
cam = S.make_camera(S.CORNELL_EYE, S.CORNELL_LOOK, S.CORNELL_UP, 50.0, W, H)
scene = hpt.Scene(L, sp, tr)
n_local = hpt.local_pixels(W, H, hpt.make_params(seed=1))
buf = torch.zeros((n_local, 3), dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for spass in [int(x) for x in os.environ.get("SPASS", "32,64,128").split(",")]:
    ts = []
    for rep in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        scene.render_pt_device(cam, W, H, 4, 256, hpt.make_params(seed=1, samples_per_pass=spass, flags=int(os.environ.get('FLAGS', '0'))), buf.data_ptr(), st)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("samples_per_pass %d: %.1f ms (min %.1f)" % (spass, float(np.median(ts[1:])), min(ts[1:])), flush=True)
