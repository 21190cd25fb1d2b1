"""Per-rank render time of the config-3 image split over G virtual ranks (load balance of the tile map)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import path_tracing_amd as hpt
from path_tracing_amd import scene_io as S
G = int(os.environ.get("G", "8")); spp = int(os.environ.get("SPP", "256"))
L, sp, tr = S.cornell_with_sphere(100000)
W = H = 1024
cam = S.make_camera(S.CORNELL_EYE, S.CORNELL_LOOK, S.CORNELL_UP, 50.0, W, H)
scene = hpt.Scene(L, sp, tr)
n_local = hpt.local_pixels(W, H, hpt.make_params(seed=1, rank=0, world=G))
buf = torch.zeros((n_local, 3), dtype=torch.float32, device="cuda")
ms = []
for rep in range(2):
    ms = []
    for r in range(G):
        p = hpt.make_params(seed=1, rank=r, world=G, flags=int(os.environ.get('FLAGS', '0')), samples_per_pass=int(os.environ.get('SPASS', '0')))
        scene.render_pt_device(cam, W, H, 4, spp, p, buf.data_ptr(), torch.cuda.current_stream().cuda_stream)
        ms.append(scene.stats()["ms_total"])
ms = np.array(ms)
print("G=%d per-rank ms: %s | mean %.2f max %.2f -> imbalance %.1f%% | sum %.1f" % (G, np.round(ms, 2), ms.mean(), ms.max(), 100 * (ms.max() / ms.mean() - 1), ms.sum()))
