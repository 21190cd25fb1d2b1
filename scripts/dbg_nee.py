import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle, path_tracing_amd as hpt
from path_tracing_amd import scene_io as S
sc = S.load_scene("tests/golden/scenes/input.txt"); L, sp, tr = S.flatten_for_pt(sc)
W, H, spp = 96, 64, 1
cam = S.camera_for(sc, W, H)
L1 = L[:1]
ref, st = oracle.pt_render(L1, sp, tr, cam, W, H, 1, spp, seed=8)
with hpt.Scene(L1, sp, tr) as s:
    a = s.render_pt(cam, W, H, 1, spp, hpt.make_params(seed=8, flags=hpt.FLAG_COUNT_WORK))
    stg = s.stats()
d = np.abs(a - ref).max(axis=2)
print("diff px", int((d > 0).sum()), "of", W * H, "nonzero ref", int((ref.max(axis=2) > 0).sum()), "nonzero gpu", int((a.max(axis=2) > 0).sum()))
print("oracle rays", st["closest_rays"], st["shadow_rays"], "gpu", stg["closest_rays"], stg["shadow_rays"])
ys, xs = np.nonzero(d > 0)
for y, x in list(zip(ys, xs))[:12]:
    print((y, x), "gpu", a[y, x], "ref", ref[y, x], "ratio", a[y, x] / np.maximum(ref[y, x], 1e-30))
# where gpu has light but ref does not and vice versa
print("gpu>0 & ref==0:", int(((a.max(axis=2) > 0) & (ref.max(axis=2) == 0)).sum()), " ref>0 & gpu==0:", int(((a.max(axis=2) == 0) & (ref.max(axis=2) > 0)).sum()))
