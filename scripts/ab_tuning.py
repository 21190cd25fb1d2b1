"""A/B of launch-tuning bits (hpt_params.reserved) in one process, interleaved rounds: ms per pass, single pipeline.
AB_VARIANTS: comma-separated integers (0x.. allowed) OR-ed into `reserved`; the image must not change."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import path_tracing_amd as hpt
from path_tracing_amd import scene_io as S
W = H = int(os.environ.get("AB_SIZE", "1024")); spp = int(os.environ.get("AB_SPP", "64"))
kind = os.environ.get("AB_SCENE", "sphere"); ntri = int(os.environ.get("AB_TRIS", "100000"))
L, sp, tr = S.cornell_with_sphere(ntri) if kind == "sphere" else S.cornell_random_triangles(ntri)
cam = S.make_camera(S.CORNELL_EYE, S.CORNELL_LOOK, S.CORNELL_UP, 50.0, W, H)
# a variant is `reserved` or `reserved:extra_flags` (development switches live in flags bits 16-31)
variants = [tuple(int(x, 0) for x in (v.split(":") + ["0"])[:2]) for v in os.environ.get("AB_VARIANTS", "0,0x80").split(",")]
flags = hpt.FLAG_TIME_KERNELS | (0 if os.environ.get("AB_DUAL") else hpt.FLAG_SINGLE_PIPELINE)
res = {v: [] for v in variants}
ref = None
with hpt.Scene(L, sp, tr) as scene:
    for r in range(int(os.environ.get("AB_ROUNDS", "5")) + 1):
        for v in variants:
            p = hpt.make_params(seed=1, flags=flags | v[1], samples_per_pass=int(os.environ.get("AB_SPASS", "0"))); p.reserved = v[0]
            img = scene.render_pt(cam, W, H, 4, spp, p)
            st = scene.stats()
            if ref is None: ref = img
            assert np.array_equal(img, ref), "variant %s changes the image" % (v,)
            if r: res[v].append((st["ms_total"], st["ms_extend"] + st["ms_connect"], st["ms_resume"], st["ms_shade"], st["ms_other"]))
print("scene %s %d tris, %dx%d, %d spp, %s" % (kind, len(tr), W, H, spp, "two pipelines" if os.environ.get("AB_DUAL") else "single pipeline"))
for v in variants:
    m = np.median(np.array(res[v]), axis=0)
    print("variant %#6x:%#x render %.2f ms (min %.2f) | first %.2f resume %.2f shade %.2f other %.2f" % (v[0], v[1], m[0], np.array(res[v])[:, 0].min(), m[1], m[2], m[3], m[4]), flush=True)
