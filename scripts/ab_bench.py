"""A/B timing of render variants on the cfg3 workload, interleaved rounds in one process."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import path_tracing_amd as hpt
from path_tracing_amd import scene_io as S

ntri = int(os.environ.get("AB_TRIS", "100000"))
spp = int(os.environ.get("AB_SPP", "32"))
W = H = int(os.environ.get("AB_SIZE", "1024"))
rounds = int(os.environ.get("AB_ROUNDS", "5"))
variants = [int(v, 0) for v in os.environ.get("AB_VARIANTS", "0").split(",")]
L, sp, tr = S.cornell_with_sphere(ntri) if os.environ.get("AB_SCENE", "sphere") == "sphere" else S.cornell_random_triangles(ntri)
cam = S.make_camera(S.CORNELL_EYE, S.CORNELL_LOOK, S.CORNELL_UP, 50.0, W, H)
scene = hpt.Scene(L, sp, tr)
ref = None
res = {v: [] for v in variants}
kern = {}
extra = {}
for r in range(rounds + 1):
    for v in variants:
        p = hpt.make_params(seed=1, flags=hpt.FLAG_TIME_KERNELS, samples_per_pass=int(os.environ.get("AB_SPASS", "0")))
        p.reserved = v
        img = scene.render_pt(cam, W, H, 4, spp, p)
        st = scene.stats()
        if ref is None: ref = img
        assert np.array_equal(img, ref), "variant %d changes the image" % v
        if r > 0:
            res[v].append(st["ms_total"]); kern[v] = (st["ms_extend"] + st["ms_resume"], st["ms_shade"], st["ms_connect"], st["ms_other"], st["n_extend"]); extra[v] = (st["ms_resume"], st["long_rays_last_pass"] / max(st["traced_rays_last_pass"], 1))
for v in variants:
    a = np.array(res[v])
    print("variant %#x: median %.2f ms  min %.2f  -> %.1f Msamples/s | extend %.1f shade %.1f connect %.1f other %.1f (n_ext %d)" % (
        v, np.median(a), a.min(), W * H * spp / np.median(a) / 1e3, *kern[v]) + " resume %.1f ms, long rays %.3f" % extra[v], flush=True)
p = hpt.make_params(seed=1, flags=hpt.FLAG_COUNT_WORK); p.reserved = variants[0]
scene.render_pt(cam, W, H, 4, spp, p)
st = scene.stats()
print("SIMD efficiency closest %.3f shadow %.3f | boxes/ray closest %.1f shadow %.1f | tris/ray closest %.2f shadow %.2f | steps/ray closest %.1f shadow %.1f" % (
    st["lane_steps_closest"] / max(st["wave_steps_closest"], 1), st["lane_steps_shadow"] / max(st["wave_steps_shadow"], 1),
    st["boxes_closest"] / st["closest_rays"], st["boxes_shadow"] / max(st["shadow_rays"], 1),
    st["tris_closest"] / st["closest_rays"], st["tris_shadow"] / max(st["shadow_rays"], 1),
    st["lane_steps_closest"] / st["closest_rays"], st["lane_steps_shadow"] / max(st["shadow_rays"], 1)))
print("leaf trips: lane/ray closest %.2f shadow %.2f ; leaf SIMD eff closest %.3f shadow %.3f ; node wave-trips per 64 rays closest %.1f ; leaf wave-trips per 64 rays closest %.1f" % (
    st["leaf_lane_closest"] / st["closest_rays"], st["leaf_lane_shadow"] / max(st["shadow_rays"], 1),
    st["leaf_lane_closest"] / max(st["leaf_wave_closest"], 1), st["leaf_lane_shadow"] / max(st["leaf_wave_shadow"], 1),
    st["wave_steps_closest"] / st["closest_rays"], st["leaf_wave_closest"] / st["closest_rays"]))
print({k: st[k] for k in ("samples", "closest_rays", "shadow_rays", "path_iters", "bvh_nodes", "bvh_depth")})
