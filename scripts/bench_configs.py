"""Timings of every BASELINE.json configuration on one GPU (bench.py is the contract line for configs[2]).

One JSON object on stdout: per configuration the device time of a render (median of `reps`), Msamples/s, the
per-kernel-class times of the last render (HIP events around every launch) and, for the PT configurations, the work
counts and the two rooflines bench.py reports (algorithmic GB/s against HBM peak; useful lane-operations against the
VALU peak).  AB_ONLY=cfg1,cfg4b,... restricts the run (profiling)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import path_tracing_amd as hpt
from path_tracing_amd import scene_io as S
import bench

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ONLY = set(filter(None, os.environ.get("AB_ONLY", "").split(",")))
out = {}


def timed(scene, fn, reps=3):
    ms = []
    for _ in range(reps + 1):
        fn(hpt.FLAG_TIME_KERNELS)
        st = scene.stats()
        ms.append(st["ms_total"])
    return float(np.median(ms[1:])), st


def kernel_classes(st, bdpt):
    if bdpt:
        return {"extend_ms": st["ms_extend"], "vertex_ms": st["ms_shade"], "connect_ms": st["ms_connect"], "other_ms (light trace, reduce, resolve)": st["ms_other"]}
    return {"trace_first_ms": st["ms_extend"] + st["ms_connect"], "trace_resume_ms": st["ms_resume"], "shade_ms": st["ms_shade"], "other_ms": st["ms_other"]}


def pt_config(name, L, sp, tr, cam, W, H, spp, reps=3):
    if ONLY and name not in ONLY:
        return
    with hpt.Scene(L, sp, tr) as scene:
        st0 = scene.stats()
        ms, st = timed(scene, lambda f: scene.render_pt(cam, W, H, 4, spp, hpt.make_params(seed=1, flags=f)), reps)
        scene.render_pt(cam, W, H, 4, min(spp, 16), hpt.make_params(seed=1, flags=hpt.FLAG_COUNT_WORK))
        wc = scene.stats()
    k = spp / min(spp, 16)                       # the counting render traced min(spp, 16) samples per pixel
    boxes = (wc["boxes_closest"] + wc["boxes_shadow"]) * k; tris = (wc["tris_closest"] + wc["tris_shadow"]) * k
    rays = (wc["closest_rays"] + wc["shadow_rays"]) * k
    trace_s = (st["ms_extend"] + st["ms_connect"] + st["ms_resume"]) * 1e-3
    bytes_ = 32.0 * boxes / 2 + 36.0 * tris + 44.0 * wc["closest_rays"] * k + 36.0 * wc["shadow_rays"] * k
    out[name] = {"ms": ms, "Msamples_per_s": W * H * spp / ms / 1e3, "triangles": len(tr), "kernels_last_render": kernel_classes(st, False),
                 "rays_per_sample": rays / (W * H * spp), "nodes_per_ray": boxes / 2 / max(rays, 1), "tris_per_ray": tris / max(rays, 1),
                 "bvh_depth": st0["bvh_depth"], "ms_bvh_build": st0["ms_bvh_build"],
                 "roofline_trace_hbm": {"achieved_GBps": bytes_ / trace_s / 1e9, "peak_GBps": bench.HBM_PEAK_GBS, "frac": bytes_ / trace_s / 1e9 / bench.HBM_PEAK_GBS},
                 "roofline_trace_valu": {"achieved_T_lane_ops": (bench.LANE_OPS_PER_BOX * boxes + bench.LANE_OPS_PER_TRI * tris) / trace_s / 1e12,
                                         "peak_T_lane_ops": bench.VALU_PEAK_LANE_OPS / 1e12,
                                         "frac": (bench.LANE_OPS_PER_BOX * boxes + bench.LANE_OPS_PER_TRI * tris) / trace_s / bench.VALU_PEAK_LANE_OPS}}


def bdpt_config(name, scene_file, W, H, spp, spl, reps=2):
    if ONLY and name not in ONLY:
        return
    sc = S.load_scene(os.path.join(HERE, "tests/golden/scenes", scene_file))
    L, sp, tr = S.flatten_for_pt(sc)
    cam = S.make_camera(sc.eye, sc.look_at, sc.view_up, sc.fov, W, H, tan_in_float=True)
    with hpt.Scene(L, sp, tr) as scene:
        scene.set_groups(*S.object_order(sc))
        ms, st = timed(scene, lambda f: scene.render_bdpt(cam, W, H, 4, 4, spp, spl, hpt.make_params(seed=1, flags=f)), reps)
        cspp = min(spp, 4)                       # the counting render traces min(spp, 4) samples per pixel
        scene.render_bdpt(cam, W, H, 4, 4, cspp, spl, hpt.make_params(seed=1, flags=hpt.FLAG_COUNT_WORK))
        wc = scene.stats()
    n_lv = len(L) * spl * 4
    k = spp / cspp
    connect_s = st["ms_connect"] * 1e-3
    # algorithmic bytes of the connection stage (what replaces the reference loop src/cpu_bdpt.cpp:387-440): a 16-B table entry per
    # surviving pair + an 8-B validity word per 64 candidates + per shadow ray its two end points (24 B) and what its walk reads:
    # 64-B nodes, 36-B triangles, 16-B spheres, 24-B group boxes
    cbytes = k * (16.0 * wc["bd_survivors"] + 8.0 * wc["bd_pairs"] / 64.0 + 24.0 * wc["bd_shadow_rays"] + 64.0 * wc["bd_nodes"]
                  + 36.0 * wc["bd_tris"] + 16.0 * wc["bd_spheres"] + 24.0 * wc["bd_group_boxes"])
    # useful lane-operations: per node visit two slab tests, per triangle / sphere test, per surviving pair two BSDF values (~2 x 120),
    # per unoccluded pair the MIS weight (two pdfs, ~2 x 90, and the ratio sums)
    lane_ops = k * (2 * bench.LANE_OPS_PER_BOX * wc["bd_nodes"] + bench.LANE_OPS_PER_TRI * wc["bd_tris"] + 25.0 * wc["bd_spheres"]
                    + 12.0 * wc["bd_group_boxes"] + 240.0 * wc["bd_survivors"] + 220.0 * wc["bd_unoccluded"] + 30.0 * wc["bd_pairs"])
    pmc = None
    try:
        cand = json.load(open(os.path.join(HERE, "profiles", "r03_bdpt_pmc.json")))
        if cand.get("kernel_source_sha") == bench.kernel_source_sha():
            kc = cand["kernels"].get("k_bdpt_connect", {})
            pmc = {"file": "profiles/r03_bdpt_pmc.json", "valu_issue_utilization": kc.get("valu_utilization"), "active_lanes": kc.get("valu_active_lanes_avg"),
                   "note": "summed over the profiled BDPT renders (all three configurations)"}
    except Exception:
        pmc = None
    out[name] = {"ms": ms, "Msamples_per_s": W * H * spp / ms / 1e3, "light_vertices": n_lv,
                 "connections_per_s": W * H * spp * 4.0 * n_lv / ms / 1e3 if n_lv else None,
                 "kernels_last_render": kernel_classes(st, True),
                 "connect_work_per_render": {"candidate_pairs": wc["bd_pairs"] * k, "pairs_after_culls": wc["bd_survivors"] * k,
                                             "shadow_rays": wc["bd_shadow_rays"] * k, "unoccluded": wc["bd_unoccluded"] * k,
                                             "nodes_per_shadow_ray": wc["bd_nodes"] / max(wc["bd_shadow_rays"], 1),
                                             "tris_per_shadow_ray": wc["bd_tris"] / max(wc["bd_shadow_rays"], 1),
                                             "spheres_per_shadow_ray": wc["bd_spheres"] / max(wc["bd_shadow_rays"], 1)},
                 "roofline_connect": {"kernel": "k_bdpt_connect", "bound": "hbm", "binding_limit": "valu issue (culls, two BSDF values, MIS weight and the shadow walk per pair)",
                                      "shadow_rays_per_s": wc["bd_shadow_rays"] * k / connect_s if connect_s > 0 else None,
                                      "candidate_pairs_per_s": wc["bd_pairs"] * k / connect_s if connect_s > 0 else None,
                                      "algorithmic_bytes_per_render": cbytes, "achieved_GBps": cbytes / connect_s / 1e9 if connect_s > 0 else None,
                                      "peak_GBps": bench.HBM_PEAK_GBS, "frac": cbytes / connect_s / 1e9 / bench.HBM_PEAK_GBS if connect_s > 0 else None,
                                      "valu_useful": {"achieved_T_lane_ops": lane_ops / connect_s / 1e12 if connect_s > 0 else None,
                                                      "peak_T_lane_ops": bench.VALU_PEAK_LANE_OPS / 1e12,
                                                      "frac": lane_ops / connect_s / bench.VALU_PEAK_LANE_OPS if connect_s > 0 else None},
                                      "pmc": pmc},
                 "note": "connections_per_s counts every (eye vertex, light vertex) pair of up to 4 eye vertices per sample; "
                         "the connect kernel is VALU-bound (profiles/: SQ pass), its table traffic is streamed by k_bdpt_reduce"}


cam = lambda W, H: S.make_camera(S.CORNELL_EYE, S.CORNELL_LOOK, S.CORNELL_UP, 50.0, W, H)
bdpt_config("cfg1_bdpt_input_256x256_4spp_spl8", "input.txt", 256, 256, 4, 8, reps=3)
pt_config("cfg2_pt_cornell_diffuse_512x512_64spp", *S.cornell_diffuse(), cam(512, 512), 512, 512, 64)
pt_config("cfg3_pt_100k_1024x1024_256spp", *S.cornell_with_sphere(100_000), cam(1024, 1024), 1024, 1024, 256)
bdpt_config("cfg4_bdpt_mis_test_1024x1024_64spp_spl8", "mis_test.txt", 1024, 1024, 64, 8)
bdpt_config("cfg4b_bdpt_input_1024x1024_8spp_spl8", "input.txt", 1024, 1024, 8, 8)
pt_config("cfg5_pt_1Mtri_4096x4096_4spp_1gpu", *S.cornell_with_sphere(1_000_000), cam(4096, 4096), 4096, 4096, 4, reps=2)
print(json.dumps(out, indent=1))
