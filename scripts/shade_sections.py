#!/usr/bin/env python3
"""Where k_shade's lanes go: wave trips and active lanes per section of the kernel, from two development builds
(`make variant VARIANT=sp1 EXTRA=-DHPT_SHADE_PROFILE=1`, `... sp2 ... =2`, ... `sp5`) and a counting render of the config-3 scene."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, json
sys.path.insert(0, %r)
import path_tracing_amd as hpt
from path_tracing_amd import scene_io as S
L, sp, tr = S.cornell_with_sphere(100000)
cam = S.make_camera(S.CORNELL_EYE, S.CORNELL_LOOK, S.CORNELL_UP, 50.0, 1024, 1024)
with hpt.Scene(L, sp, tr) as scene:
    scene.render_pt(cam, 1024, 1024, 4, 16, hpt.make_params(seed=1, flags=hpt.FLAG_COUNT_WORK))
    st = scene.stats()
print(json.dumps({k: int(st[k]) for k in ("bd_pairs", "bd_survivors", "bd_shadow_rays", "bd_unoccluded", "bd_nodes", "bd_tris", "bd_spheres", "bd_group_boxes", "path_iters")}))
''' % ROOT
names = {1: ["a: queue entry (hit record, path state)", "b: light hit (emission rule)", "c: surface hit (material, frame, Lambda)", "d: one try of the unit-ball rejection loop"],
         2: ["e: next-event geometry behind the loop", "f: direction sampling (bsdf_sample)", "g: throughput update, continuation", "h: staged next-event evaluation (dense)"],
         3: ["f1: bsdf_sample, smooth dielectric branch", "f2: bsdf_sample, mirror branch", "f3: bsdf_sample, specular lobe (visible-normal sample)", "f4: bsdf_sample, value + pdf of the sampled direction"],
         4: ["v1: bsdf_eval_pdf past its early returns (both callers)", "v2: ... the value (G, Fresnel, specular term)", "v3: ... dielectric Fresnel", "v4: ... the pdf"],
         5: ["g1: delta continuation", "g2: non-delta continuation (throughput, cosine / pdf)", "g3: path goes on (state written back)", "h1: next-event contribution kept (shadow record written)"]}
out = {}
for n in (1, 2, 3, 4, 5):
    env = dict(os.environ, HPT_LIBRARY=os.path.join(ROOT, "path_tracing_amd", "csrc", "libhpt_sp%d.so" % n))
    r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
    st = json.loads(r.stdout.strip().splitlines()[-1])
    f = ["bd_pairs", "bd_survivors", "bd_shadow_rays", "bd_unoccluded", "bd_nodes", "bd_tris", "bd_spheres", "bd_group_boxes"]
    trips = st["path_iters"] / 64.0
    for i, nm in enumerate(names[n]):
        w, l = st[f[2 * i]], st[f[2 * i + 1]]
        out[nm] = {"wave_entries": w, "entries_per_64_path_iterations": w / trips, "active_lanes_avg": l / max(w, 1)}
print(json.dumps(out, indent=1))
