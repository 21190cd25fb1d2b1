#!/usr/bin/env python3
"""Reproducer of the hipcc SLP-vectorizer miscompile that csrc/Makefile holds off with -fno-slp-vectorize.

The product source sets the hit-level words of a next-event record (local wo, Lambda(wo), throughput, material, shadow
segment start) BEFORE the nested branches behind the unit-ball rejection loop of k_shade.  This script writes a copy of
pt_kernels.hip with those five assignments moved back INSIDE both branches (the form that was miscompiled), builds that
copy twice -- HIPFLAGS as they are, and without -fno-slp-vectorize -- and renders tests/golden/scenes/input.txt with both
libraries against the committed golden image.  A miscompiled build loses the y component of the float3 values the
vectorizer paired into packed-f32 operations: green = 0 in the next-event contributions.

  python scripts/micro/slp_repro.py build     (no GPU needed: writes build_slp_repro/libhpt_slp_{on,off}.so)
  python scripts/micro/slp_repro.py run       (on the GPU box)
"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "path_tracing_amd", "csrc")
OUT = os.path.join(CSRC, "build_slp_repro")
HOIST = """                    n_wo = ctx.wo; n_lam = pre.lam_o; n_thr = throughput; n_mat = mat_idx;
                    s_p1 = pos + normal * kEps;
"""
INNER = "n_wo = ctx.wo; n_lam = pre.lam_o; n_thr = throughput; n_mat = mat_idx; s_p1 = pos + normal * kEps; "


def make_source():
    src = open(os.path.join(CSRC, "pt_kernels.hip")).read()
    assert src.count(HOIST) == 1, "k_shade no longer has the hoisted assignments in the expected form"
    src = src.replace(HOIST, "")
    a = "                                nee = true;\n                                n_wi = to_local(light_dir, ctx.T, ctx.B, ctx.N);"
    b = "                                    nee = true;\n                                    n_pdf_light = pdf_light_area * dist2 / fmaxf(cos_light, 1e-6f);"
    assert src.count(a) == 1 and src.count(b) == 1
    src = src.replace(a, "                                " + INNER + "\n" + a).replace(b, "                                    " + INNER + "\n" + b)
    os.makedirs(OUT, exist_ok=True)
    open(os.path.join(OUT, "pt_kernels.hip"), "w").write(src)


def build():
    make_source()
    mk = open(os.path.join(CSRC, "Makefile")).read()
    flags = re.search(r"^HIPFLAGS \?= (.*)$", mk, re.M).group(1).replace("$(ARCH)", "gfx950")
    print(subprocess.run(["/opt/rocm/bin/hipcc", "--version"], capture_output=True, text=True).stdout.splitlines()[0])
    for tag, fl in (("off", flags), ("on", flags.replace("-fno-slp-vectorize", ""))):
        objs = []
        for f in ("pt_kernels.hip", "bdpt_kernels.hip", "hpt_api.cpp", "hpt_multi.cpp"):
            path = os.path.join(OUT if f == "pt_kernels.hip" else CSRC, f)
            o = os.path.join(OUT, "%s_%s.o" % (os.path.splitext(f)[0], tag))
            subprocess.check_call(["/opt/rocm/bin/hipcc"] + fl.split() + ["-I", CSRC, "-x", "hip", "-c", "-o", o, path])
            objs.append(o)
        objs.append(os.path.join(CSRC, "scene_build.o"))
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-o", os.path.join(OUT, "libhpt_slp_%s.so" % tag)] + objs + ["-ldl"])
        print("built libhpt_slp_%s.so (SLP vectorizer %s)" % (tag, tag))


def run():
    code = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np
import path_tracing_amd as hpt
from path_tracing_amd import scene_io as S
g = np.load(os.path.join(%r, "tests", "golden", "pt_input_64x64_d4_8spp_seed7.npz"))
sc = S.load_scene(os.path.join(%r, "tests", "golden", "scenes", "input.txt"))
L, sp, tr = S.flatten_for_pt(sc)
cam = S.make_camera(sc.eye, sc.look_at, sc.view_up, 50.0, 64, 64)
with hpt.Scene(L, sp, tr) as scene:
    img = scene.render_pt(cam, 64, 64, 4, 8, hpt.make_params(seed=7))
ref = g["image"]
bad = int((np.abs(img - ref).max(axis=2) > 1e-6).sum())
print("pixels that differ from the golden image: %%d of %%d; channel means %%s (golden %%s)" %% (bad, 64 * 64, img.mean(axis=(0, 1)).round(4), ref.mean(axis=(0, 1)).round(4)))
''' % (ROOT, ROOT, ROOT)
    for tag in ("off", "on"):
        env = dict(os.environ, HPT_LIBRARY=os.path.join(OUT, "libhpt_slp_%s.so" % tag))
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        print("SLP vectorizer %-3s: %s" % (tag, (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1]))


if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1] if len(sys.argv) > 1 else "build"]()
