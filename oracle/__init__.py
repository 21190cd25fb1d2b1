"""TEST INFRASTRUCTURE ONLY -- ctypes front-end of the CPU oracle (oracle/liboracle.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
package.  PARITY STATUS: parity unpinned (see oracle/ref_math.hpp).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class PtOpts(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("rng_mode", C.c_int), ("trig_mode", C.c_int),
                ("sample_offset", C.c_int), ("x0", C.c_int), ("y0", C.c_int), ("x1", C.c_int), ("y1", C.c_int),
                ("threads", C.c_int), ("glass_shadow_opaque", C.c_int), ("max_delta", C.c_int),
                ("output_sum", C.c_int), ("russian_roulette", C.c_int), ("ball_draw_reversed", C.c_int)]


class PtStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("samples", "closest_rays", "shadow_rays", "bounces",
                                          "delta_bounces", "tri_tests", "sphere_tests",
                                          "boxes_closest", "tris_closest", "boxes_shadow", "tris_shadow")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class Bvh(C.Structure):
    """The tree libhpt.so exports (include/hpt.h, hpt_bvh_info): handed to pt_render(bvh=...) as a dict."""
    _fields_ = [("qnodes", C.c_void_p), ("num_nodes", C.c_int), ("tris", C.c_void_p), ("num_tris", C.c_int),
                ("num_rounds", C.c_int), ("qorigin", C.c_float * 3), ("qscale", C.c_float * 3)]


def build(force: bool = False) -> str:
    """Compiles oracle/liboracle.so with the committed Makefile (g++, -ffp-contract=off)."""
    so = os.path.join(_HERE, "liboracle.so")
    if force or not os.path.exists(so):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"] + (["-B"] if force else []))
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
        _LIB.oracle_pt_render.restype = C.c_int
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None


def pt_render(lights, spheres, tris, camera, W, H, max_depth, spp, *, seed=1, rng_mode=0, trig_mode=0,
              sample_offset=0, window=None, threads=0, glass_shadow_opaque=0, max_delta=64, output_sum=False, ball_draw_reversed=False, russian_roulette=False,
              bvh=None):
    """Unidirectional PT + NEE (restates src/pt_cu.cu:20-250).  Returns (image[H,W,3] f32, stats dict).
    Pixels outside `window` = (x0, y0, x1, y1) stay zero.  bvh: the dict path_tracing_amd.export_bvh_host /
    Scene.export_bvh return -- triangles are then found by walking that tree on the host (same image as the scan) and
    stats carry boxes_* / tris_*: the independent count behind the bench's algorithmic bytes."""
    img = np.zeros((H, W, 3), np.float32)
    o = PtOpts()
    o.seed, o.rng_mode, o.trig_mode, o.sample_offset = int(seed), rng_mode, trig_mode, sample_offset
    o.x0, o.y0, o.x1, o.y1 = window if window else (0, 0, W, H)
    o.threads, o.glass_shadow_opaque, o.max_delta, o.output_sum = threads, glass_shadow_opaque, max_delta, int(output_sum)
    o.ball_draw_reversed = int(ball_draw_reversed)
    o.russian_roulette = int(russian_roulette)
    st = PtStats()
    cam = np.ascontiguousarray(camera)
    lights = np.ascontiguousarray(lights)
    spheres = np.ascontiguousarray(spheres)
    tris = np.ascontiguousarray(tris)
    b = None
    if bvh is not None:
        qn = np.ascontiguousarray(bvh["qnodes"], np.uint32)
        tw = np.ascontiguousarray(bvh["tris"], np.uint32)
        b = Bvh()
        b.qnodes, b.num_nodes, b.tris, b.num_tris = qn.ctypes.data, int(bvh["num_nodes"]), tw.ctypes.data, int(bvh["num_tris"])
        b.num_rounds = int(bvh["num_rounds"])
        b.qorigin = (C.c_float * 3)(*[float(v) for v in bvh["qorigin"]])
        b.qscale = (C.c_float * 3)(*[float(v) for v in bvh["qscale"]])
    fn = lib().oracle_pt_render_bvh
    fn.restype = C.c_int
    rc = fn(_p(lights), len(lights), _p(spheres), len(spheres), _p(tris), len(tris),
            cam.ctypes.data_as(C.c_void_p), img.ctypes.data_as(C.c_void_p),
            W, H, max_depth, spp, C.byref(o), C.byref(st), C.byref(b) if b is not None else None)
    if rc != 0:
        raise RuntimeError("oracle_pt_render failed rc=%d" % rc)
    return img, st.as_dict()


def sincos_2pi(u):
    u = np.ascontiguousarray(u, np.float32)
    s = np.empty_like(u)
    c = np.empty_like(u)
    lib().oracle_sincos_2pi(_p(u), len(u), _p(s), _p(c))
    return s, c


def pcg_uniforms(seed, pixel, sample, n):
    out = np.empty(n, np.float32)
    lib().oracle_pcg_uniforms(C.c_uint64(seed), C.c_uint32(pixel), C.c_uint32(sample), n, _p(out))
    return out


def bsdf_eval_pdf(mat6, wo, wi, n):
    m = np.zeros(7, np.float32)
    m[:6] = mat6
    out = np.zeros(4, np.float32)
    lib().oracle_bsdf_eval_pdf(_p(m), _p(np.ascontiguousarray(wo, np.float32)), _p(np.ascontiguousarray(wi, np.float32)),
                               _p(np.ascontiguousarray(n, np.float32)), _p(out))
    return out[:3].copy(), float(out[3])


def bsdf_sample(mat6, wo, n, u3, cur_eta=1.0, trig_mode=0):
    m = np.zeros(7, np.float32)
    m[:6] = mat6
    out = np.zeros(9, np.float32)
    lib().oracle_bsdf_sample(trig_mode, _p(m), _p(np.ascontiguousarray(wo, np.float32)),
                             _p(np.ascontiguousarray(n, np.float32)), _p(np.ascontiguousarray(u3, np.float32)),
                             C.c_float(cur_eta), _p(out))
    return dict(wi=out[0:3].copy(), f=out[3:6].copy(), pdf=float(out[6]), is_delta=bool(out[7]), new_eta=float(out[8]))


def function_kats(records):
    """[n, 24] float32 records -> [n, 40] results of the restated reference functions (layout: oracle_function_kats)."""
    rec = np.ascontiguousarray(records, np.float32).reshape(-1, 24)
    out = np.zeros((len(rec), 40), np.float32)
    lib().oracle_function_kats(_p(rec), len(rec), _p(out))
    return out


def closest_hits(lights, spheres, tris, ro, rd):
    ro = np.ascontiguousarray(ro, np.float32)
    rd = np.ascontiguousarray(rd, np.float32)
    n = len(ro)
    t = np.empty(n, np.float32)
    prim = np.empty(n, np.int32)
    lib().oracle_closest_hits(_p(lights), len(lights), _p(spheres), len(spheres), _p(tris), len(tris),
                              _p(ro), _p(rd), n, _p(t), _p(prim))
    return t, prim


def visibility(spheres, tris, p1, p2, glass_opaque=False):
    p1 = np.ascontiguousarray(p1, np.float32)
    p2 = np.ascontiguousarray(p2, np.float32)
    n = len(p1)
    vis = np.empty(n, np.int32)
    lib().oracle_visibility(_p(spheres), len(spheres), _p(tris), len(tris), _p(p1), _p(p2), n, int(glass_opaque), _p(vis))
    return vis


class BdptOpts(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("rng_mode", C.c_int), ("threads", C.c_int),
                ("x0", C.c_int), ("y0", C.c_int), ("x1", C.c_int), ("y1", C.c_int), ("max_delta", C.c_int)]


class BdptStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("samples", "closest_rays", "shadow_rays", "connections", "tri_tests", "sphere_tests")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def bdpt_render(lights, spheres, tris, order, eye, look_at, view_up, fov_deg, W, H, eye_depth=4, light_depth=4,
                spp=4, spl=8, *, seed=1, rng_mode=0, threads=0, window=None, rows=None, max_delta=0):
    """The cpu_bdpt estimator (restates src/cpu_bdpt.cpp:173-488).  rng_mode=1 replays the
    reference's std::mt19937 streams (bit-reproducible with threads=1; `rows` limits the render to
    the first rows).  Returns (image[H,W,3], stats)."""
    img = np.zeros((H, W, 3), np.float32)
    o = BdptOpts()
    o.seed, o.rng_mode, o.threads, o.max_delta = int(seed), rng_mode, threads, max_delta
    o.x0, o.y0, o.x1, o.y1 = window if window else (0, 0, W, H)
    if rows is not None:
        o.y1 = rows
    st = BdptStats()
    cam = np.concatenate([np.asarray(eye, np.float32), np.asarray(look_at, np.float32), np.asarray(view_up, np.float32),
                          np.asarray([fov_deg], np.float32)]).astype(np.float32)
    kind, index, group = (np.ascontiguousarray(a, np.int32) for a in order)
    lights = np.ascontiguousarray(lights); spheres = np.ascontiguousarray(spheres); tris = np.ascontiguousarray(tris)
    fn = lib().oracle_bdpt_render
    fn.restype = C.c_int
    rc = fn(_p(lights), len(lights), _p(spheres), len(spheres), _p(tris), len(tris), _p(kind), _p(index), _p(group), len(kind),
            _p(cam), _p(img), W, H, eye_depth, light_depth, spp, spl, C.byref(o), C.byref(st))
    if rc != 0:
        raise RuntimeError("oracle_bdpt_render failed rc=%d" % rc)
    return img, st.as_dict()
