"""path_tracing_amd -- MI355X-native unidirectional path-tracing hot path.

Python front-end of the C ABI in include/hpt.h (libhpt.so: hand-written HIP kernels for
gfx950 + host BVH build).  Mirrors the reference's launch-and-accumulate helper API
(reference include/pt_cu_helper.h:5-6, src/pt_cu_helper.cpp:12-77):

    Scene(lights, spheres, triangles)      ~ move_data_to_cuda_pt (upload once)
    Scene.render_pt(cam, W, H, depth, spp) ~ run_cuda_pt / pt_render_wrapper

There is no CPU fallback: if libhpt.so is missing or no HIP device is visible, calls raise.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import layouts, scene_io  # noqa: F401
from .layouts import CAMERA, LIGHT, SPHERE, TRIANGLE

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libhpt.so")

FLAG_BRUTE_FORCE = 1
FLAG_COUNT_WORK = 2
FLAG_OUTPUT_SUM = 4
FLAG_TIME_KERNELS = 8
FLAG_RUSSIAN_ROULETTE = 16
FLAG_SINGLE_PIPELINE = 32
FLAG_NO_HOST_WAIT = 64


class HptError(RuntimeError):
    pass


class Params(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("sample_offset", C.c_int32), ("max_delta", C.c_int32),
                ("rank", C.c_int32), ("world", C.c_int32), ("tile", C.c_int32),
                ("samples_per_pass", C.c_int32), ("flags", C.c_int32), ("reserved", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("closest_rays", C.c_uint64), ("shadow_rays", C.c_uint64),
                ("boxes_closest", C.c_uint64), ("tris_closest", C.c_uint64),
                ("boxes_shadow", C.c_uint64), ("tris_shadow", C.c_uint64), ("path_iters", C.c_uint64),
                ("ms_total", C.c_double), ("ms_extend", C.c_double), ("ms_shade", C.c_double),
                ("ms_connect", C.c_double), ("ms_other", C.c_double),
                ("n_extend", C.c_uint32), ("n_shade", C.c_uint32), ("n_connect", C.c_uint32), ("n_other", C.c_uint32),
                ("bvh_nodes", C.c_uint32), ("bvh_depth", C.c_uint32), ("n_tris", C.c_uint32), ("n_materials", C.c_uint32),
                ("ms_bvh_build", C.c_double), ("ms_upload", C.c_double),
                ("lane_steps_closest", C.c_uint64), ("wave_steps_closest", C.c_uint64),
                ("lane_steps_shadow", C.c_uint64), ("wave_steps_shadow", C.c_uint64),
                ("leaf_lane_closest", C.c_uint64), ("leaf_wave_closest", C.c_uint64),
                ("leaf_lane_shadow", C.c_uint64), ("leaf_wave_shadow", C.c_uint64),
                ("ms_resume", C.c_double), ("n_resume", C.c_uint32), ("split_budget", C.c_uint32),
                ("traced_rays_last_pass", C.c_uint64), ("long_rays_last_pass", C.c_uint64),
                ("bd_pairs", C.c_uint64), ("bd_survivors", C.c_uint64), ("bd_shadow_rays", C.c_uint64), ("bd_unoccluded", C.c_uint64),
                ("bd_nodes", C.c_uint64), ("bd_tris", C.c_uint64), ("bd_spheres", C.c_uint64), ("bd_group_boxes", C.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class BvhInfo(C.Structure):
    _fields_ = [("num_nodes", C.c_int32), ("num_tris", C.c_int32), ("bvh_depth", C.c_int32), ("num_rounds", C.c_int32),
                ("qorigin", C.c_float * 3), ("qscale", C.c_float * 3)]


# exported tree (include/hpt.h, hpt_bvh_info): 32-B quantised nodes as 8 uint32 words, 48-B triangles as 12 words
QNODE_WORDS = 8
TRI_WORDS = 12

_lib = None


def _share_the_hip_runtime_with_torch():
    """One HIP runtime per process.  PyTorch's ROCm wheels bundle their own libamdhip64.so (soname libamdhip64.so.7) and
    ask for it by the unversioned name, so if libhpt.so is loaded first it brings in /opt/rocm's copy, `import torch`
    then loads a second one, and whichever initialises second finds no device ("No HIP GPUs are available" / "no
    ROCm-capable device is detected").  When torch is installed but not imported yet, its copy is loaded here first;
    libhpt.so's NEEDED libamdhip64.so.7 then resolves to it by soname, and a later `import torch` reuses it."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec and spec.origin:
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)


def load_library() -> C.CDLL:
    """Loads csrc/libhpt.so; raises HptError if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        _share_the_hip_runtime_with_torch()
        path = os.environ.get("HPT_LIBRARY") or LIB_PATH        # HPT_LIBRARY: another build of libhpt.so (development A/B runs)
        if not os.path.exists(path):
            raise HptError("%s not built: run `make -C path_tracing_amd/csrc` "
                           "(or __graft_entry__.build()); there is no CPU fallback" % path)
        lib = C.CDLL(path)
        lib.hpt_last_error.restype = C.c_char_p
        lib.hpt_local_pixels.restype = C.c_int64
        lib.hpt_local_pixels.argtypes = [C.c_int, C.c_int, C.POINTER(Params)]
        for name in ("hpt_scene_create", "hpt_render_pt", "hpt_render_pt_device", "hpt_untile", "hpt_scene_set_groups",
                     "hpt_render_bdpt", "hpt_render_bdpt_device", "hpt_bdpt_render_wrapper",
                     "hpt_pt_render_wrapper", "hpt_get_stats", "hpt_trace_closest", "hpt_trace_visibility",
                     "hpt_device_count", "hpt_multi_create", "hpt_multi_num_devices", "hpt_multi_set_groups",
                     "hpt_multi_render_pt", "hpt_multi_render_bdpt", "hpt_multi_get_timing", "hpt_wrapper_set_devices",
                     "hpt_probe_functions", "hpt_tonemap", "hpt_tonemap_host", "hpt_bvh_export_host", "hpt_scene_export_bvh"):
            if hasattr(lib, name):          # (an older build loaded through HPT_LIBRARY for an A/B run lacks the newest entry points)
                getattr(lib, name).restype = C.c_int
        lib.hpt_scene_destroy.restype = None
        lib.hpt_wrapper_cache_clear.restype = None
        lib.hpt_wrapper_cache_clear.argtypes = []
        lib.hpt_scene_destroy.argtypes = [C.c_void_p]
        lib.hpt_multi_destroy.restype = None
        lib.hpt_multi_destroy.argtypes = [C.c_void_p]
        _lib = lib
    return _lib


def _check(rc: int):
    if rc != 0:
        raise HptError("hpt error %d: %s" % (rc, load_library().hpt_last_error().decode("utf-8", "replace")))


def _vp(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None


def device_count() -> int:
    return int(load_library().hpt_device_count())


def make_params(seed=1, sample_offset=0, max_delta=0, rank=0, world=1, tile=0, samples_per_pass=0, flags=0) -> Params:
    p = Params()
    p.seed, p.sample_offset, p.max_delta = int(seed), int(sample_offset), int(max_delta)
    p.rank, p.world, p.tile, p.samples_per_pass, p.flags, p.reserved = rank, world, tile, samples_per_pass, flags, 0
    return p


def local_pixels(W: int, H: int, params: Params) -> int:
    n = int(load_library().hpt_local_pixels(W, H, C.byref(params)))
    if n < 0:
        _check(1)
    return n


class Scene:
    """Device-resident scene + BVH (the upload half of the reference's helper API:
    move_data_to_cuda_pt, src/pt_cu_helper.cpp:12-64).  Inputs are arrays of the reference's
    records (layouts.LIGHT / SPHERE / TRIANGLE), e.g. from scene_io.flatten_for_pt."""

    def __init__(self, lights, spheres, triangles):
        self._lib = load_library()
        self._h = C.c_void_p()
        lights = np.ascontiguousarray(lights, LIGHT)
        spheres = np.ascontiguousarray(spheres, SPHERE)
        triangles = np.ascontiguousarray(triangles, TRIANGLE)
        _check(self._lib.hpt_scene_create(_vp(lights), len(lights), _vp(spheres), len(spheres),
                                          _vp(triangles), len(triangles), C.byref(self._h)))
        self.num_lights, self.num_spheres, self.num_triangles = len(lights), len(spheres), len(triangles)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.hpt_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- rendering ----------------------------------------------------------------------
    def render_pt(self, camera, W, H, eye_depth=4, spp=8, params: Params | None = None) -> np.ndarray:
        """Blocking whole-image render (run_cuda_pt, src/pt_cu_helper.cpp:66-77).
        Returns float32 [H, W, 3], row 0 = top, linear mean radiance."""
        params = params or make_params()
        cam = np.ascontiguousarray(camera, CAMERA)
        img = np.empty((H, W, 3), np.float32)
        _check(self._lib.hpt_render_pt(self._h, _vp(cam.reshape(1)), W, H, eye_depth, spp, C.byref(params), _vp(img)))
        return img

    def render_pt_device(self, camera, W, H, eye_depth, spp, params: Params, d_local_ptr: int, stream: int = 0):
        """Asynchronous render of this rank's tiles into device memory (packed local order)."""
        cam = np.ascontiguousarray(camera, CAMERA)
        _check(self._lib.hpt_render_pt_device(self._h, _vp(cam.reshape(1)), W, H, eye_depth, spp, C.byref(params),
                                              C.c_void_p(d_local_ptr), C.c_void_p(stream)))

    # -- bidirectional estimator (run_cuda_bdpt, reference include/bdpt_cu_helper.h:6; semantics of run_cpu_bdpt) ---
    def set_groups(self, kind, index, group):
        """Scene-file grouping of the objects (kind 0 sphere / 1 triangle, index, group id; insertion order)."""
        k = np.ascontiguousarray(kind, np.int32); i = np.ascontiguousarray(index, np.int32); g = np.ascontiguousarray(group, np.int32)
        _check(self._lib.hpt_scene_set_groups(self._h, _vp(k), _vp(i), _vp(g), len(k)))

    def render_bdpt(self, camera, W, H, eye_depth=4, light_depth=4, spp=8, spl=8, params: Params | None = None) -> np.ndarray:
        params = params or make_params()
        cam = np.ascontiguousarray(camera, CAMERA)
        img = np.empty((H, W, 3), np.float32)
        _check(self._lib.hpt_render_bdpt(self._h, _vp(cam.reshape(1)), W, H, eye_depth, light_depth, spp, spl, C.byref(params), _vp(img)))
        return img

    def render_bdpt_device(self, camera, W, H, eye_depth, light_depth, spp, spl, params: Params, d_local_ptr: int, stream: int = 0):
        cam = np.ascontiguousarray(camera, CAMERA)
        _check(self._lib.hpt_render_bdpt_device(self._h, _vp(cam.reshape(1)), W, H, eye_depth, light_depth, spp, spl, C.byref(params),
                                                C.c_void_p(d_local_ptr), C.c_void_p(stream)))

    def stats(self) -> dict:
        st = Stats()
        _check(self._lib.hpt_get_stats(self._h, C.byref(st)))
        return st.as_dict()

    def export_bvh(self) -> dict:
        """The tree this scene's device holds (include/hpt.h, hpt_scene_export_bvh): dict(info, qnodes [n, 8] uint32,
        tris [m, 12] uint32 words of the 48-B leaf-order records)."""
        return _export_bvh(lambda info, qn, qcap, tr, tcap: self._lib.hpt_scene_export_bvh(self._h, info, qn, qcap, tr, tcap))

    # -- ray probes (tests) ---------------------------------------------------------------
    def trace_closest(self, origins, dirs, brute_force=False):
        o = np.ascontiguousarray(origins, np.float32)
        d = np.ascontiguousarray(dirs, np.float32)
        n = len(o)
        t = np.empty(n, np.float32)
        prim = np.empty(n, np.int32)
        _check(self._lib.hpt_trace_closest(self._h, _vp(o), _vp(d), n, FLAG_BRUTE_FORCE if brute_force else 0, _vp(t), _vp(prim)))
        return t, prim

    def trace_visibility(self, p1, p2, brute_force=False):
        a = np.ascontiguousarray(p1, np.float32)
        b = np.ascontiguousarray(p2, np.float32)
        n = len(a)
        vis = np.empty(n, np.int32)
        _check(self._lib.hpt_trace_visibility(self._h, _vp(a), _vp(b), n, FLAG_BRUTE_FORCE if brute_force else 0, _vp(vis)))
        return vis


class MultiScene:
    """The blocking render call fanned out over the devices of one node inside ONE process (include/hpt.h,
    hpt_multi_*): scene and BVH built once and uploaded to every device, image tiles per device, RCCL gather
    on the first device (exchange=0) or peer copies (exchange=1; the only mode that accepts several ranks on
    one device).  Same results as Scene.render_pt / render_bdpt, bit for bit."""

    def __init__(self, lights, spheres, triangles, device_ids=None, num_devices=0, exchange=0):
        self._lib = load_library()
        self._h = C.c_void_p()
        lights = np.ascontiguousarray(lights, LIGHT)
        spheres = np.ascontiguousarray(spheres, SPHERE)
        triangles = np.ascontiguousarray(triangles, TRIANGLE)
        ids = None
        if device_ids is not None:
            ids = np.ascontiguousarray(device_ids, np.int32)
            num_devices = len(ids)
        _check(self._lib.hpt_multi_create(_vp(lights), len(lights), _vp(spheres), len(spheres), _vp(triangles), len(triangles),
                                          _vp(ids), int(num_devices), int(exchange), C.byref(self._h)))
        self.num_devices = int(self._lib.hpt_multi_num_devices(self._h))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.hpt_multi_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_groups(self, kind, index, group):
        k = np.ascontiguousarray(kind, np.int32); i = np.ascontiguousarray(index, np.int32); g = np.ascontiguousarray(group, np.int32)
        _check(self._lib.hpt_multi_set_groups(self._h, _vp(k), _vp(i), _vp(g), len(k)))

    def render_pt(self, camera, W, H, eye_depth=4, spp=8, params: Params | None = None) -> np.ndarray:
        params = params or make_params()
        cam = np.ascontiguousarray(camera, CAMERA)
        img = np.empty((H, W, 3), np.float32)
        _check(self._lib.hpt_multi_render_pt(self._h, _vp(cam.reshape(1)), W, H, eye_depth, spp, C.byref(params), _vp(img)))
        return img

    def render_bdpt(self, camera, W, H, eye_depth=4, light_depth=4, spp=8, spl=8, params: Params | None = None) -> np.ndarray:
        params = params or make_params()
        cam = np.ascontiguousarray(camera, CAMERA)
        img = np.empty((H, W, 3), np.float32)
        _check(self._lib.hpt_multi_render_bdpt(self._h, _vp(cam.reshape(1)), W, H, eye_depth, light_depth, spp, spl, C.byref(params), _vp(img)))
        return img

    def timing(self) -> dict:
        per = (C.c_double * self.num_devices)()
        g = C.c_double(); t = C.c_double()
        _check(self._lib.hpt_multi_get_timing(self._h, per, C.byref(g), C.byref(t)))
        return {"render_ms_per_device": list(per), "gather_ms": g.value, "total_ms": t.value}


def _export_bvh(call) -> dict:
    info = BvhInfo()
    _check(call(C.byref(info), None, C.c_size_t(0), None, C.c_size_t(0)))
    qn = np.zeros((max(info.num_nodes, 1), QNODE_WORDS), np.uint32)
    tr = np.zeros((max(info.num_tris, 1), TRI_WORDS), np.uint32)
    _check(call(C.byref(info), _vp(qn), C.c_size_t(qn.nbytes), _vp(tr), C.c_size_t(tr.nbytes)))
    return {"num_nodes": info.num_nodes, "num_tris": info.num_tris, "bvh_depth": info.bvh_depth, "num_rounds": info.num_rounds,
            "qorigin": np.array(info.qorigin, np.float32), "qscale": np.array(info.qscale, np.float32),
            "qnodes": qn[:info.num_nodes], "tris": tr[:info.num_tris]}


def export_bvh_host(lights, spheres, triangles) -> dict:
    """The tree hpt_scene_create would build and upload for these records, built on the host (no device needed)."""
    lib = load_library()
    lights = np.ascontiguousarray(lights, LIGHT)
    spheres = np.ascontiguousarray(spheres, SPHERE)
    triangles = np.ascontiguousarray(triangles, TRIANGLE)
    return _export_bvh(lambda info, qn, qcap, tr, tcap: lib.hpt_bvh_export_host(
        _vp(lights), len(lights), _vp(spheres), len(spheres), _vp(triangles), len(triangles), info, qn, qcap, tr, tcap))


def probe_functions(records) -> np.ndarray:
    """Device BSDF / Fresnel / GGX functions on [n, 24] float32 records -> [n, 40] results (tests; include/hpt.h)."""
    rec = np.ascontiguousarray(records, np.float32).reshape(-1, 24)
    out = np.zeros((len(rec), 40), np.float32)
    _check(load_library().hpt_probe_functions(_vp(rec), len(rec), _vp(out)))
    return out


def tonemap(image) -> np.ndarray:
    """8-bit output stage on the device (reference src/main_cli.cpp:225-242): [H, W, 3] float32 -> uint8 RGB."""
    img = np.ascontiguousarray(image, np.float32)
    out = np.zeros(img.shape, np.uint8)
    _check(load_library().hpt_tonemap_host(_vp(img), _vp(out), C.c_int64(img.size // 3), 0))
    return out


def wrapper_set_devices(n: int) -> None:
    """Number of devices the one-shot wrappers (pt_render_wrapper / bdpt_render_wrapper) fan out over."""
    _check(load_library().hpt_wrapper_set_devices(int(n)))


def untile(d_gathered_ptr: int, d_image_ptr: int, W: int, H: int, params: Params, stream: int = 0):
    """[rank][local slot] packed framebuffers -> row-major W*H image (device pointers)."""
    _check(load_library().hpt_untile(C.c_void_p(d_gathered_ptr), C.c_void_p(d_image_ptr), W, H, C.byref(params), C.c_void_p(stream)))


def wrapper_cache_clear() -> None:
    """Releases the scene the one-shot wrappers keep between calls (include/hpt.h)."""
    load_library().hpt_wrapper_cache_clear()


def pt_render_wrapper(lights, spheres, triangles, camera, W, H, eye_depth, spp, seed=-1,
                      scene_min=(0, 0, 0), scene_max=(0, 0, 0), light_depth=4, light_sample=8) -> np.ndarray:
    """One-shot render with the reference's pt_render_wrapper argument list
    (reference include/pt_cu.cuh:6-13)."""
    lib = load_library()
    lights = np.ascontiguousarray(lights, LIGHT)
    spheres = np.ascontiguousarray(spheres, SPHERE)
    triangles = np.ascontiguousarray(triangles, TRIANGLE)
    cam = np.ascontiguousarray(camera, CAMERA).reshape(1)
    img = np.empty((H, W, 3), np.float32)
    mn = (C.c_float * 3)(*scene_min)
    mx = (C.c_float * 3)(*scene_max)
    _check(lib.hpt_pt_render_wrapper(_vp(lights), len(lights), _vp(spheres), len(spheres), _vp(triangles), len(triangles),
                                     mn, mx, _vp(cam), _vp(img), W, H, light_depth, light_sample, eye_depth, spp,
                                     C.c_int64(seed)))
    return img
