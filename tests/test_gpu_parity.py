"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI,
against the CPU oracle on the same seeded inputs, against the committed golden vectors, and --
at the benchmark's full sizes -- through size-independent properties.

Tolerance: BASELINE.json's bar is per-pixel RMSE < 1e-3 on linear radiance.  Oracle and kernels
evaluate the same IEEE expressions in the same order with the same random streams, so the
observed difference is exactly zero; the tests assert the stated RMSE bar and, more tightly,
max-abs <= 1e-6.
"""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden_cases, rmse, scene_by_name

pytestmark = pytest.mark.gpu

RMSE_BAR = 1e-3     # north_star tolerance
TIGHT = 1e-6


def assert_parity(img, ref):
    assert img.shape == ref.shape
    assert rmse(img, ref) < RMSE_BAR
    assert float(np.abs(img - ref).max()) <= TIGHT


@pytest.mark.parametrize("fname", golden_cases())
def test_gpu_matches_committed_goldens(hpt, sio, fname):
    g = np.load(os.path.join(GOLDEN, fname))
    (L, sp, tr), (eye, look, up) = scene_by_name(sio, str(g["scene"]))
    W, H, depth, spp, seed = (int(g[k]) for k in ("W", "H", "depth", "spp", "seed"))
    cam = sio.make_camera(eye, look, up, 50.0, W, H)
    with hpt.Scene(L, sp, tr) as scene:
        img = scene.render_pt(cam, W, H, depth, spp, hpt.make_params(seed=seed, flags=hpt.FLAG_COUNT_WORK))
        st = scene.stats()
    assert_parity(img, g["image"])
    # same rays traced as the oracle traced
    assert st["closest_rays"] == int(g["closest_rays"]) and st["shadow_rays"] == int(g["shadow_rays"])


@pytest.mark.parametrize("name,W,H,depth,spp,seed", [
    ("input", 96, 80, 4, 6, 21),          # glass, mirror, conductors, 4 cone lights (config 1 scene)
    ("input", 33, 17, 7, 5, 22),          # odd size, deeper paths
    ("cornell_diffuse", 128, 128, 4, 4, 23),   # config 2 scene shape
    ("mis_test", 64, 64, 4, 4, 24),       # config 4 scene under the PT estimator
])
def test_gpu_matches_oracle_live(hpt, sio, oracle_mod, name, W, H, depth, spp, seed):
    (L, sp, tr), (eye, look, up) = scene_by_name(sio, name)
    cam = sio.make_camera(eye, look, up, 50.0, W, H)
    ref, st = oracle_mod.pt_render(L, sp, tr, cam, W, H, depth, spp, seed=seed)
    with hpt.Scene(L, sp, tr) as scene:
        img = scene.render_pt(cam, W, H, depth, spp, hpt.make_params(seed=seed))
        img_brute = scene.render_pt(cam, W, H, depth, spp, hpt.make_params(seed=seed, flags=hpt.FLAG_BRUTE_FORCE))
    assert_parity(img, ref)
    assert np.array_equal(img, img_brute)        # BVH == scan


def test_parallel_light_and_transparent_blockers(hpt, sio, oracle_mod):
    # a directional light (pt_cu.cu:130-149) above a glass pane: shadow rays pass eta > 0 blockers
    L, sp, tr = sio.cornell_diffuse()
    L["is_parallel"] = 1
    L["dir"] = np.array([0.2, -1.0, 0.1], np.float32) / np.linalg.norm([0.2, -1.0, 0.1]).astype(np.float32)
    L["illum"] = (0.5, 0.4, 0.3)
    pane = tr[:2].copy()
    pane["v0"], pane["v1"], pane["v2"] = [(-0.3, 0.0, 0.0), (-0.3, 0.0, 0.0)], [(0.3, 0.0, 0.0), (0.3, 0.0, 0.6)], [(0.3, 0.0, 0.6), (-0.3, 0.0, 0.6)]
    pane["mtl"]["base_color"] = 1.0
    pane["mtl"]["roughness"], pane["mtl"]["metallic"], pane["mtl"]["eta"] = 0.0, 0.0, 1.5
    tr = np.concatenate([tr, pane])
    cam = sio.make_camera(sio.CORNELL_EYE, sio.CORNELL_LOOK, sio.CORNELL_UP, 50.0, 64, 64)
    ref, st = oracle_mod.pt_render(L, sp, tr, cam, 64, 64, 4, 4, seed=31)
    assert st["shadow_rays"] > 1000
    with hpt.Scene(L, sp, tr) as scene:
        img = scene.render_pt(cam, 64, 64, 4, 4, hpt.make_params(seed=31))
    assert_parity(img, ref)
    assert img.mean() > 0.01


def test_ray_probes_match_scan_at_100k_triangles(hpt, sio, oracle_mod):
    L, sp, tr = sio.cornell_with_sphere(100_000)
    rng = np.random.default_rng(5)
    n = 100_000
    o = rng.uniform(-0.45, 0.45, (n, 3)).astype(np.float32)
    o[: n // 2] = np.array([0, 0, -1], np.float32)
    tgt = np.array([-0.15, 0.2, 0.45]) + rng.normal(size=(n, 3)) * 0.12
    d = tgt - o
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    p2 = (o + d * rng.uniform(0.05, 1.5, (n, 1))).astype(np.float32)
    with hpt.Scene(L, sp, tr) as scene:
        t_bvh, p_bvh = scene.trace_closest(o, d)
        t_scan, p_scan = scene.trace_closest(o, d, brute_force=True)
        v_bvh = scene.trace_visibility(o, p2)
        v_scan = scene.trace_visibility(o, p2, brute_force=True)
        st = scene.stats()
    assert np.array_equal(t_bvh, t_scan) and np.array_equal(p_bvh, p_scan)
    assert np.array_equal(v_bvh, v_scan)
    assert (p_bvh >= 13).mean() > 0.3 and 0.2 < v_bvh.mean() < 0.9          # the sphere is actually exercised
    assert st["bvh_depth"] <= 30
    k = 1500                                                                # oracle: 100k triangle tests per ray
    t_o, p_o = oracle_mod.closest_hits(L, sp, tr, o[:k], d[:k])
    v_o = oracle_mod.visibility(sp, tr, o[:k], p2[:k])
    assert np.array_equal(t_o, t_bvh[:k]) and np.array_equal(p_o, p_bvh[:k]) and np.array_equal(v_o, v_bvh[:k])


def test_exact_ties_resolve_like_the_scan(hpt, oracle_mod, sio):
    # two coincident triangles and a coplanar duplicate quad: the scan keeps the first (strict '<')
    L, sp, tr = sio.cornell_diffuse()
    dup = tr[:12].copy()
    dup["mtl"]["base_color"] = (0.1, 0.9, 0.1)
    tr2 = np.concatenate([tr, dup, dup])
    rng = np.random.default_rng(2)
    n = 20000
    o = np.tile(np.array([0, 0, -1], np.float32), (n, 1))
    d = rng.normal(size=(n, 3)); d[:, 2] = np.abs(d[:, 2]) + 0.2
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    t_o, p_o = oracle_mod.closest_hits(L, sp, tr2, o, d)
    with hpt.Scene(L, sp, tr2) as scene:
        t_g, p_g = scene.trace_closest(o, d)
    assert np.array_equal(t_o, t_g) and np.array_equal(p_o, p_g)
    assert p_g.max() < 1 + 36        # always the first copy


@pytest.mark.parametrize("case", ["empty", "lights_only", "one_triangle", "no_lights", "spheres_only"])
def test_edge_scenes(hpt, sio, oracle_mod, case):
    from path_tracing_amd.layouts import LIGHT, SPHERE, TRIANGLE
    L0, sp0, tr0 = sio.flatten_for_pt(sio.load_scene(os.path.join(GOLDEN, "scenes", "input.txt")))
    L, sp, tr = np.zeros(0, LIGHT), np.zeros(0, SPHERE), np.zeros(0, TRIANGLE)
    if case == "lights_only":
        L = L0
    elif case == "one_triangle":
        L, tr = L0, tr0[6:7]
    elif case == "no_lights":
        sp, tr = sp0, tr0
    elif case == "spheres_only":
        L, sp = L0, sp0
    cam = sio.make_camera((0, 0, -1), (0, 0, 1), (0, 1, 0), 50.0, 40, 24)
    ref, _ = oracle_mod.pt_render(L, sp, tr, cam, 40, 24, 4, 3, seed=4)
    with hpt.Scene(L, sp, tr) as scene:
        img = scene.render_pt(cam, 40, 24, 4, 3, hpt.make_params(seed=4))
    assert_parity(img, ref)
    if case in ("empty", "no_lights"):
        assert not img.any()


def test_pass_size_sample_offset_and_determinism(hpt, sio, input_scene):
    sc, (L, sp, tr) = input_scene
    cam = sio.camera_for(sc, 72, 56)
    with hpt.Scene(L, sp, tr) as scene:
        a = scene.render_pt(cam, 72, 56, 4, 7, hpt.make_params(seed=9))
        b = scene.render_pt(cam, 72, 56, 4, 7, hpt.make_params(seed=9, samples_per_pass=2))
        c = scene.render_pt(cam, 72, 56, 4, 7, hpt.make_params(seed=9, samples_per_pass=7, tile=8))
        other = scene.render_pt(cam, 72, 56, 4, 7, hpt.make_params(seed=10))
        s0 = scene.render_pt(cam, 72, 56, 4, 3, hpt.make_params(seed=9, flags=hpt.FLAG_OUTPUT_SUM))
        s1 = scene.render_pt(cam, 72, 56, 4, 4, hpt.make_params(seed=9, sample_offset=3, flags=hpt.FLAG_OUTPUT_SUM))
    assert np.array_equal(a, b) and np.array_equal(a, c)
    assert not np.array_equal(a, other)
    assert np.allclose((s0 + s1) / 7.0, a, rtol=1e-5, atol=1e-6)     # progressive accumulation on the host


def test_one_shot_wrapper_and_clock_seed(hpt, sio, input_scene, oracle_mod):
    sc, (L, sp, tr) = input_scene
    cam = sio.camera_for(sc, 32, 32)
    img = hpt.pt_render_wrapper(L, sp, tr, cam, 32, 32, 4, 4, seed=77)
    ref, _ = oracle_mod.pt_render(L, sp, tr, cam, 32, 32, 4, 4, seed=77)
    assert_parity(img, ref)
    unseeded = hpt.pt_render_wrapper(L, sp, tr, cam, 32, 32, 4, 4)      # seed < 0: time(NULL) like the reference
    assert np.isfinite(unseeded).all() and unseeded.mean() > 0.05


def test_virtual_ranks_assemble_bitwise(hpt, sio, input_scene):
    """G in {1,2,3,8} virtual ranks rendered back to back on one device and un-tiled by the
    device kernel give the same image bit for bit (the multi-GPU data path minus the wire)."""
    import torch
    sc, (L, sp, tr) = input_scene
    W, H, spp = 100, 76, 3
    cam = sio.camera_for(sc, W, H)
    stream = torch.cuda.current_stream().cuda_stream
    with hpt.Scene(L, sp, tr) as scene:
        ref = scene.render_pt(cam, W, H, 4, spp, hpt.make_params(seed=15))
        for world, tile in ((1, 32), (2, 32), (3, 16), (8, 8)):
            n_local = hpt.local_pixels(W, H, hpt.make_params(world=world, tile=tile))
            gathered = torch.zeros((world, n_local, 3), dtype=torch.float32, device="cuda")
            for r in range(world):
                p = hpt.make_params(seed=15, rank=r, world=world, tile=tile)
                scene.render_pt_device(cam, W, H, 4, spp, p, gathered[r].data_ptr(), stream)
            image = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
            hpt.untile(gathered.data_ptr(), image.data_ptr(), W, H, hpt.make_params(world=world, tile=tile), stream)
            torch.cuda.synchronize()
            assert np.array_equal(image.cpu().numpy(), ref), "world=%d" % world


def test_full_size_properties_config3(hpt, sio, oracle_mod):
    """Config 3 shape (about 100k triangles, 1024 x 1024) at reduced spp: properties that do not
    need the oracle at full size, plus an oracle check of a window of the same image."""
    import torch
    L, sp, tr = sio.cornell_with_sphere(100_000)
    W = H = 1024
    spp = 4
    cam = sio.make_camera(sio.CORNELL_EYE, sio.CORNELL_LOOK, sio.CORNELL_UP, 50.0, W, H)
    with hpt.Scene(L, sp, tr) as scene:
        a = scene.render_pt(cam, W, H, 4, spp, hpt.make_params(seed=1, flags=hpt.FLAG_COUNT_WORK))
        st = scene.stats()
        b = scene.render_pt(cam, W, H, 4, spp, hpt.make_params(seed=1, samples_per_pass=1))
        assert np.array_equal(a, b)                                   # pass size does not matter
        stream = torch.cuda.current_stream().cuda_stream
        world = 8
        n_local = hpt.local_pixels(W, H, hpt.make_params(world=world))
        gathered = torch.zeros((world, n_local, 3), dtype=torch.float32, device="cuda")
        for r in range(world):
            scene.render_pt_device(cam, W, H, 4, spp, hpt.make_params(seed=1, rank=r, world=world), gathered[r].data_ptr(), stream)
        image = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
        hpt.untile(gathered.data_ptr(), image.data_ptr(), W, H, hpt.make_params(world=world), stream)
        torch.cuda.synchronize()
        assert np.array_equal(image.cpu().numpy(), a)                 # 8 virtual ranks == 1 rank
    assert np.isfinite(a).all() and a.min() >= 0.0
    assert st["samples"] == W * H * spp
    assert 3.0 < st["closest_rays"] / st["samples"] < 5.5
    assert (st["boxes_closest"] + st["boxes_shadow"]) / (st["closest_rays"] + st["shadow_rays"]) < 3 * 2 * np.log2(len(tr))
    # the oracle on a 24 x 16 window that straddles the sphere's silhouette (100k triangle tests per ray)
    x0, y0, x1, y1 = 560, 380, 584, 396
    ref, _ = oracle_mod.pt_render(L, sp, tr, cam, W, H, 4, spp, seed=1, window=(x0, y0, x1, y1))
    assert_parity(a[y0:y1, x0:x1], ref[y0:y1, x0:x1])


def test_kernel_timing_stats_are_reported(hpt, sio, input_scene):
    sc, (L, sp, tr) = input_scene
    cam = sio.camera_for(sc, 256, 256)
    with hpt.Scene(L, sp, tr) as scene:
        scene.render_pt(cam, 256, 256, 4, 8, hpt.make_params(seed=1, flags=hpt.FLAG_TIME_KERNELS))
        st = scene.stats()
    assert st["n_extend"] >= 4 and st["n_extend"] == st["n_shade"] and st["n_connect"] >= 1
    assert st["ms_extend"] > 0 and st["ms_shade"] > 0 and st["ms_total"] >= st["ms_extend"]


def test_errors_surface_as_exceptions(hpt, sio, input_scene):
    sc, (L, sp, tr) = input_scene
    cam = sio.camera_for(sc, 16, 16)
    with hpt.Scene(L, sp, tr) as scene:
        with pytest.raises(hpt.HptError):
            scene.render_pt(cam, 16, 16, 4, 0)
        with pytest.raises(hpt.HptError):
            scene.render_pt(cam, 16, 16, 0, 1)
        with pytest.raises(hpt.HptError):
            scene.render_pt(cam, 16, 16, 4, 1, hpt.make_params(world=2, rank=0))


def test_reference_named_cpp_entry_point(hpt, sio, input_scene, oracle_mod, monkeypatch):
    """pt_render_wrapper with the reference's C++ signature (by-value float3 / CudaCamera),
    called through its mangled name exactly as the reference's pt_cu_helper.cpp would."""
    from test_boundary import REF_SYMBOL
    from conftest import ROOT

    class F3(C.Structure):
        _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]

    class Cam(C.Structure):
        _fields_ = [(n, F3) for n in ("eye", "U", "V", "W", "UL", "dx", "dy")]

    sc, (L, sp, tr) = input_scene
    W = H = 40
    cam = sio.camera_for(sc, W, H)
    lib = C.CDLL(os.path.join(ROOT, "path_tracing_amd", "csrc", "libhpt_ref.so"))
    fn = getattr(lib, REF_SYMBOL)
    fn.restype = None
    fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, F3, F3, Cam, C.c_void_p] + [C.c_int] * 6
    ccam = Cam.from_buffer_copy(np.ascontiguousarray(cam).tobytes())
    img = np.zeros((H, W, 3), np.float32)
    monkeypatch.setenv("HPT_SEED", "41")
    fn(L.ctypes.data, len(L), sp.ctypes.data, len(sp), tr.ctypes.data, len(tr), F3(0, 0, 0), F3(0, 0, 0), ccam,
       img.ctypes.data, W, H, 4, 8, 4, 3)
    ref, _ = oracle_mod.pt_render(L, sp, tr, cam, W, H, 4, 3, seed=41)
    assert_parity(img, ref)


# ---- bidirectional estimator (config 1 / config 4; oracle: oracle/bdpt_oracle.cpp, which replays the reference's
# ---- own cpu_bdpt.cpp image bit for bit -- tests/test_bdpt_oracle.py) ---------------------------------------------

def _bdpt_case(sio, oracle_mod, name):
    sc = sio.load_scene(os.path.join(GOLDEN, "scenes", name + ".txt"))
    L, sp, tr = sio.flatten_for_pt(sc)
    return sc, L, sp, tr, sio.object_order(sc)


@pytest.mark.parametrize("name,W,H,spp,spl,depth", [
    ("input", 96, 64, 3, 8, 4),        # config 1 scene: glass, mirror, conductors, 4 cone lights, 2 groups
    ("input", 33, 21, 2, 3, 6),
    ("mis_test", 64, 64, 4, 8, 4),     # config 4 scene (degenerate: cos(360 deg) = 1 rejects the cone test)
])
def test_bdpt_gpu_matches_cpu_bdpt_oracle(hpt, sio, oracle_mod, name, W, H, spp, spl, depth):
    sc, L, sp, tr, order = _bdpt_case(sio, oracle_mod, name)
    ref, st = oracle_mod.bdpt_render(L, sp, tr, order, sc.eye, sc.look_at, sc.view_up, sc.fov, W, H, depth, depth, spp, spl, seed=8)
    cam = sio.make_camera(sc.eye, sc.look_at, sc.view_up, sc.fov, W, H, tan_in_float=True)
    with hpt.Scene(L, sp, tr) as scene:
        scene.set_groups(*order)
        img = scene.render_bdpt(cam, W, H, depth, depth, spp, spl, hpt.make_params(seed=8))
        img2 = scene.render_bdpt(cam, W, H, depth, depth, spp, spl, hpt.make_params(seed=8, samples_per_pass=1, tile=8))
    assert_parity(img, ref)
    assert np.array_equal(img, img2)


def test_bdpt_config1_full_size(hpt, sio, oracle_mod):
    """BASELINE config 1: input.txt, 256 x 256, 4 spp (spl 8, depth 4/4) -- GPU vs the cpu_bdpt oracle."""
    sc, L, sp, tr, order = _bdpt_case(sio, oracle_mod, "input")
    ref, st = oracle_mod.bdpt_render(L, sp, tr, order, sc.eye, sc.look_at, sc.view_up, sc.fov, 256, 256, 4, 4, 4, 8, seed=1)
    cam = sio.make_camera(sc.eye, sc.look_at, sc.view_up, sc.fov, 256, 256, tan_in_float=True)
    with hpt.Scene(L, sp, tr) as scene:
        scene.set_groups(*order)
        img = scene.render_bdpt(cam, 256, 256, 4, 4, 4, 8, hpt.make_params(seed=1))
    assert_parity(img, ref)
    assert st["shadow_rays"] > 100 * st["samples"]


def test_bdpt_synthetic_scene_and_ties(hpt, sio, oracle_mod):
    # single implicit group (spheres then triangles), coincident triangles: the CPU loop keeps the LAST of equal hits
    L, sp, tr = sio.cornell_with_sphere(1500)
    dup = tr[:12].copy(); dup["mtl"]["base_color"] = (0.1, 0.8, 0.1)
    tr = np.concatenate([tr, dup])
    order = sio.object_order(None, sp, tr)
    ref, _ = oracle_mod.bdpt_render(L, sp, tr, order, sio.CORNELL_EYE, sio.CORNELL_LOOK, sio.CORNELL_UP, 50.0, 48, 48, 4, 4, 2, 4, seed=3)
    cam = sio.make_camera(sio.CORNELL_EYE, sio.CORNELL_LOOK, sio.CORNELL_UP, 50.0, 48, 48, tan_in_float=True)
    with hpt.Scene(L, sp, tr) as scene:
        img = scene.render_bdpt(cam, 48, 48, 4, 4, 2, 4, hpt.make_params(seed=3))
    assert_parity(img, ref)


def test_bdpt_no_lights_and_one_shot_wrapper(hpt, sio, oracle_mod):
    from path_tracing_amd.layouts import LIGHT
    sc, L, sp, tr, order = _bdpt_case(sio, oracle_mod, "input")
    cam = sio.make_camera(sc.eye, sc.look_at, sc.view_up, sc.fov, 24, 24, tan_in_float=True)
    with hpt.Scene(np.zeros(0, LIGHT), sp, tr) as scene:
        assert not scene.render_bdpt(cam, 24, 24, 4, 4, 2, 2).any()            # src/cpu_bdpt.cpp:178
    # reference wrapper argument list: illum arrives divided by light_sample (src/bdpt_cu_helper.cpp:60-62)
    lib = hpt.load_library()
    L8 = L.copy(); L8["illum"] = L["illum"] / np.float32(8)
    img = np.empty((24, 24, 3), np.float32)
    z3 = (C.c_float * 3)()
    rc = lib.hpt_bdpt_render_wrapper(L8.ctypes.data_as(C.c_void_p), len(L8), sp.ctypes.data_as(C.c_void_p), len(sp),
                                     tr.ctypes.data_as(C.c_void_p), len(tr), z3, z3, np.ascontiguousarray(cam).ctypes.data_as(C.c_void_p),
                                     img.ctypes.data_as(C.c_void_p), 24, 24, 4, 8, 4, 2, 8, C.c_int64(6))
    assert rc == 0
    ref, _ = oracle_mod.bdpt_render(L, sp, tr, sio.object_order(None, sp, tr), sc.eye, sc.look_at, sc.view_up, sc.fov, 24, 24, 4, 4, 2, 8, seed=6)
    assert_parity(img, ref)


def test_russian_roulette_is_opt_in_and_matches_the_oracle(hpt, sio, oracle_mod, input_scene):
    """Not in the reference: default off; when asked for, the GPU and the oracle play the same
    roulette (survival q = clamp(max throughput, 0.05, 1), one extra uniform per non-delta bounce)."""
    sc, (L, sp, tr) = input_scene
    cam = sio.camera_for(sc, 64, 48)
    plain, _ = oracle_mod.pt_render(L, sp, tr, cam, 64, 48, 8, 16, seed=12)
    ref, st = oracle_mod.pt_render(L, sp, tr, cam, 64, 48, 8, 16, seed=12, russian_roulette=True)
    with hpt.Scene(L, sp, tr) as scene:
        img = scene.render_pt(cam, 64, 48, 8, 16, hpt.make_params(seed=12, flags=hpt.FLAG_RUSSIAN_ROULETTE | hpt.FLAG_COUNT_WORK))
        rays = scene.stats()["closest_rays"]
        off = scene.render_pt(cam, 64, 48, 8, 16, hpt.make_params(seed=12))
    assert_parity(img, ref)
    assert_parity(off, plain)
    assert rays == st["closest_rays"]
    _, st_plain = oracle_mod.pt_render(L, sp, tr, cam, 64, 48, 8, 16, seed=12)
    assert st["closest_rays"] < 0.9 * st_plain["closest_rays"]            # roulette does cut work at depth 8
    assert abs(img.mean() - plain.mean()) < 0.1 * plain.mean()           # and keeps the expectation (up to the 15-clamp)


def test_split_trace_step_does_not_change_the_image(hpt, sio, oracle_mod):
    """A trace step is two launches: every ray gets `budget` node steps, the rays that need more restart in
    a resume launch with their partial hit as the limit.  Whatever the budget (1 = practically every ray
    resumes, 63 = no split), the image is the oracle's, bit for bit; the default budget sets some rays aside
    on a scene with a dense mesh and reports them in the stats."""
    L, sp, tr = sio.cornell_with_sphere(3000)
    cam = sio.make_camera(sio.CORNELL_EYE, sio.CORNELL_LOOK, sio.CORNELL_UP, 50.0, 96, 96)
    ref, _ = oracle_mod.pt_render(L, sp, tr, cam, 96, 96, 4, 4, seed=3)
    with hpt.Scene(L, sp, tr) as scene:
        fractions = {}
        for budget in (0, 1, 2, 3, 12, 63):
            p = hpt.make_params(seed=3, flags=hpt.FLAG_TIME_KERNELS)
            p.reserved = budget << 1
            img = scene.render_pt(cam, 96, 96, 4, 4, p)
            st = scene.stats()
            assert_parity(img, ref)
            fractions[budget] = st["long_rays_last_pass"] / max(st["traced_rays_last_pass"], 1)
            assert st["split_budget"] == (hpt_default_budget() if budget == 0 else (0 if budget == 63 else budget))
            assert (st["n_resume"] > 0) == (budget != 63)
        assert fractions[63] == 0.0 and fractions[1] > 0.9
        assert 0.0 < fractions[0] < 0.5 and fractions[12] < fractions[0] <= fractions[3] <= fractions[1]
        # counting renders use the plain traversal, so the work counts do not depend on the split
        p = hpt.make_params(seed=3, flags=hpt.FLAG_COUNT_WORK)
        scene.render_pt(cam, 96, 96, 4, 4, p)
        assert scene.stats()["split_budget"] == 0


def hpt_default_budget():
    return 6          # kTraceBudget, csrc/pt_kernels.h


def test_resume_launch_layouts(hpt, sio, oracle_mod):
    """The resume launch walks the four-wide twin of the tree, 12 stack levels per lane in LDS and deeper ones in global
    memory.  Development switches force the other forms -- two levels in LDS (flags bit 19: every long ray then uses
    the global-memory levels), the binary walk (bit 20) with 12 levels, the whole stack (bit 18) or two levels in LDS --
    on a mesh of 30 000 triangles and on the incoherent random-triangle cloud, with the default budget and with a
    budget of 1 (practically every ray resumes); every form is the oracle's image bit for bit."""
    W18, W19, W20 = 1 << 18, 1 << 19, 1 << 20
    for L, sp, tr in (sio.cornell_with_sphere(30000), sio.cornell_random_triangles(20000)):
        cam = sio.make_camera(sio.CORNELL_EYE, sio.CORNELL_LOOK, sio.CORNELL_UP, 50.0, 80, 64)
        bvh = hpt.export_bvh_host(L, sp, tr)
        assert bvh["bvh_depth"] > 12
        ref, _ = oracle_mod.pt_render(L, sp, tr, cam, 80, 64, 4, 3, seed=9, bvh=bvh)
        with hpt.Scene(L, sp, tr) as scene:
            for dev in (0, W19, W20, W20 | W18, W20 | W19):
                for budget in (6, 1):
                    p = hpt.make_params(seed=9, flags=dev | hpt.FLAG_TIME_KERNELS)
                    p.reserved = budget << 1
                    img = scene.render_pt(cam, 80, 64, 4, 3, p)
                    assert scene.stats()["n_resume"] > 0
                    assert_parity(img, ref)


def test_split_is_kept_for_scenes_whose_rays_are_all_long(hpt, sio, oracle_mod):
    """The renderer reads back (asynchronously) how many rays the split set aside.  Round 2 dropped the split for the
    next frames of a scene where that is most of them; with the four-wide resume launch the split is the faster form
    there too, so it stays on -- frame after frame, same image, and the share is reported."""
    L, sp, tr = sio.cornell_random_triangles(4000)
    cam = sio.make_camera(sio.CORNELL_EYE, sio.CORNELL_LOOK, sio.CORNELL_UP, 50.0, 64, 64)
    ref, _ = oracle_mod.pt_render(L, sp, tr, cam, 64, 64, 4, 2, seed=5)
    with hpt.Scene(L, sp, tr) as scene:
        a = scene.render_pt(cam, 64, 64, 4, 2, hpt.make_params(seed=5))
        st = scene.stats()
        assert st["split_budget"] == hpt_default_budget()
        assert st["long_rays_last_pass"] * 2 > st["traced_rays_last_pass"]
        b = scene.render_pt(cam, 64, 64, 4, 2, hpt.make_params(seed=5))
        assert scene.stats()["split_budget"] == hpt_default_budget()
        p = hpt.make_params(seed=5); p.reserved = 63 << 1               # 63 = no split (development)
        c = scene.render_pt(cam, 64, 64, 4, 2, p)
        assert scene.stats()["split_budget"] == 0
    assert_parity(a, ref); assert_parity(b, ref); assert_parity(c, ref)


def test_one_shot_wrappers_keep_the_scene_between_calls(hpt, sio, oracle_mod):
    """pt_render_wrapper is called once per frame by the reference's interactive front-end (src/main.cpp:416).
    A second call with byte-identical arrays reuses the uploaded scene (no BVH build, no allocation); a changed
    triangle rebuilds.  The images are those of a fresh scene either way."""
    import time
    L, sp, tr = sio.cornell_with_sphere(20000)
    cam = sio.make_camera(sio.CORNELL_EYE, sio.CORNELL_LOOK, sio.CORNELL_UP, 50.0, 64, 64)
    hpt.wrapper_cache_clear()
    t0 = time.perf_counter(); a = hpt.pt_render_wrapper(L, sp, tr, cam, 64, 64, 4, 2, seed=4); t1 = time.perf_counter()
    b = hpt.pt_render_wrapper(L, sp, tr, cam, 64, 64, 4, 2, seed=4); t2 = time.perf_counter()
    with hpt.Scene(L, sp, tr) as scene:
        fresh = scene.render_pt(cam, 64, 64, 4, 2, hpt.make_params(seed=4))
    assert np.array_equal(a, fresh) and np.array_equal(b, fresh)
    assert (t2 - t1) < 0.5 * (t1 - t0)                     # the 20 000-triangle BVH build is gone
    tr2 = tr.copy()
    tr2["v0"][-1] = np.asarray(tr2["v0"][-1]) + np.float32(0.25)       # move one corner of the last triangle
    c = hpt.pt_render_wrapper(L, sp, tr2, cam, 64, 64, 4, 2, seed=4)
    with hpt.Scene(L, sp, tr2) as scene:
        fresh2 = scene.render_pt(cam, 64, 64, 4, 2, hpt.make_params(seed=4))
    assert np.array_equal(c, fresh2)
    d = hpt.pt_render_wrapper(L, sp, tr, cam, 64, 64, 4, 2, seed=5)
    assert not np.array_equal(d, a)                        # another seed, another image
    hpt.wrapper_cache_clear()
    e = hpt.pt_render_wrapper(L, sp, tr, cam, 64, 64, 4, 2, seed=4)
    assert np.array_equal(e, fresh)
    hpt.wrapper_cache_clear()


def test_two_pipelines_render_the_same_image_as_one(hpt, sio):
    """Two passes of a render are in flight at a time (two streams, a workspace each); the per-pixel sums are
    still added in sample order, so the image is that of HPT_FLAG_SINGLE_PIPELINE -- for an even and an odd
    number of passes, at a size where the passes are large (8 Mi slots in two passes)."""
    L, sp, tr = sio.cornell_with_sphere(5000)
    cam = sio.make_camera(sio.CORNELL_EYE, sio.CORNELL_LOOK, sio.CORNELL_UP, 50.0, 256, 256)
    with hpt.Scene(L, sp, tr) as scene:
        one = scene.render_pt(cam, 256, 256, 4, 7, hpt.make_params(seed=2, samples_per_pass=2, flags=hpt.FLAG_SINGLE_PIPELINE))
        for spass in (1, 2, 3, 4, 7):
            two = scene.render_pt(cam, 256, 256, 4, 7, hpt.make_params(seed=2, samples_per_pass=spass, flags=hpt.FLAG_TIME_KERNELS))
            assert np.array_equal(one, two), spass
        big = sio.make_camera(sio.CORNELL_EYE, sio.CORNELL_LOOK, sio.CORNELL_UP, 50.0, 1024, 1024)
        a = scene.render_pt(big, 1024, 1024, 4, 8, hpt.make_params(seed=2, flags=hpt.FLAG_SINGLE_PIPELINE))
        b = scene.render_pt(big, 1024, 1024, 4, 8, hpt.make_params(seed=2, samples_per_pass=4))           # 2 x 4 Mi slots
        c = scene.render_pt(big, 1024, 1024, 4, 8, hpt.make_params(seed=2, samples_per_pass=3, flags=hpt.FLAG_RUSSIAN_ROULETTE))
        d = scene.render_pt(big, 1024, 1024, 4, 8, hpt.make_params(seed=2, flags=hpt.FLAG_RUSSIAN_ROULETTE | hpt.FLAG_SINGLE_PIPELINE))
        # a render that fits one pass (8 Mi slots; also an odd sample count) is cut into two concurrent half passes by
        # default -- the share of one rank of a multi-GPU render is such a render
        e = scene.render_pt(big, 1024, 1024, 4, 8, hpt.make_params(seed=2, flags=hpt.FLAG_TIME_KERNELS))
        n_two = scene.stats()["n_shade"]
        scene.render_pt(big, 1024, 1024, 4, 8, hpt.make_params(seed=2, flags=hpt.FLAG_TIME_KERNELS | hpt.FLAG_SINGLE_PIPELINE))
        n_one = scene.stats()["n_shade"]
        f = scene.render_pt(big, 1024, 1024, 4, 7, hpt.make_params(seed=2))
        g = scene.render_pt(big, 1024, 1024, 4, 7, hpt.make_params(seed=2, flags=hpt.FLAG_SINGLE_PIPELINE))
    assert np.array_equal(a, b) and np.array_equal(c, d) and not np.array_equal(a, c)
    assert np.array_equal(a, e) and np.array_equal(f, g) and n_two == 2 * n_one


def test_bdpt_virtual_ranks_assemble_bitwise(hpt, sio, oracle_mod):
    """The bidirectional estimator shards by image tile like the PT path: G virtual ranks rendered back to
    back and un-tiled give the single-device image bit for bit (every rank traces the same light subpaths
    from the shared seed)."""
    import torch
    sc = sio.load_scene(os.path.join(GOLDEN, "scenes", "input.txt"))
    L, sp, tr = sio.flatten_for_pt(sc)
    order = sio.object_order(sc)
    W, H, spp, spl = 52, 40, 2, 4
    cam = sio.make_camera(sc.eye, sc.look_at, sc.view_up, sc.fov, W, H, tan_in_float=True)
    stream = torch.cuda.current_stream().cuda_stream
    with hpt.Scene(L, sp, tr) as scene:
        scene.set_groups(*order)
        ref = scene.render_bdpt(cam, W, H, 4, 4, spp, spl, hpt.make_params(seed=21))
        for world, tile in ((2, 32), (3, 16), (8, 8)):
            n_local = hpt.local_pixels(W, H, hpt.make_params(world=world, tile=tile))
            gathered = torch.zeros((world, n_local, 3), dtype=torch.float32, device="cuda")
            for r in range(world):
                p = hpt.make_params(seed=21, rank=r, world=world, tile=tile)
                scene.render_bdpt_device(cam, W, H, 4, 4, spp, spl, p, gathered[r].data_ptr(), stream)
            image = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
            hpt.untile(gathered.data_ptr(), image.data_ptr(), W, H, hpt.make_params(world=world, tile=tile), stream)
            torch.cuda.synchronize()
            assert np.array_equal(image.cpu().numpy(), ref), "world=%d" % world


def _random_scene(sio, seed):
    """Seeded random scene in the Cornell box: 40-200 triangles of every material class (diffuse, rough
    dielectric-opaque, rough and mirror conductors, glass), 0-3 spheres, 1-3 lights (cone, point-like,
    parallel)."""
    from path_tracing_amd.layouts import SPHERE
    rng = np.random.default_rng(seed)
    palette = [(0.7, 0.7, 0.7, 1.0, 0.0, 0.0), (0.8, 0.3, 0.2, 0.5, 0.0, 0.0), (0.9, 0.8, 0.3, 0.3, 0.9, 0.0),
               (0.95, 0.95, 0.95, 0.0, 1.0, 0.0), (1.0, 1.0, 1.0, 0.0, 0.0, 1.5), (0.2, 0.6, 0.9, 0.05, 0.0, 0.0)]
    rows = [t for _, tl in sio._CORNELL_WALLS for t in tl]
    mats = [m6 for m6, tl in sio._CORNELL_WALLS for _ in tl]
    n = int(rng.integers(40, 200))
    c = rng.uniform([-0.4, -0.4, -0.1], [0.4, 0.4, 0.9], size=(n, 1, 3))
    v = (c + rng.uniform(-0.12, 0.12, size=(n, 3, 3))).reshape(n, 9)
    rows += [tuple(r) for r in v.astype(np.float32)]
    mats += [palette[int(k)] for k in rng.integers(0, len(palette), size=n)]
    tris = sio._tris_from(rows, mats)
    ns = int(rng.integers(0, 4))
    spheres = np.zeros(ns, SPHERE)
    for k in range(ns):
        m = palette[int(rng.integers(0, len(palette)))]
        spheres[k]["center"] = rng.uniform([-0.3, -0.3, 0.0], [0.3, 0.3, 0.8]); spheres[k]["r"] = rng.uniform(0.05, 0.15)
        spheres[k]["mtl"]["base_color"] = m[0:3]; spheres[k]["mtl"]["roughness"] = m[3]
        spheres[k]["mtl"]["metallic"] = m[4]; spheres[k]["mtl"]["eta"] = m[5]; spheres[k]["id"] = k
    lights = []
    for k in range(int(rng.integers(1, 4))):
        kind = int(rng.integers(0, 3))
        pos = tuple(rng.uniform([-0.3, 0.2, 0.0], [0.3, 0.45, 0.8]))
        if kind == 2:
            lights.append(sio._one_light(pos, (0.2, -1.0, 0.1), (0.6, 0.6, 0.6), 0.0, 1, 0.05))      # parallel
        else:
            lights.append(sio._one_light(pos, (0.0, -1.0, 0.1), tuple(rng.uniform(0.5, 1.5, size=3)), 180.0 if kind == 0 else 40.0, 0, float(rng.uniform(0.03, 0.1))))
    return np.concatenate(lights), spheres, tris


@pytest.mark.parametrize("seed", [101, 102, 103, 104, 105, 106])
def test_random_scenes_match_the_oracle(hpt, sio, oracle_mod, seed):
    """Unstructured scenes with every material and light class: the default render path (split trace steps,
    two pipelines when there are two passes) equals the oracle bit for bit; so does the one-shot wrapper."""
    L, sp, tr = _random_scene(sio, seed)
    W, H, depth, spp = 48, 40, 5, 3
    cam = sio.make_camera(sio.CORNELL_EYE, sio.CORNELL_LOOK, sio.CORNELL_UP, 50.0, W, H)
    ref, st = oracle_mod.pt_render(L, sp, tr, cam, W, H, depth, spp, seed=seed)
    with hpt.Scene(L, sp, tr) as scene:
        img = scene.render_pt(cam, W, H, depth, spp, hpt.make_params(seed=seed, flags=hpt.FLAG_COUNT_WORK))
        gs = scene.stats()
        two = scene.render_pt(cam, W, H, depth, spp, hpt.make_params(seed=seed, samples_per_pass=2))
    assert_parity(img, ref)
    assert np.array_equal(img, ref) and np.array_equal(two, ref)
    assert gs["closest_rays"] == st["closest_rays"] and gs["shadow_rays"] == st["shadow_rays"]
    one_shot = hpt.pt_render_wrapper(L, sp, tr, cam, W, H, depth, spp, seed=seed)
    assert np.array_equal(one_shot, ref)
    hpt.wrapper_cache_clear()


@pytest.mark.parametrize("seed", [201, 202, 203])
def test_random_scenes_bdpt_match_the_oracle(hpt, sio, oracle_mod, seed):
    """The same kind of scene through the bidirectional estimator (one group: spheres, then triangles)."""
    L, sp, tr = _random_scene(sio, seed)
    W, H, spp, spl = 32, 24, 2, 2
    order = (np.concatenate([np.zeros(len(sp), np.int32), np.ones(len(tr), np.int32)]),
             np.concatenate([np.arange(len(sp), dtype=np.int32), np.arange(len(tr), dtype=np.int32)]),
             np.zeros(len(sp) + len(tr), np.int32))
    ref = oracle_mod.bdpt_render(L, sp, tr, order, sio.CORNELL_EYE, sio.CORNELL_LOOK, sio.CORNELL_UP, 50.0, W, H, 4, 4, spp, spl, seed=seed)
    ref = ref[0] if isinstance(ref, tuple) else ref
    cam = sio.make_camera(sio.CORNELL_EYE, sio.CORNELL_LOOK, sio.CORNELL_UP, 50.0, W, H, tan_in_float=True)
    with hpt.Scene(L, sp, tr) as scene:
        scene.set_groups(*order)
        img = scene.render_bdpt(cam, W, H, 4, 4, spp, spl, hpt.make_params(seed=seed))
        nog = None
    with hpt.Scene(L, sp, tr) as scene:            # no grouping handed over = the same single group
        nog = scene.render_bdpt(cam, W, H, 4, 4, spp, spl, hpt.make_params(seed=seed))
    assert_parity(img, ref)
    assert np.array_equal(img, ref) and np.array_equal(nog, ref)


def test_many_materials_and_lights_beyond_the_lds_staging(hpt, sio, oracle_mod):
    """More materials (> 128) and lights (> 32) than k_shade stages in LDS, an image large enough for several trips per
    workgroup (the next-event records of three trips are evaluated together) and a mix of ball, cone and parallel
    lights: the global-memory paths of the shading kernel and of its staged evaluation equal the oracle bit for bit."""
    rng = np.random.default_rng(77)
    n = 400
    rows = [t for _, tl in sio._CORNELL_WALLS for t in tl]
    mats = [m6 for m6, tl in sio._CORNELL_WALLS for _ in tl]
    c = rng.uniform([-0.4, -0.4, -0.1], [0.4, 0.4, 0.9], size=(n, 1, 3))
    v = (c + rng.uniform(-0.1, 0.1, size=(n, 3, 3))).reshape(n, 9)
    rows += [tuple(r) for r in v.astype(np.float32)]
    for k in range(n):                                             # every triangle its own material
        kind = k % 4
        base = tuple(rng.uniform(0.1, 1.0, size=3))
        mats.append(base + ((1.0, 0.0, 0.0) if kind == 0 else (float(rng.uniform(0.2, 0.8)), 0.0, 0.0) if kind == 1
                            else (float(rng.uniform(0.1, 0.5)), 0.9, 0.0) if kind == 2 else (0.0, 1.0, 0.0)))
    tris = sio._tris_from(rows, mats)
    lights = []
    for k in range(40):
        pos = tuple(rng.uniform([-0.4, 0.1, -0.1], [0.4, 0.45, 0.9]))
        if k % 5 == 4:
            lights.append(sio._one_light(pos, (0.1, -1.0, 0.2), (0.05, 0.05, 0.05), 0.0, 1, 0.03))
        else:
            lights.append(sio._one_light(pos, (0.0, -1.0, 0.0), tuple(rng.uniform(0.02, 0.1, size=3)), 180.0 if k % 2 else 50.0, 0, 0.03))
    L = np.concatenate(lights)
    sp = np.zeros(0, sio.SPHERE)
    W, H, depth, spp = 160, 120, 4, 4            # 76 800 paths per sample: 1024-path chunks, four trips per workgroup
    cam = sio.make_camera(sio.CORNELL_EYE, sio.CORNELL_LOOK, sio.CORNELL_UP, 50.0, W, H)
    ref, st = oracle_mod.pt_render(L, sp, tr := tris, cam, W, H, depth, spp, seed=5)
    with hpt.Scene(L, sp, tr) as scene:
        assert scene.stats()["n_materials"] > 128
        img = scene.render_pt(cam, W, H, depth, spp, hpt.make_params(seed=5, flags=hpt.FLAG_COUNT_WORK))
        gs = scene.stats()
        fast = scene.render_pt(cam, W, H, depth, spp, hpt.make_params(seed=5))
    assert st["shadow_rays"] > 50_000 and gs["shadow_rays"] == st["shadow_rays"]
    assert np.array_equal(img, ref) and np.array_equal(fast, ref)


def test_no_host_wait_only_enqueues(hpt, sio):
    """HPT_FLAG_NO_HOST_WAIT: the device render call enqueues every tail iteration unseen and returns while the device
    is still at work, on a scene whose mirror wall keeps paths alive past eye_depth (the default looks at a counter
    every other tail iteration and so holds the calling thread for most of the render); same image."""
    import time
    import torch
    L, sp, tr = sio.cornell_with_sphere(20000)
    W = H = 1024
    cam = sio.make_camera(sio.CORNELL_EYE, sio.CORNELL_LOOK, sio.CORNELL_UP, 50.0, W, H)
    stream = torch.cuda.current_stream().cuda_stream
    out = torch.zeros((W * H, 3), dtype=torch.float32, device="cuda")
    with hpt.Scene(L, sp, tr) as scene:
        images, host_ms, dev_ms, pending = [], [], [], []
        for flags in (0, hpt.FLAG_NO_HOST_WAIT):
            p = hpt.make_params(seed=4, max_delta=12, flags=flags)
            scene.render_pt_device(cam, W, H, 4, 32, p, out.data_ptr(), stream)          # warm-up: workspace allocation
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            scene.render_pt_device(cam, W, H, 4, 32, p, out.data_ptr(), stream)
            pending.append(not torch.cuda.current_stream().query())      # is the device still at work when the call returns?
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            images.append(out.cpu().numpy().copy()); host_ms.append((t1 - t0) * 1e3); dev_ms.append((t2 - t0) * 1e3)
    assert np.array_equal(images[0], images[1])
    assert images[0].mean() > 0
    # the bidirectional estimator honours the flag too (glass and mirror spheres of input.txt: tail iterations exist)
    sc = sio.load_scene(os.path.join(GOLDEN, "scenes", "input.txt"))
    Lb, spb, trb = sio.flatten_for_pt(sc)
    camb = sio.camera_for(sc, 96, 64)
    with hpt.Scene(Lb, spb, trb) as scene:
        scene.set_groups(*sio.object_order(sc))
        a = scene.render_bdpt(camb, 96, 64, 4, 4, 3, 4, hpt.make_params(seed=6, max_delta=6))
        b = scene.render_bdpt(camb, 96, 64, 4, 4, 3, 4, hpt.make_params(seed=6, max_delta=6, flags=hpt.FLAG_NO_HOST_WAIT))
    assert np.array_equal(a, b) and a.mean() > 0
    # functional gate: the blind call has returned while its stream still had work queued (the render takes ~20 ms, the
    # enqueue about one).  The wall-clock split is recorded, not asserted: it depends on the box's load.
    assert pending[1], (host_ms, dev_ms)
    print("no_host_wait: host ms (default, blind) = %s, host+device ms = %s" % (host_ms, dev_ms))


def test_bdpt_connection_work_counts(hpt, sio, oracle_mod):
    """HPT_FLAG_COUNT_WORK on the bidirectional path: the connection stage reports its candidate pairs, the pairs that pass
    the culls, the shadow rays it traces and their BVH work (the numerator of the BDPT roofline, scripts/bench_configs.py).
    The shadow-ray count is checked against an independent one: the oracle's restated cpu_bdpt loop counts a connection
    exactly where it calls its visibility test (reference src/cpu_bdpt.cpp:417-421); image unchanged by counting."""
    sc = sio.load_scene(os.path.join(GOLDEN, "scenes", "input.txt"))
    L, sp, tr = sio.flatten_for_pt(sc)
    W, H, spp, spl = 48, 40, 2, 4
    cam = sio.make_camera(sc.eye, sc.look_at, sc.view_up, sc.fov, W, H, tan_in_float=True)
    ref, so = oracle_mod.bdpt_render(L, sp, tr, sio.object_order(sc), sc.eye, sc.look_at, sc.view_up, sc.fov, W, H, 4, 4, spp, spl, seed=8)
    with hpt.Scene(L, sp, tr) as scene:
        scene.set_groups(*sio.object_order(sc))
        plain = scene.render_bdpt(cam, W, H, 4, 4, spp, spl, hpt.make_params(seed=8))
        counted = scene.render_bdpt(cam, W, H, 4, 4, spp, spl, hpt.make_params(seed=8, flags=hpt.FLAG_COUNT_WORK))
        st = scene.stats()
    assert np.array_equal(plain, counted)
    assert_parity(plain, ref)
    n_lv = len(L) * spl * 4
    assert st["bd_pairs"] % n_lv == 0 and st["bd_pairs"] > 0                   # every connected eye vertex meets every light vertex
    assert st["bd_pairs"] >= st["bd_survivors"] >= st["bd_shadow_rays"] >= st["bd_unoccluded"] > 0
    assert st["bd_shadow_rays"] == so["connections"] == so["shadow_rays"]
    assert st["bd_group_boxes"] >= st["bd_shadow_rays"] and st["bd_nodes"] > 0 and st["bd_tris"] > 0
