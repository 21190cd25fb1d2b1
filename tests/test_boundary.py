"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/hpt.h declares, and argument errors are reported through return codes (no GPU work)."""
import ctypes as C
import os
import re

import numpy as np

from conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "hpt.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hpt_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(hpt):
    lib = hpt.load_library()
    names = _declared_symbols()
    assert "hpt_render_pt" in names and "hpt_pt_render_wrapper" in names and len(names) >= 12
    for n in names:
        assert hasattr(lib, n), "libhpt.so does not export %s" % n


def test_params_struct_matches_header(hpt):
    assert C.sizeof(hpt.Params) == 40
    assert hpt.Stats.ms_total.offset == 64 and hpt.Stats.bvh_nodes.offset == 120
    assert hpt.Stats.ms_resume.offset == 216 and C.sizeof(hpt.Stats) == 312      # static_assert in csrc/hpt_api.cpp


def test_local_pixels_and_tiling_rules(hpt):
    p = hpt.make_params(world=1)
    assert hpt.local_pixels(1024, 1024, p) == 1024 * 1024
    assert hpt.local_pixels(50, 37, p) == 2 * 2 * 32 * 32
    p8 = hpt.make_params(world=8, rank=3)
    assert hpt.local_pixels(1024, 1024, p8) == 1024 * 1024 // 8
    p3 = hpt.make_params(world=3, rank=2, tile=16)
    assert hpt.local_pixels(100, 60, p3) == ((7 * 4 + 2) // 3) * 256
    bad = hpt.make_params(tile=12)
    lib = hpt.load_library()
    assert lib.hpt_local_pixels(64, 64, C.byref(bad)) == -1
    assert b"multiple of 8" in lib.hpt_last_error()
    bad = hpt.make_params(world=2, rank=2)
    assert lib.hpt_local_pixels(64, 64, C.byref(bad)) == -1


def test_argument_errors_are_return_codes(hpt):
    lib = hpt.load_library()
    assert lib.hpt_scene_create(None, 0, None, 0, None, 3, C.byref(C.c_void_p())) != 0
    assert b"null primitive array" in lib.hpt_last_error()
    assert lib.hpt_scene_create(None, -1, None, 0, None, 0, C.byref(C.c_void_p())) != 0
    assert lib.hpt_render_pt(None, None, 8, 8, 4, 1, None, None) != 0
    assert lib.hpt_get_stats(None, None) != 0
    lib.hpt_scene_destroy(None)     # no-op, like free(NULL)


def test_host_tiling_mirror_roundtrip():
    from path_tracing_amd import tiling
    rng = np.random.default_rng(0)
    for (W, H, tile, world) in ((50, 37, 32, 1), (64, 64, 8, 4), (100, 60, 16, 3), (33, 9, 8, 8)):
        img = rng.random((H, W, 3)).astype(np.float32)
        locs = np.stack([tiling.tile_image(img, tile, r, world) for r in range(world)])
        assert locs.shape[1] == tiling.tiling_dims(W, H, tile, world)[3]
        back = tiling.untile_image(locs, W, H, tile, world)
        assert np.array_equal(back, img)
        # every pixel belongs to exactly one rank
        cover = np.zeros((H, W), np.int32)
        for r in range(world):
            x, y, v = tiling.local_to_pixel(W, H, tile, r, world)
            np.add.at(cover, (y[v], x[v]), 1)
        assert (cover == 1).all()


REF_SYMBOL = "_Z17pt_render_wrapperPK9CudaLightiPK10CudaSphereiPK12CudaTrianglei6float3S8_10CudaCameraPS8_iiiiii"


def test_reference_named_adapter_exports_the_reference_symbol(tmp_path):
    """libhpt_ref.so must define pt_render_wrapper under the exact mangled name a caller compiled
    against the reference's declaration (include/pt_cu.cuh:6-13) asks for.  The expected name is
    derived independently here: a probe translation unit that declares the function from scratch
    (opaque record types, CUDA-style float3) is compiled and its undefined symbol is read back."""
    import subprocess
    so = os.path.join(ROOT, "path_tracing_amd", "csrc", "libhpt_ref.so")
    assert os.path.exists(so)
    probe = tmp_path / "probe.cpp"
    probe.write_text(
        "struct float3 { float x, y, z; };\n"
        "struct CudaLight; struct CudaSphere; struct CudaTriangle;\n"
        "struct CudaCamera { float3 eye, U, V, W, UL, dx, dy; };\n"
        "void pt_render_wrapper(const CudaLight *, int, const CudaSphere *, int, const CudaTriangle *, int,\n"
        "                       float3, float3, const CudaCamera, float3 *, int, int, int, int, int, int);\n"
        "void call(const CudaLight *l, const CudaSphere *s, const CudaTriangle *t, CudaCamera c, float3 *img){\n"
        "    float3 z = {0, 0, 0}; pt_render_wrapper(l, 1, s, 1, t, 1, z, z, c, img, 8, 8, 4, 8, 4, 1); }\n")
    obj = tmp_path / "probe.o"
    subprocess.check_call(["g++", "-c", "-o", str(obj), str(probe)])
    undefined = subprocess.check_output(["nm", "-u", str(obj)], text=True)
    wanted = [ln.split()[-1] for ln in undefined.splitlines() if "pt_render_wrapper" in ln]
    assert wanted == [REF_SYMBOL]
    defined = subprocess.check_output(["nm", "-D", "--defined-only", so], text=True)
    assert REF_SYMBOL in defined
    # bdpt_render_wrapper (reference include/bdpt_cu.cuh:30-37): same argument list plus the trailing spl
    assert "_Z19bdpt_render_wrapperPK9CudaLightiPK10CudaSphereiPK12CudaTrianglei6float3S8_10CudaCameraPS8_iiiiiii" in defined
