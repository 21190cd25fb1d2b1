"""The C++ host mirror (path_tracing_amd/csrc/host/: scene grammar, flattening, camera, output
stage, CLI clone) against the Python mirror and, on the GPU, end to end through pt_cli."""
import ctypes as C
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

CSRC = os.path.join(ROOT, "path_tracing_amd", "csrc")


def _host():
    lib = C.CDLL(os.path.join(CSRC, "libhpt_host.so"))
    lib.hpt_host_flatten_scene_file.restype = C.c_int
    return lib


def _flatten_cpp(path, W, H):
    from path_tracing_amd.layouts import CAMERA, LIGHT, SPHERE, TRIANGLE
    lib = _host()
    nl, ns, nt = C.c_int(), C.c_int(), C.c_int()
    pl, ps, pt = C.c_void_p(), C.c_void_p(), C.c_void_p()
    cam = np.zeros((), CAMERA)
    res = (C.c_int * 2)()
    rc = lib.hpt_host_flatten_scene_file(path.encode(), C.byref(nl), C.byref(ns), C.byref(nt), C.byref(pl), C.byref(ps), C.byref(pt),
                                         cam.ctypes.data_as(C.c_void_p), W, H, res)
    assert rc == 0
    get = lambda p, n, dt: np.frombuffer(C.string_at(p, n * dt.itemsize), dt).copy() if n else np.zeros(0, dt)
    return get(pl, nl.value, LIGHT), get(ps, ns.value, SPHERE), get(pt, nt.value, TRIANGLE), cam, (res[0], res[1])


@pytest.mark.parametrize("name", ["input", "mis_test"])
def test_cpp_parser_flattening_and_camera_match_python_mirror(sio, name):
    path = os.path.join(GOLDEN, "scenes", name + ".txt")
    L, sp, tr, cam, res = _flatten_cpp(path, 200, 120)
    sc = sio.load_scene(path)
    L2, sp2, tr2 = sio.flatten_for_pt(sc)
    assert res == tuple(sc.resolution)
    assert L.tobytes() == L2.tobytes() and sp.tobytes() == sp2.tobytes() and tr.tobytes() == tr2.tobytes()
    cam2 = sio.camera_for(sc, 200, 120, 50.0)
    assert cam.tobytes() == cam2.tobytes()


def test_object_model_view_flattens_to_the_same_bytes(sio):
    """The reference-shaped std::map<int, AABB> of Object* (what move_data_to_cuda_pt takes) and the flat storage
    give the same boundary records."""
    from path_tracing_amd.layouts import LIGHT, SPHERE, TRIANGLE
    lib = _host()
    for name in ("input", "mis_test"):
        path = os.path.join(GOLDEN, "scenes", name + ".txt")
        nl, ns, nt = C.c_int(), C.c_int(), C.c_int()
        pl, ps, pt = C.c_void_p(), C.c_void_p(), C.c_void_p()
        assert lib.hpt_host_flatten_via_groups(path.encode(), C.byref(nl), C.byref(ns), C.byref(nt), C.byref(pl), C.byref(ps), C.byref(pt)) == 0
        d = sio.load_scene_fast(path)
        assert C.string_at(pt, nt.value * TRIANGLE.itemsize) == d["tris"].tobytes()
        assert C.string_at(ps, ns.value * SPHERE.itemsize) == d["spheres"].tobytes()
        assert C.string_at(pl, nl.value * LIGHT.itemsize) == d["lights"].tobytes()
        order = sio.object_order(sio.load_scene(path))
        for a, b in zip(order, d["order"]):
            assert np.array_equal(a, b)


def test_parser_grammar_edge_cases(tmp_path, sio):
    """Observable rules of the reference's `>>` loop (src/main_cli.cpp:99-141): unknown characters
    are dropped one at a time (stale numbers behind an M line), '//' comments run to the end of the line, numbers may
    carry a '+', a malformed number ends the parse, groups sort by id and keep file order inside."""
    text = ("// header comment with tags: T S M L\n"
            "E 0 0 -1  V 0 0 1 0 1 0\nR 64 48\nF 35.5\n"
            "M 0.5 0.25 0.125 1 0 0 9 9 9\n"                      # three stale numbers: skipped digit by digit
            "G 3\nT 0 0 0 +1 0 0 0 1e0 0\n"
            "G 1\nS .5 -.5 0.25 1.5e-1 // trailing comment T 1 2 3\n"
            "M 1 1 1 0 1 0\nT 0 0 1 1 0 1 0 1 1\n"
            "G 3\nT 0 0 2 1 0 2 0 1 2\n"
            "L 0 0.4 0 0 -1 0 1 1 1 90 0 0.05\n"
            "T 5 5 5 x 1 2 3 4 5 6\n"                             # 'x' is not a number: the parse ends here
            "T 9 9 9 9 9 9 9 9 9\n")
    path = tmp_path / "edge.txt"
    path.write_text(text)
    d = sio.load_scene_fast(str(path), W=64, H=48)
    sc = sio.parse_scene_text(text)
    assert d["resolution"] == (64, 48) and abs(d["fov"] - 35.5) < 1e-6
    assert len(d["tris"]) == 3 and len(d["spheres"]) == 1 and len(d["lights"]) == 1
    # group 1 first (sphere, triangle), then group 3 in file order
    assert list(d["order"][2]) == [1, 1, 3, 3] and list(d["order"][0]) == [0, 1, 1, 1]
    assert np.allclose(d["tris"]["v0"][:, 2], [1, 0, 2]) and np.allclose(d["tris"]["v1"][1], [1, 0, 0])
    assert np.allclose(d["spheres"]["center"][0], [0.5, -0.5, 0.25]) and np.isclose(d["spheres"]["r"][0], 0.15)
    assert np.allclose(d["tris"]["mtl"]["base_color"][1], [0.5, 0.25, 0.125]) and d["tris"]["mtl"]["metallic"][0] == 1.0
    # same records as the Python mirror of the grammar, except the partial last record the failed stream leaves behind
    # in the reference (and in the mirror): the fast parser drops it
    L2, sp2, tr2 = sio.flatten_for_pt(sc)
    assert sp2.tobytes() == d["spheres"].tobytes() and L2.tobytes() == d["lights"].tobytes()
    assert len(tr2) == 4 and tr2[np.argsort(tr2["id"])][:3].tobytes() == d["tris"][np.argsort(d["tris"]["id"])].tobytes()
    assert d["camera"].tobytes() == sio.make_camera(d["eye"], d["look_at"], d["view_up"], 50.0, 64, 48).tobytes()


def test_million_triangle_scene_text_loads_in_well_under_a_second(tmp_path, sio):
    """f2: configs[4]-sized input (1M 'T' lines, ~100 MB of text) through the mapped-file tokenizer."""
    L, sp, tr = sio.cornell_with_sphere(1_000_000)
    path = str(tmp_path / "c5.txt")
    with open(path, "w") as fh:
        fh.write(sio.scene_to_text(L, sp, tr, 4096, 4096))
    assert os.path.getsize(path) > 80e6
    d = sio.load_scene_fast(path)
    assert len(d["tris"]) == len(tr) and d["tris"].tobytes() == tr.tobytes() and d["lights"].tobytes() == L.tobytes()
    print("parse of %d triangles: %.1f ms" % (len(tr), d["parse_ms"]))
    assert d["parse_ms"] < 1000.0


def test_obj_reader(tmp_path, sio):
    L, sp, tr = sio.cornell_with_sphere(3000)
    path = str(tmp_path / "mesh.obj")
    sio.write_obj(path, tr)
    d = sio.load_scene_fast(path, obj=True)
    assert len(d["tris"]) == len(tr)
    for k in ("v0", "v1", "v2"):
        assert np.array_equal(d["tris"][k], tr[k])
    assert np.allclose(d["tris"]["mtl"]["base_color"], 0.7) and (d["tris"]["mtl"]["roughness"] == 1.0).all()
    # polygons are fanned from their first vertex; vt / vn / o / g / s / usemtl / comments are skipped
    quad = tmp_path / "quad.obj"
    quad.write_text("# c\nmtllib x.mtl\no q\nv -1 -1 0\nv 1 -1 0 1.0\nvt 0 0\nvn 0 0 1\nv 1 1 0\nv -1 1 0\ns off\nusemtl m\nf 1/1/1 2/1/1 3/1/1 4/1/1\nf -4 -2 -1\n")
    q = sio.load_scene_fast(str(quad), obj=True)
    assert len(q["tris"]) == 3
    assert np.array_equal(q["tris"]["v0"], [[-1, -1, 0]] * 3)
    assert np.array_equal(q["tris"]["v1"], [[1, -1, 0], [1, 1, 0], [1, 1, 0]]) and np.array_equal(q["tris"]["v2"], [[1, 1, 0], [-1, 1, 0], [-1, 1, 0]])
    bad = tmp_path / "bad.obj"
    bad.write_text("v 0 0 0\nf 1 2 3\n")
    with pytest.raises(IOError):
        sio.load_scene_fast(str(bad), obj=True)


def _decode_png(data):
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, ihdr = 8, b"", None
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        assert zlib.crc32(tag + body) == struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0]
        if tag == b"IHDR": ihdr = struct.unpack(">IIBBBBB", body)
        if tag == b"IDAT": idat += body
        pos += 12 + n
    W, H = ihdr[0], ihdr[1]
    assert ihdr[2:] == (8, 2, 0, 0, 0)
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(H, 1 + 3 * W)
    assert not raw[:, 0].any()
    return raw[:, 1:].reshape(H, W, 3)


def test_png_encoder(tmp_path):
    lib = _host()
    rng = np.random.default_rng(0)
    rgb = rng.integers(0, 256, (7, 5, 3), dtype=np.uint8)
    png = str(tmp_path / "o.png")
    assert lib.hpt_host_write_png_rgb8(png.encode(), rgb.ctypes.data_as(C.c_void_p), 5, 7) == 0
    assert np.array_equal(_decode_png(open(png, "rb").read()), rgb)


def test_tonemap_threshold_table_reproduces_the_host_loop(hpt):
    """The device tone-map looks bytes up in 255 thresholds computed with the host's powf (include/hpt.h, hpt_tonemap):
    count(thresholds <= x) must equal the reference's per-channel expression (src/main_cli.cpp:233-241) for every x --
    checked on random floats, on the neighbours of every threshold, and on the special values."""
    lib = hpt.load_library()
    thr = np.zeros(256, np.float32)
    lib.hpt_tonemap_table(thr.ctypes.data_as(C.c_void_p))
    assert thr[0] == -np.inf and (np.diff(thr[1:]) > 0).all() and thr[1] > 0 and thr[255] <= 1.0
    rng = np.random.default_rng(1)
    xs = [rng.random(3_000_000, dtype=np.float32), (rng.random(300_000, dtype=np.float32) ** 8).astype(np.float32),
          np.array([-1.0, -0.0, 0.0, 1e-30, 1.0, np.nextafter(np.float32(1), np.float32(0)), 1.5, np.inf, -np.inf, np.nan], np.float32)]
    for k in range(1, 256):
        t = thr[k]
        xs.append(np.array([np.nextafter(t, np.float32(-1)), t, np.nextafter(t, np.float32(2))], np.float32))
    x = np.concatenate(xs)
    x = np.concatenate([x, np.zeros((-len(x)) % 3, np.float32)])
    want = np.zeros(len(x), np.uint8)
    lib.hpt_tonemap_reference(x.ctypes.data_as(C.c_void_p), want.ctypes.data_as(C.c_void_p), C.c_int64(len(x) // 3), 0)
    got = np.searchsorted(thr[1:], x, side="right").astype(np.uint8)       # thresholds <= x; NaN sorts last in numpy:
    got[np.isnan(x)] = 0                                                   # on the device every comparison with NaN is false
    assert np.array_equal(got, want)
    # BGR order of the reference's cv::Vec3b
    px = np.array([[0.1, 0.5, 0.9]], np.float32)
    a, b = np.zeros(3, np.uint8), np.zeros(3, np.uint8)
    lib.hpt_tonemap_reference(px.ctypes.data_as(C.c_void_p), a.ctypes.data_as(C.c_void_p), C.c_int64(1), 0)
    lib.hpt_tonemap_reference(px.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), C.c_int64(1), 1)
    assert list(a) == list(b[::-1])


def test_cli_help_and_missing_scene():
    cli = os.path.join(CSRC, "pt_cli")
    out = subprocess.run([cli, "--help"], capture_output=True, text=True)
    assert out.returncode == 0 and "--spp <int>" in out.stdout and "--seed" in out.stdout and "--gpus" in out.stdout
    out = subprocess.run([cli, "--input", "/nonexistent/scene.txt"], capture_output=True, text=True)
    assert out.returncode != 0 and "[Error] Cannot open input file" in out.stderr


@pytest.mark.gpu
def test_cli_renders_the_same_image_as_the_oracle(tmp_path, sio, oracle_mod):
    cli = os.path.join(CSRC, "pt_cli")
    scene = os.path.join(GOLDEN, "scenes", "input.txt")
    out = str(tmp_path / "img.pfm")
    run = subprocess.run([cli, "--input", scene, "--output", out, "--spp", "3", "--seed", "17", "--width", "48", "--height", "40"],
                         capture_output=True, text=True)
    assert run.returncode == 0, run.stderr
    for needle in ("Mode   : pt", "Triangle:\n36", "Ball:\n5", "Light:\n4", "moved", "[Render] Finished in", "[Success] Image saved!"):
        assert needle in run.stdout
    blob = open(out, "rb").read()
    head = b"PF\n48 40\n-1.0\n"
    img = np.frombuffer(blob[len(head):], np.float32).reshape(40, 48, 3)[::-1]
    sc = sio.load_scene(scene)
    L, sp, tr = sio.flatten_for_pt(sc)
    ref, _ = oracle_mod.pt_render(L, sp, tr, sio.camera_for(sc, 48, 40), 48, 40, 4, 3, seed=17)
    assert float(np.sqrt(((img - ref) ** 2).mean())) < 1e-3 and float(np.abs(img - ref).max()) <= 1e-6


@pytest.mark.gpu
def test_cli_obj_ingestion(tmp_path):
    cli = os.path.join(CSRC, "pt_cli")
    obj = tmp_path / "quad.obj"
    obj.write_text("v -0.2 -0.2 0.5\nv 0.2 -0.2 0.5\nv 0.2 0.2 0.5\nv -0.2 0.2 0.5\nf 1 2 3 4\nf -4//1 -3//1 -2//1\n")
    out = str(tmp_path / "q.png")
    run = subprocess.run([cli, "--input", os.path.join(GOLDEN, "scenes", "input.txt"), "--obj", str(obj), "--output", out,
                          "--spp", "2", "--seed", "1", "--width", "32", "--height", "32"], capture_output=True, text=True)
    assert run.returncode == 0 and "OBJ triangles: 3" in run.stdout and "Triangle:\n39" in run.stdout
    assert os.path.getsize(out) > 100


@pytest.mark.gpu
def test_cli_bdpt_mode_matches_the_cpu_bdpt_oracle(tmp_path, sio, oracle_mod):
    """pt_cli --mode bdpt: scene file -> groups -> device BDPT; the camera is the CLI's (fov 50, src/main_cli.cpp:158)."""
    cli = os.path.join(CSRC, "pt_cli")
    scene = os.path.join(GOLDEN, "scenes", "input.txt")
    out = str(tmp_path / "bd.pfm")
    run = subprocess.run([cli, "--mode", "bdpt", "--input", scene, "--output", out, "--spp", "2", "--spl", "4", "--seed", "9",
                          "--width", "40", "--height", "32"], capture_output=True, text=True)
    assert run.returncode == 0, run.stderr
    assert "Mode   : bdpt" in run.stdout and "[Success] Image saved!" in run.stdout
    blob = open(out, "rb").read()
    head = b"PF\n40 32\n-1.0\n"
    img = np.frombuffer(blob[len(head):], np.float32).reshape(32, 40, 3)[::-1]
    sc = sio.load_scene(scene)
    L, sp, tr = sio.flatten_for_pt(sc)
    order = sio.object_order(sc)
    # the oracle derives its camera from (eye, look_at, up, fov) with a float tangent; the CLI's init_camera takes the
    # tangent in double, so compare through the GPU library with the CLI's own camera instead of the oracle's
    import path_tracing_amd as hpt
    cam = sio.camera_for(sc, 40, 32, 50.0)
    with hpt.Scene(L, sp, tr) as s:
        s.set_groups(*order)
        ref = s.render_bdpt(cam, 40, 32, 4, 4, 2, 4, hpt.make_params(seed=9))
    assert np.array_equal(img, ref)
    # and the estimator is cpu_bdpt's: same image statistics as the oracle with its own (float-tangent) camera
    orc, _ = oracle_mod.bdpt_render(L, sp, tr, order, sc.eye, sc.look_at, sc.view_up, 50.0, 40, 32, 4, 4, 2, 4, seed=9)
    assert float(np.sqrt(((img - orc) ** 2).mean())) < 1e-3


@pytest.mark.gpu
def test_device_tonemap_and_png_output(tmp_path, hpt):
    """f3: the 8-bit output stage runs on the device and gives the bytes of the reference's host loop."""
    lib = hpt.load_library()
    rng = np.random.default_rng(5)
    img = (rng.random((37, 53, 3)) * 1.4 - 0.2).astype(np.float32)
    img[0, 0] = (np.nan, np.inf, -np.inf)
    for bgr in (0, 1):
        want = np.zeros((37, 53, 3), np.uint8); got = np.zeros((37, 53, 3), np.uint8)
        lib.hpt_tonemap_reference(img.ctypes.data_as(C.c_void_p), want.ctypes.data_as(C.c_void_p), C.c_int64(37 * 53), bgr)
        assert lib.hpt_tonemap_host(img.ctypes.data_as(C.c_void_p), got.ctypes.data_as(C.c_void_p), C.c_int64(37 * 53), bgr) == 0
        assert np.array_equal(got, want)
    # a size that is not a multiple of four bytes, and one value per thresholds' neighbourhood
    thr = np.zeros(256, np.float32); lib.hpt_tonemap_table(thr.ctypes.data_as(C.c_void_p))
    edge = np.concatenate([[np.nextafter(t, np.float32(-1)), t, np.nextafter(t, np.float32(2))] for t in thr[1:]]).astype(np.float32)
    edge = edge[:len(edge) // 3 * 3][:3 * 253]
    want = np.zeros(len(edge), np.uint8); got = np.zeros(len(edge), np.uint8)
    lib.hpt_tonemap_reference(edge.ctypes.data_as(C.c_void_p), want.ctypes.data_as(C.c_void_p), C.c_int64(len(edge) // 3), 0)
    assert lib.hpt_tonemap_host(edge.ctypes.data_as(C.c_void_p), got.ctypes.data_as(C.c_void_p), C.c_int64(len(edge) // 3), 0) == 0
    assert np.array_equal(got, want)
    # end to end: write_image = device tone-map + PNG
    png = str(tmp_path / "o.png")
    assert _host().hpt_host_write_image(png.encode(), img.ctypes.data_as(C.c_void_p), 53, 37) == 0
    want = np.zeros((37, 53, 3), np.uint8)
    lib.hpt_tonemap_reference(img.ctypes.data_as(C.c_void_p), want.ctypes.data_as(C.c_void_p), C.c_int64(37 * 53), 0)
    assert np.array_equal(_decode_png(open(png, "rb").read()), want)
    pfm = str(tmp_path / "o.pfm")
    assert _host().hpt_host_write_image(pfm.encode(), img.ctypes.data_as(C.c_void_p), 53, 37) == 0
    blob = open(pfm, "rb").read()
    head = b"PF\n53 37\n-1.0\n"
    assert blob.startswith(head)
    back = np.frombuffer(blob[len(head):], np.float32).reshape(37, 53, 3)[::-1]
    assert np.array_equal(back, img, equal_nan=True)


@pytest.mark.gpu
def test_config3_scene_through_the_text_and_obj_front_end(tmp_path, hpt, sio):
    """f2: the ~100k-triangle scene of configs[2] written out in the reference's text grammar (and its sphere as a
    Wavefront OBJ), read back by the C++ front-end and rendered by pt_cli: same image as the numpy-built arrays."""
    cli = os.path.join(CSRC, "pt_cli")
    L, sp, tr = sio.cornell_with_sphere(100_000)
    W = H = 256
    cam = sio.make_camera(sio.CORNELL_EYE, sio.CORNELL_LOOK, sio.CORNELL_UP, 50.0, W, H)
    with hpt.Scene(L, sp, tr) as scene:
        ref = scene.render_pt(cam, W, H, 4, 4, hpt.make_params(seed=5))
    txt = str(tmp_path / "c3.txt")
    with open(txt, "w") as fh:
        fh.write(sio.scene_to_text(L, sp, tr, W, H))
    d = sio.load_scene_fast(txt, W=W, H=H)
    assert d["tris"].tobytes() == tr.tobytes() and d["lights"].tobytes() == L.tobytes() and d["camera"].tobytes() == cam.tobytes()
    out = str(tmp_path / "c3.pfm")
    run = subprocess.run([cli, "--input", txt, "--output", out, "--spp", "4", "--seed", "5"], capture_output=True, text=True)
    assert run.returncode == 0, run.stderr
    assert "Triangle:\n%d" % len(tr) in run.stdout
    head = b"PF\n%d %d\n-1.0\n" % (W, H)
    img = np.frombuffer(open(out, "rb").read()[len(head):], np.float32).reshape(H, W, 3)[::-1]
    assert np.array_equal(img, ref)
    # the walls as text, the sphere as an OBJ (group 2, 0.7 grey diffuse = the sphere's material): same triangles
    walls = str(tmp_path / "walls.txt")
    with open(walls, "w") as fh:
        fh.write(sio.scene_to_text(L, sp, tr[:12], W, H))
    obj = str(tmp_path / "sphere.obj")
    sio.write_obj(obj, tr[12:])
    out2 = str(tmp_path / "c3obj.pfm")
    run = subprocess.run([cli, "--input", walls, "--obj", obj, "--output", out2, "--spp", "4", "--seed", "5"], capture_output=True, text=True)
    assert run.returncode == 0, run.stderr
    assert "OBJ triangles: %d" % (len(tr) - 12) in run.stdout
    img2 = np.frombuffer(open(out2, "rb").read()[len(head):], np.float32).reshape(H, W, 3)[::-1]
    assert np.array_equal(img2, ref)


@pytest.mark.gpu
def test_cli_gpus_flag_fans_out_inside_the_blocking_call(tmp_path, hpt):
    cli = os.path.join(CSRC, "pt_cli")
    scene = os.path.join(GOLDEN, "scenes", "input.txt")
    n = hpt.device_count()
    outs = []
    for gpus in (1, n):
        out = str(tmp_path / ("g%d.pfm" % gpus))
        run = subprocess.run([cli, "--input", scene, "--output", out, "--spp", "2", "--seed", "3", "--width", "64", "--height", "48",
                              "--gpus", str(gpus)], capture_output=True, text=True)
        assert run.returncode == 0 and "[Success] Image saved!" in run.stdout, run.stderr
        outs.append(open(out, "rb").read())
    assert outs[0] == outs[1]
    # more devices than the node has: reported, no image
    run = subprocess.run([cli, "--input", scene, "--output", str(tmp_path / "x.pfm"), "--spp", "1", "--gpus", str(n + 1)], capture_output=True, text=True)
    assert "device id outside" in run.stderr


def test_parser_survives_mutated_inputs(tmp_path):
    """The mapped-file tokenizer reads whatever bytes it is given: 300 mutated copies of the reference's scene files and
    of an OBJ snippet (deleted spans, injected tags and malformed numbers, random bytes, truncation) go through the scene
    and the OBJ reader in a child process, which must neither crash nor hang."""
    import random
    import subprocess
    import sys
    rng = random.Random(5)
    seeds = [open(os.path.join(GOLDEN, "scenes", f), "rb").read() for f in ("input.txt", "mis_test.txt")]
    seeds.append(b"v 0 0 0\nv 1 0 0\nv 0 1 0\nv 1 1 0\nf 1 2 3 4\nf 1/1/1 2/2/2 3/3/3\nf -1 -2 -3\n")
    tokens = [b"T", b"S", b"L", b"M", b"R", b"E", b"V", b"U", b"F", b"G", b"v", b"f", b"#", b"-", b"1e40", b"nan", b"inf", b"-0", b"1e-50",
              b"0x10", b"+5", b".", b"e", b"1.2.3", b"999999999999999999999", b"-2147483649", b"/", b"//", b"1//2", b"\t", b"\r\n", b"\x00", b"\xff"]

    def mutate(b):
        b = bytearray(b)
        for _ in range(rng.randint(1, 8)):
            op = rng.randint(0, 5)
            pos = rng.randint(0, max(0, len(b) - 1)) if b else 0
            if op == 0 and b:
                del b[pos:pos + rng.randint(1, 40)]
            elif op == 1:
                b[pos:pos] = rng.choice(tokens) + b" "
            elif op == 2 and b:
                b[pos] = rng.randint(0, 255)
            elif op == 3:
                b[pos:pos] = bytes(rng.randint(0, 255) for _ in range(rng.randint(1, 10)))
            elif op == 4 and b:
                b = b[:pos]
            else:
                b[pos:pos] = b"\n" + rng.choice(tokens) + b" " + b" ".join(rng.choice(tokens) for _ in range(rng.randint(0, 12))) + b"\n"
        return bytes(b)

    paths = []
    for k in range(300):
        p = tmp_path / ("m%d.txt" % k)
        p.write_bytes(mutate(rng.choice(seeds)) if k % 10 else bytes(rng.randint(0, 255) for _ in range(rng.randint(0, 300))))
        paths.append(str(p))
    child = ("import sys\nsys.path.insert(0, %r)\nfrom path_tracing_amd import scene_io as S\n"
             "for p in sys.argv[1:]:\n    for obj in (False, True):\n        try:\n            S.load_scene_fast(p, obj=obj, W=16, H=8)\n"
             "        except IOError:\n            pass\nprint('survived')\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    run = subprocess.run([sys.executable, "-c", child] + paths, capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and "survived" in run.stdout, (run.returncode, run.stderr[-500:])
