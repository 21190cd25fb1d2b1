"""Scene text grammar, flattening and camera set-up for the path-tracing hot path.

Python mirror of what the reference's front-end does *before* it reaches the
drop-in boundary, used by tests and bench.py to produce inputs in the boundary's
layouts (the C++ mirror used by the CLI lives in csrc/host/):

* parse_scene_text   -- reference: src/main_cli.cpp:99-141 (token-by-token grammar,
                        SURVEY.md Appendix A)
* flatten_for_pt     -- reference: src/pt_cu_helper.cpp:12-64 (move_data_to_cuda_pt)
* flatten_for_bdpt   -- reference: src/bdpt_cu_helper.cpp:13-70
* make_camera        -- reference: src/main_cli.cpp:25-40 (init_camera) + :162-166
* synthetic scenes   -- SURVEY.md section 8(d): S2 (diffuse Cornell), S3/S5 (Cornell +
                        tessellated sphere), random-triangle stress scene
"""
from __future__ import annotations

import math
import re
from dataclasses import dataclass, field

import numpy as np

from .layouts import (CAMERA, LIGHT, MAT_CONDUCTOR, MAT_DIELECTRIC, MAT_UBER, SPHERE, TRIANGLE)

f32 = np.float32
_FLOAT_RE = re.compile(r"[+-]?(?:\d+\.?\d*(?:[eE][+-]?\d+)?|\.\d+(?:[eE][+-]?\d+)?)")
_INT_RE = re.compile(r"[+-]?\d+")


@dataclass
class SceneDesc:
    eye: np.ndarray = field(default_factory=lambda: np.zeros(3, f32))
    look_at: np.ndarray = field(default_factory=lambda: np.zeros(3, f32))
    view_up: np.ndarray = field(default_factory=lambda: np.array([0, 1, 0], f32))
    fov: float = 50.0
    resolution: tuple = (200, 200)
    # groups[gid] = list of ("S", center(3), r, mat(6), obj_id) / ("T", verts(3,3), mat(6), obj_id)
    groups: dict = field(default_factory=dict)
    lights: list = field(default_factory=list)     # dicts: pos, dir, illum, cutoff_deg, is_parallel, ball_r


class _Stream:
    """Minimal std::istream emulation: operator>>(char) and operator>>(float/int)."""

    def __init__(self, text: str):
        self.s = text
        self.i = 0
        self.ok = True

    def _skip_ws(self):
        n = len(self.s)
        while self.i < n and self.s[self.i].isspace():
            self.i += 1

    def get_char(self):
        if not self.ok:
            return None
        self._skip_ws()
        if self.i >= len(self.s):
            self.ok = False
            return None
        c = self.s[self.i]
        self.i += 1
        return c

    def get_float(self):
        if not self.ok:
            return f32(0)
        self._skip_ws()
        m = _FLOAT_RE.match(self.s, self.i)
        if not m:
            self.ok = False
            return f32(0)
        self.i = m.end()
        return f32(m.group(0))

    def get_int(self):
        if not self.ok:
            return 0
        self._skip_ws()
        m = _INT_RE.match(self.s, self.i)
        if not m:
            self.ok = False
            return 0
        self.i = m.end()
        return int(m.group(0))

    def get_vec3(self):
        return np.array([self.get_float(), self.get_float(), self.get_float()], f32)

    def skip_line(self):
        j = self.s.find("\n", self.i)
        self.i = len(self.s) if j < 0 else j + 1


def parse_scene_text(text: str) -> SceneDesc:
    """Token grammar of the reference CLI (src/main_cli.cpp:99-141); GROUPING == 1."""
    sc = SceneDesc()
    st = _Stream(text)
    mat = np.zeros(6, f32)
    gid = 0
    obj_id = 0
    while True:
        t = st.get_char()
        if t is None:
            break
        if t == "E":
            sc.eye = st.get_vec3()
        elif t == "V":
            sc.look_at = st.get_vec3()
            sc.view_up = st.get_vec3()
        elif t == "F":
            sc.fov = float(st.get_float())
        elif t == "R":
            w = st.get_int()
            h = st.get_int()
            sc.resolution = (w, h)
        elif t == "S":
            c = st.get_vec3()
            r = st.get_float()
            sc.groups.setdefault(gid, []).append(("S", c, r, mat.copy(), obj_id))
            obj_id += 1
        elif t == "T":
            v = np.stack([st.get_vec3(), st.get_vec3(), st.get_vec3()])
            sc.groups.setdefault(gid, []).append(("T", v, mat.copy(), obj_id))
            obj_id += 1
        elif t == "M":
            mat = np.array([st.get_float() for _ in range(6)], f32)
        elif t == "G":
            gid = st.get_int()
        elif t == "/":
            t2 = st.get_char()
            if t2 == "/":
                st.skip_line()
        elif t == "L":
            pos = st.get_vec3()
            d = st.get_vec3()
            illum = st.get_vec3()
            cutoff_deg = st.get_float()
            is_par = st.get_int()
            ball_r = st.get_float()
            sc.lights.append(dict(pos=pos, dir=d, illum=illum, cutoff_deg=cutoff_deg,
                                  is_parallel=is_par, ball_r=ball_r))
        # any other character is skipped, one at a time
    return sc


def load_scene(path: str) -> SceneDesc:
    with open(path, "r", encoding="utf-8", errors="replace") as fh:
        return parse_scene_text(fh.read())


def _mat_type(eta: float, metallic: float) -> int:
    # reference: src/geometric.cu:41-49
    if eta > 0.0:
        return MAT_DIELECTRIC
    if metallic > 0.0:
        return MAT_CONDUCTOR
    return MAT_UBER


def _fill_mtl(rec, mat6):
    rec["mtl"]["base_color"] = mat6[0:3]
    rec["mtl"]["roughness"] = mat6[3]
    rec["mtl"]["metallic"] = mat6[4]
    rec["mtl"]["eta"] = mat6[5]
    rec["mtl"]["type"] = _mat_type(float(mat6[5]), float(mat6[4]))


def _radians(deg: np.float32) -> np.float32:
    return f32(deg) * f32(0.01745329251994329576923690768489)     # glm::radians<float>


def flatten(scene: SceneDesc, illum_divisor: float = 1.0):
    """Groups in map order, objects in insertion order, spheres and triangles into
    separate arrays; light directions normalised (normalize_cuda)."""
    spheres, tris = [], []
    for gid in sorted(scene.groups):
        for obj in scene.groups[gid]:
            if obj[0] == "S":
                rec = np.zeros((), SPHERE)
                rec["center"] = obj[1]
                rec["r"] = obj[2]
                _fill_mtl(rec, obj[3])
                rec["id"] = obj[4]
                spheres.append(rec)
            else:
                rec = np.zeros((), TRIANGLE)
                rec["v0"], rec["v1"], rec["v2"] = obj[1][0], obj[1][1], obj[1][2]
                _fill_mtl(rec, obj[2])
                rec["id"] = obj[3]
                tris.append(rec)
    lights = np.zeros(len(scene.lights), LIGHT)
    for i, L in enumerate(scene.lights):
        d = L["dir"].astype(f32)
        ln = f32(np.sqrt(f32(f32(f32(d[0] * d[0]) + f32(d[1] * d[1])) + f32(d[2] * d[2]))))
        lights[i]["pos"] = L["pos"]
        lights[i]["dir"] = np.array([d[0] / ln, d[1] / ln, d[2] / ln], f32)
        lights[i]["illum"] = (L["illum"] / f32(illum_divisor)).astype(f32) if illum_divisor != 1.0 else L["illum"]
        lights[i]["cutoff"] = _radians(L["cutoff_deg"])
        lights[i]["is_parallel"] = L["is_parallel"]
        lights[i]["light_ball"]["center"] = L["pos"]
        lights[i]["light_ball"]["r"] = L["ball_r"]
        lights[i]["light_ball"]["mtl_old"]["Kd"] = L["illum"]
    sph = np.array(spheres, SPHERE) if spheres else np.zeros(0, SPHERE)
    tri = np.array(tris, TRIANGLE) if tris else np.zeros(0, TRIANGLE)
    return lights, sph, tri


def flatten_for_pt(scene: SceneDesc):
    return flatten(scene, 1.0)


def flatten_for_bdpt(scene: SceneDesc, light_sample: int):
    return flatten(scene, float(light_sample))


def _norm_glm(v):
    # glm::normalize = v * inversesqrt(dot(v, v)), float32 throughout
    v = v.astype(f32)
    d = f32(f32(f32(v[0] * v[0]) + f32(v[1] * v[1])) + f32(v[2] * v[2]))
    inv = f32(1.0) / f32(np.sqrt(d))
    return (v * inv).astype(f32)


def _cross(a, b):
    return np.array([f32(f32(a[1] * b[2]) - f32(a[2] * b[1])),
                     f32(f32(a[2] * b[0]) - f32(a[0] * b[2])),
                     f32(f32(a[0] * b[1]) - f32(a[1] * b[0]))], f32)


_LIBM = None


def _tanf(x: float) -> np.float32:
    global _LIBM
    if _LIBM is None:
        import ctypes
        import ctypes.util
        _LIBM = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
        _LIBM.tanf.restype = ctypes.c_float
        _LIBM.tanf.argtypes = [ctypes.c_float]
    return f32(_LIBM.tanf(float(x)))


def make_camera(eye, look_at, view_up, fov_deg: float, W: int, H: int, tan_in_float: bool = False) -> np.ndarray:
    """init_camera (src/main_cli.cpp:25-40): top-left origin, +dy goes down.
    The reference CLI hard-codes fov 50 (main_cli.cpp:158); cpu_bdpt uses the parsed one and takes
    the tangent in float (std::tan(float), src/cpu_bdpt.cpp:192): tan_in_float=True."""
    eye = np.asarray(eye, f32)
    look_at = np.asarray(look_at, f32)
    view_up = np.asarray(view_up, f32)
    aspect = f32(f32(W) / f32(H))
    theta = f32(f32(f32(fov_deg) * f32(3.14159265358979323846)) / f32(180.0))
    half_h = _tanf(float(f32(theta / f32(2.0)))) if tan_in_float else f32(math.tan(float(f32(theta / f32(2.0)))))
    half_w = f32(aspect * half_h)
    w = _norm_glm(eye - look_at)
    u = _norm_glm(_cross(view_up, w))
    v = _cross(w, u)
    UL = (((eye - (half_w * u).astype(f32)).astype(f32) + (half_h * v).astype(f32)).astype(f32) - w).astype(f32)
    dx = (((f32(2.0) * half_w) * u).astype(f32) / f32(W)).astype(f32)
    dy = (((f32(-2.0) * half_h) * v).astype(f32) / f32(H)).astype(f32)
    cam = np.zeros((), CAMERA)
    cam["eye"] = eye
    cam["UL"] = UL
    cam["dx"] = dx
    cam["dy"] = dy
    return cam


def camera_for(scene: SceneDesc, W: int, H: int, fov_deg: float = 50.0) -> np.ndarray:
    return make_camera(scene.eye, scene.look_at, scene.view_up, fov_deg, W, H)


# ----------------------------------------------------------------------------------------
# synthetic scenes
# ----------------------------------------------------------------------------------------
_CORNELL_WALLS = [
    # (material6, [triangles as 9 floats]) -- the 12 wall triangles of the Cornell-style box
    ((0.2, 0.2, 0.2, 0.3, 0.0, 0.0), [(-0.5, -0.5, -1.1, -0.5, -0.5, 1.0, 0.5, -0.5, 1.0),
                                      (-0.5, -0.5, -1.1, 0.5, -0.5, -1.1, 0.5, -0.5, 1.0)]),
    ((1.0, 0.0, 0.0, 1.0, 0.0, 0.0), [(-0.5, -0.5, 1.0, -0.5, -0.5, -1.1, -0.5, 0.5, 1.0),
                                      (-0.5, 0.5, 1.0, -0.5, -0.5, -1.1, -0.5, 0.5, -1.1)]),
    ((0.0, 1.0, 0.0, 1.0, 0.0, 0.0), [(0.5, -0.5, -1.1, 0.5, -0.5, 1.0, 0.5, 0.5, 1.0),
                                      (0.5, -0.5, -1.1, 0.5, 0.5, 1.0, 0.5, 0.5, -1.1)]),
    ((1.0, 1.0, 1.0, 0.0, 1.0, 0.0), [(0.5, -0.5, 1.0, -0.5, -0.5, 1.0, -0.5, 0.5, 1.0),
                                      (0.5, -0.5, 1.0, -0.5, 0.5, 1.0, 0.5, 0.5, 1.0)]),
    ((0.3, 0.3, 0.3, 1.0, 0.0, 0.0), [(0.5, -0.5, -1.1, -0.5, -0.5, -1.1, -0.5, 0.5, -1.1),
                                      (0.5, -0.5, -1.1, -0.5, 0.5, -1.1, 0.5, 0.5, -1.1)]),
    ((0.2, 0.2, 0.2, 1.0, 0.0, 0.0), [(-0.5, 0.5, -1.1, -0.5, 0.5, 1.0, 0.5, 0.5, 1.0),
                                      (-0.5, 0.5, -1.1, 0.5, 0.5, -1.1, 0.5, 0.5, 1.0)]),
]


def _box_tris(cx, cz, half, y0, y1, angle):
    """24... no: 12 triangles of an upright box (4 sides, top, bottom), rotated about y."""
    ca, sa = math.cos(angle), math.sin(angle)
    corners = []
    for sx, sz in ((-1, -1), (1, -1), (1, 1), (-1, 1)):
        x, z = sx * half, sz * half
        corners.append((cx + ca * x - sa * z, cz + sa * x + ca * z))
    tris = []
    for i in range(4):
        (xa, za), (xb, zb) = corners[i], corners[(i + 1) % 4]
        tris.append((xa, y0, za, xb, y0, zb, xb, y1, zb))
        tris.append((xa, y0, za, xb, y1, zb, xa, y1, za))
    (x0, z0), (x1, z1), (x2, z2), (x3, z3) = corners
    tris.append((x0, y1, z0, x1, y1, z1, x2, y1, z2))
    tris.append((x0, y1, z0, x2, y1, z2, x3, y1, z3))
    tris.append((x0, y0, z0, x1, y0, z1, x2, y0, z2))
    tris.append((x0, y0, z0, x2, y0, z2, x3, y0, z3))
    return tris


def _tris_from(rows, mats):
    n = len(rows)
    out = np.zeros(n, TRIANGLE)
    a = np.asarray(rows, f32).reshape(n, 3, 3)
    m = np.asarray(mats, f32).reshape(n, 6)
    out["v0"], out["v1"], out["v2"] = a[:, 0], a[:, 1], a[:, 2]
    out["mtl"]["base_color"] = m[:, 0:3]
    out["mtl"]["roughness"] = m[:, 3]
    out["mtl"]["metallic"] = m[:, 4]
    out["mtl"]["eta"] = m[:, 5]
    out["mtl"]["type"] = np.where(m[:, 5] > 0, MAT_DIELECTRIC, np.where(m[:, 4] > 0, MAT_CONDUCTOR, MAT_UBER))
    out["id"] = np.arange(n)
    return out


def _one_light(pos, direction, illum, cutoff_deg, is_parallel, ball_r):
    L = np.zeros(1, LIGHT)
    d = np.asarray(direction, f32)
    ln = f32(np.sqrt(f32(f32(f32(d[0] * d[0]) + f32(d[1] * d[1])) + f32(d[2] * d[2]))))
    L[0]["pos"] = pos
    L[0]["dir"] = d / ln
    L[0]["illum"] = illum
    L[0]["cutoff"] = _radians(f32(cutoff_deg))
    L[0]["is_parallel"] = is_parallel
    L[0]["light_ball"]["center"] = pos
    L[0]["light_ball"]["r"] = ball_r
    L[0]["light_ball"]["mtl_old"]["Kd"] = illum
    return L


CORNELL_EYE = (0.0, 0.0, -1.0)
CORNELL_LOOK = (0.0, 0.0, 1.0)
CORNELL_UP = (0.0, 1.0, 0.0)


def cornell_diffuse():
    """S2: the Cornell-style box's 12 wall triangles + two 12-triangle boxes, every
    material diffuse (r g b 1 0 0), no spheres, one 180-degree cone light under the
    ceiling.  Returns (lights, spheres, triangles)."""
    rows, mats = [], []
    for m6, tl in _CORNELL_WALLS:
        for t in tl:
            rows.append(t)
            mats.append((m6[0], m6[1], m6[2], 1.0, 0.0, 0.0))
    for (cx, cz, ang) in ((0.05, 0.1, 0.53), (0.18, 0.31, 0.70)):
        for t in _box_tris(cx, cz, 0.085, -0.5, -0.4, ang):
            rows.append(t)
            mats.append((0.8, 0.7, 0.2, 1.0, 0.0, 0.0))
    tris = _tris_from(rows, mats)
    lights = _one_light((0.0, 0.49, 0.0), (0.0, -1.0, 0.0), (1.0, 1.0, 1.0), 180.0, 0, 0.1)
    return lights, np.zeros(0, SPHERE), tris


def tessellated_sphere(center, radius, nlat, nlon):
    """Lat-long sphere, 2 triangles per (lat, lon) cell; pole cells are degenerate slivers
    of zero area on one side and are dropped, giving 2*nlon*(nlat-1) triangles."""
    th = np.linspace(0.0, math.pi, nlat + 1)
    ph = np.linspace(0.0, 2.0 * math.pi, nlon + 1)
    st, ct = np.sin(th), np.cos(th)
    sp, cp = np.sin(ph), np.cos(ph)
    P = np.empty((nlat + 1, nlon + 1, 3), np.float64)
    P[..., 0] = center[0] + radius * st[:, None] * cp[None, :]
    P[..., 1] = center[1] + radius * ct[:, None] * np.ones_like(cp)[None, :]
    P[..., 2] = center[2] + radius * st[:, None] * sp[None, :]
    a = P[:-1, :-1]
    b = P[1:, :-1]
    c = P[1:, 1:]
    d = P[:-1, 1:]
    t1 = np.stack([a, b, c], axis=2).reshape(nlat, nlon, 9)      # (nlat, nlon, 3 verts * xyz)
    t2 = np.stack([a, c, d], axis=2).reshape(nlat, nlon, 9)
    # drop the zero-area pole triangles: t2 in the first latitude band, t1 in the last
    keep1 = np.ones((nlat, nlon), bool)
    keep2 = np.ones((nlat, nlon), bool)
    keep2[0, :] = False
    keep1[-1, :] = False
    rows = np.concatenate([t1[keep1], t2[keep2]], axis=0)
    return rows.astype(f32)


def cornell_with_sphere(n_target: int = 100_000, material=(0.7, 0.7, 0.7, 1.0, 0.0, 0.0)):
    """S3/S5: diffuse-lit Cornell walls (original wall materials, incl. the mirror back
    wall) + one light + a tessellated sphere (centre (-0.15,0.2,0.45), r 0.18) with
    about n_target triangles.  Returns (lights, spheres, triangles)."""
    rows, mats = [], []
    for m6, tl in _CORNELL_WALLS:
        for t in tl:
            rows.append(t)
            mats.append(m6)
    n = max(int(math.floor(math.sqrt(n_target / 2.0))), 3)
    sph_rows = tessellated_sphere((-0.15, 0.2, 0.45), 0.18, n, n)
    walls = _tris_from(rows, mats)
    ball = _tris_from(sph_rows, np.tile(np.asarray(material, f32), (len(sph_rows), 1)))
    tris = np.concatenate([walls, ball])
    tris["id"] = np.arange(len(tris))
    lights = _one_light((0.0, 0.49, 0.0), (0.0, -1.0, 0.0), (1.0, 1.0, 1.0), 180.0, 0, 0.1)
    return lights, np.zeros(0, SPHERE), tris


def cornell_random_triangles(n: int, seed: int = 12345, edge: float = 0.01):
    """Incoherent-BVH stress variant of S3: walls + n small random triangles in the box."""
    rng = np.random.default_rng(seed)
    rows, mats = [], []
    for m6, tl in _CORNELL_WALLS:
        for t in tl:
            rows.append(t)
            mats.append(m6)
    walls = _tris_from(rows, mats)
    c = rng.uniform([-0.45, -0.45, -0.2], [0.45, 0.45, 0.95], size=(n, 1, 3))
    v = c + rng.uniform(-edge, edge, size=(n, 3, 3))
    col = rng.uniform(0.2, 0.9, size=(n, 3))
    m = np.concatenate([col, np.ones((n, 1)), np.zeros((n, 2))], axis=1)
    small = _tris_from(v.reshape(n, 9).astype(f32), m.astype(f32))
    tris = np.concatenate([walls, small])
    tris["id"] = np.arange(len(tris))
    lights = _one_light((0.0, 0.49, 0.0), (0.0, -1.0, 0.0), (1.0, 1.0, 1.0), 180.0, 0, 0.1)
    return lights, np.zeros(0, SPHERE), tris


def _triangle_block_text(v9: np.ndarray) -> str:
    """'T x0 y0 z0 ... z2' lines for an [n, 9] float32 array; shortest round-trip float32 decimals."""
    if len(v9) == 0:
        return ""
    try:
        import io
        import pyarrow as pa
        import pyarrow.csv as pacsv
        cols = [pa.array(np.full(len(v9), "T"))] + [pa.array(np.ascontiguousarray(v9[:, k], f32)) for k in range(9)]
        table = pa.Table.from_arrays(cols, names=["t"] + ["c%d" % k for k in range(9)])
        buf = io.BytesIO()
        pacsv.write_csv(table, buf, pacsv.WriteOptions(include_header=False, delimiter=" ", quoting_style="none"))
        return buf.getvalue().decode("ascii")
    except Exception:
        return "".join("T " + " ".join("%.9g" % x for x in row) + "\n" for row in v9)


def scene_to_text(lights, spheres, tris, W, H, eye=CORNELL_EYE, look=CORNELL_LOOK, up=CORNELL_UP, fov=50.0) -> str:
    """Writes boundary arrays back out in the reference's text grammar ('T'/'S'/'M'/'L' lines, SURVEY
    Appendix A): triangles in array order (one 'M' line per run of equal materials), then spheres, then lights."""
    out = ["E %.9g %.9g %.9g\n" % tuple(eye), "V %.9g %.9g %.9g %.9g %.9g %.9g\n" % (tuple(look) + tuple(up)), "F %.9g\n" % fov, "R %d %d\n" % (W, H)]

    def mat_line(m):
        return "M %.9g %.9g %.9g %.9g %.9g %.9g\n" % (float(m["base_color"][0]), float(m["base_color"][1]), float(m["base_color"][2]),
                                                      float(m["roughness"]), float(m["metallic"]), float(m["eta"]))

    last = None
    if len(tris):
        keys = np.concatenate([tris["mtl"]["base_color"].reshape(-1, 3), tris["mtl"]["roughness"].reshape(-1, 1),
                               tris["mtl"]["metallic"].reshape(-1, 1), tris["mtl"]["eta"].reshape(-1, 1)], axis=1).astype(f32)
        change = np.flatnonzero(np.any(keys[1:] != keys[:-1], axis=1)) + 1
        starts = np.concatenate([[0], change, [len(tris)]])
        v9 = np.concatenate([tris["v0"], tris["v1"], tris["v2"]], axis=1).astype(f32)
        for a, b in zip(starts[:-1], starts[1:]):
            out.append(mat_line(tris[a]["mtl"]))
            out.append(_triangle_block_text(v9[a:b]))
        last = tuple(keys[-1])
    for rec in spheres:
        m = rec["mtl"]
        key = (f32(m["base_color"][0]), f32(m["base_color"][1]), f32(m["base_color"][2]), f32(m["roughness"]), f32(m["metallic"]), f32(m["eta"]))
        if last is None or tuple(key) != tuple(last):
            out.append(mat_line(m))
            last = key
        out.append("S %.9g %.9g %.9g %.9g\n" % (rec["center"][0], rec["center"][1], rec["center"][2], rec["r"]))
    for L in lights:
        deg = float(L["cutoff"]) / 0.017453292519943295
        out.append("L %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %d %.9g\n" % (
            L["pos"][0], L["pos"][1], L["pos"][2], L["dir"][0], L["dir"][1], L["dir"][2],
            L["illum"][0], L["illum"][1], L["illum"][2], deg, int(L["is_parallel"]), L["light_ball"]["r"]))
    return "".join(out)


def write_obj(path: str, tris, quads: bool = False) -> None:
    """Writes triangles as a Wavefront OBJ ('v' / 'f' lines, three new vertices per face; faces reference them with
    negative, i.e. relative, indices on every other face to exercise both forms)."""
    with open(path, "w") as fh:
        fh.write("# %d triangles\no mesh\n" % len(tris))
        for i, t in enumerate(tris):
            for k in ("v0", "v1", "v2"):
                fh.write("v %.9g %.9g %.9g\n" % tuple(t[k]))
            if i % 2 == 0:
                fh.write("f %d/1/1 %d/2/1 %d/3/1\n" % (3 * i + 1, 3 * i + 2, 3 * i + 3))
            else:
                fh.write("f -3 -2//7 -1\n")


_HOST = None


def _host_lib():
    global _HOST
    if _HOST is None:
        import ctypes
        import os
        so = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libhpt_host.so")
        if not os.path.exists(so):
            raise RuntimeError("libhpt_host.so not built: run `make -C path_tracing_amd/csrc`")
        _HOST = ctypes.CDLL(so)
        _HOST.hpt_host_flatten_file.restype = ctypes.c_int
    return _HOST


def load_scene_fast(path: str, obj: bool = False, W: int = 0, H: int = 0) -> dict:
    """Scene text (or, obj=True, a Wavefront OBJ) through the C++ front-end (csrc/host/: mapped file +
    tokenizer): the boundary arrays flatten_for_pt would give, the (kind, index, group) ordering, the camera
    description and -- when W, H are given -- the CudaCamera the CLI would use."""
    import ctypes as C
    lib = _host_lib()
    nl, ns, nt = C.c_int(), C.c_int(), C.c_int()
    pl, ps, pt = C.c_void_p(), C.c_void_p(), C.c_void_p()
    pk, pi, pg = C.c_void_p(), C.c_void_p(), C.c_void_p()
    cam = np.zeros((), CAMERA)
    res = (C.c_int * 2)()
    ms = C.c_double()
    desc = (C.c_float * 10)()
    rc = lib.hpt_host_flatten_file(path.encode(), 1 if obj else 0, C.byref(nl), C.byref(ns), C.byref(nt), C.byref(pl), C.byref(ps), C.byref(pt),
                                   C.byref(pk), C.byref(pi), C.byref(pg), cam.ctypes.data_as(C.c_void_p), int(W), int(H), res, C.byref(ms), desc)
    if rc != 0:
        raise IOError("cannot read %s" % path)

    def grab(p, n, dt):
        return np.frombuffer(C.string_at(p, n * np.dtype(dt).itemsize), dt).copy() if n else np.zeros(0, dt)

    nobj = ns.value + nt.value
    d = np.array(list(desc), f32)
    return dict(lights=grab(pl, nl.value, LIGHT), spheres=grab(ps, ns.value, SPHERE), tris=grab(pt, nt.value, TRIANGLE),
                order=(grab(pk, nobj, np.int32), grab(pi, nobj, np.int32), grab(pg, nobj, np.int32)),
                eye=d[0:3], look_at=d[3:6], view_up=d[6:9], fov=float(d[9]), resolution=(res[0], res[1]),
                camera=cam if W > 0 and H > 0 else None, parse_ms=ms.value)


def object_order(scene_desc=None, spheres=None, tris=None):
    """(kind, index, group) arrays: the scene file's insertion order per group when a parsed
    SceneDesc is given, otherwise spheres then triangles in one group."""
    kind, index, group = [], [], []
    if scene_desc is not None:
        ns = nt = 0
        for gid in sorted(scene_desc.groups):
            for obj in scene_desc.groups[gid]:
                if obj[0] == "S":
                    kind.append(0); index.append(ns); ns += 1
                else:
                    kind.append(1); index.append(nt); nt += 1
                group.append(gid)
    else:
        for i in range(len(spheres)):
            kind.append(0); index.append(i); group.append(0)
        for i in range(len(tris)):
            kind.append(1); index.append(i); group.append(0)
    return (np.asarray(kind, np.int32), np.asarray(index, np.int32), np.asarray(group, np.int32))
