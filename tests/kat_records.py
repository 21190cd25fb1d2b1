"""Inputs of the function-level known-answer tests: seeded records for the BSDF / Fresnel / GGX
functions over the material classes of input.txt -- diffuse r = 1; rough dielectric-opaque r = 0.3 / 0.6; conductors
m = 0.9 r = 0.2 and m = 0.8 r = 0.3; mirror m = 1 r = 0; glass eta 1.5 / 2.4 -- plus random materials.
Record layout (24 floats): k_probe_functions in path_tracing_amd/csrc/pt_kernels.hip."""
import numpy as np

CLASSES = [
    ("diffuse", (0.7, 0.6, 0.5, 1.0, 0.0, 0.0)),
    ("rough_dielectric_opaque_0.3", (0.2, 0.2, 0.2, 0.3, 0.0, 0.0)),
    ("rough_dielectric_opaque_0.6", (0.8, 0.3, 0.1, 0.6, 0.0, 0.0)),
    ("conductor_0.9_0.2", (0.9, 0.7, 0.3, 0.2, 0.9, 0.0)),
    ("conductor_0.8_0.3", (0.95, 0.93, 0.88, 0.3, 0.8, 0.0)),
    ("mirror", (1.0, 1.0, 1.0, 0.0, 1.0, 0.0)),
    ("glass_1.5", (1.0, 1.0, 1.0, 0.0, 0.0, 1.5)),
    ("glass_2.4", (0.9, 0.95, 1.0, 0.0, 0.0, 2.4)),
    ("rough_glass_1.5", (1.0, 1.0, 1.0, 0.4, 0.0, 1.5)),
    ("random", None),
]


def _unit(rng, n):
    v = rng.normal(size=(n, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    return v.astype(np.float32)


def records(n_per_class: int, seed: int):
    """-> (names per record, [n, 24] float32)."""
    rng = np.random.default_rng(seed)
    rows, names = [], []
    for name, mat in CLASSES:
        n = n_per_class
        r = np.zeros((n, 24), np.float32)
        if mat is None:
            r[:, 0:3] = rng.random((n, 3))
            r[:, 3] = rng.choice([0.0, 0.0005, 0.05, 0.3, 1.0], n) * rng.random(n) ** 0.5
            r[:, 4] = rng.choice([0.0, 0.0, 0.5, 0.995, 1.0], n)
            r[:, 5] = rng.choice([0.0, 0.0, 1.33, 1.5, 2.4], n)
        else:
            r[:, 0:6] = mat
        N = _unit(rng, n)
        wo = _unit(rng, n)
        wi = _unit(rng, n)
        # the integrator flips the normal to face the ray, so wo mostly sits in N's hemisphere; keep 15 % below it
        flip = (np.einsum("ij,ij->i", wo, N) < 0) & (rng.random(n) < 0.85)
        wo[flip] *= -1
        # a few axis-aligned and grazing configurations
        k = max(1, n // 50)
        N[:k] = (0, 0, 1); N[k:2 * k] = (0, 0, -1); N[2 * k:3 * k] = (0, 1, 0)
        wo[3 * k:4 * k] = N[3 * k:4 * k]                                  # normal incidence
        wi[4 * k:5 * k] = wo[4 * k:5 * k] * -1                             # wo + wi = 0
        r[:, 6:9], r[:, 9:12], r[:, 12:15] = N, wo, wi
        r[:, 15:18] = (rng.integers(0, 1 << 24, (n, 3)) / np.float32(1 << 24)).astype(np.float32)     # 24-bit uniforms in [0, 1)
        r[:k, 15:18] = 0.0
        r[:, 18] = rng.choice([1.0, 1.5, 2.4], n)
        r[:, 19] = rng.uniform(-1.2, 1.2, n)
        r[:, 20] = rng.choice([1.0, 1.5, 2.4], n)
        r[:, 21] = rng.choice([0.0, 1.0, 1.5, 2.4], n)
        rows.append(r)
        names += [name] * n
    return np.array(names), np.concatenate(rows).astype(np.float32)


def same_bits(a, b):
    """bit-identical float32 arrays, any NaN matching any NaN"""
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
    return (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
