"""POD layouts that cross the drop-in boundary, as numpy dtypes.

These are the reference's host-side structs (reference: include/geometric.cuh:15-51,
67-78), with the sizes/offsets measured in SURVEY.md Appendix D.  The C ABI in
include/hpt.h takes arrays of exactly these records.
"""
import numpy as np

MATERIAL_OLD = np.dtype([("Kd", "<f4", 3), ("Kg", "<f4", 3), ("Ks", "<f4", 3),
                         ("glossy", "<f4"), ("exp", "<f4"), ("refract", "<f4"), ("reflect", "<f4")])
MATERIAL = np.dtype([("base_color", "<f4", 3), ("roughness", "<f4"), ("metallic", "<f4"),
                     ("eta", "<f4"), ("type", "<i4")])
SPHERE = np.dtype([("center", "<f4", 3), ("r", "<f4"), ("mtl_old", MATERIAL_OLD), ("mtl", MATERIAL), ("id", "<i4")])
TRIANGLE = np.dtype([("v0", "<f4", 3), ("v1", "<f4", 3), ("v2", "<f4", 3),
                     ("mtl_old", MATERIAL_OLD), ("mtl", MATERIAL), ("id", "<i4")])
LIGHT = np.dtype([("pos", "<f4", 3), ("dir", "<f4", 3), ("illum", "<f4", 3), ("light_ball", SPHERE),
                  ("cutoff", "<f4"), ("is_parallel", "<i4")])
CAMERA = np.dtype([("eye", "<f4", 3), ("U", "<f4", 3), ("V", "<f4", 3), ("W", "<f4", 3),
                   ("UL", "<f4", 3), ("dx", "<f4", 3), ("dy", "<f4", 3)])

assert MATERIAL_OLD.itemsize == 52 and MATERIAL.itemsize == 28
assert SPHERE.itemsize == 100 and TRIANGLE.itemsize == 120
assert LIGHT.itemsize == 144 and CAMERA.itemsize == 84

# material type tags (reference: include/geometric.cuh:20, src/geometric.cu:41-49)
MAT_DIFFUSE, MAT_DIELECTRIC, MAT_CONDUCTOR, MAT_UBER = 0, 1, 2, 3
