#!/usr/bin/env python3
"""Writes the benchmark scenes' triangle records and runs scripts/micro/bvh4_steps.cpp on them (host only)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from path_tracing_amd import scene_io as S
exe = "/tmp/bvh4_steps"
subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "path_tracing_amd/csrc"), "-o", exe,
                       os.path.join(ROOT, "scripts/micro/bvh4_steps.cpp"), os.path.join(ROOT, "path_tracing_amd/csrc/scene_build.cpp")])
for name, (L, sp, tr) in (("cornell + 100k sphere", S.cornell_with_sphere(100_000)), ("cornell + 1M sphere", S.cornell_with_sphere(1_000_000)),
                          ("100k random triangles", S.cornell_random_triangles(100_000))):
    path = "/tmp/bvh4_tris.bin"
    np.ascontiguousarray(tr).tofile(path)
    print("==", name, flush=True)
    subprocess.check_call([exe, path, "300000"])
