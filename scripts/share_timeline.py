#!/usr/bin/env python3
"""Launch-by-launch timeline of ONE rank's share of the bench workload (rank 0 of SHARE_G ranks), from a kernel trace:

  cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && SHARE_G=8 SHARE_FLAGS=32 \
      rocprofv3 --kernel-trace --output-format csv -d gpurun_out/share -- python3 scripts/share_timeline.py
  python3 scripts/share_timeline.py --read gpurun_out/share/*/*_kernel_trace.csv

SHARE_FLAGS: hpt_params.flags (32 = one pipeline, so the launches of a pass follow one another); SHARE_SIZE / SHARE_SPP /
SHARE_TRIS (0 = the diffuse Cornell box alone) change the workload."""
import csv, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def render():
    import torch
    import path_tracing_amd as hpt
    from path_tracing_amd import scene_io
    W = H = int(os.environ.get("SHARE_SIZE", "1024")); spp = int(os.environ.get("SHARE_SPP", "256"))
    G = int(os.environ.get("SHARE_G", "8")); flags = int(os.environ.get("SHARE_FLAGS", "32"), 0)
    tris = int(os.environ.get("SHARE_TRIS", "100000"))                     # 0: the diffuse Cornell box alone (config 2's scene)
    L, sp, tr = scene_io.cornell_with_sphere(tris) if tris else scene_io.cornell_diffuse()
    cam = scene_io.make_camera(scene_io.CORNELL_EYE, scene_io.CORNELL_LOOK, scene_io.CORNELL_UP, 50.0, W, H)
    scene = hpt.Scene(L, sp, tr)
    stream = torch.cuda.current_stream().cuda_stream
    p = hpt.make_params(seed=1, rank=0, world=G, flags=flags)
    local = torch.zeros((hpt.local_pixels(W, H, p), 3), dtype=torch.float32, device="cuda")
    for _ in range(3):
        scene.render_pt_device(cam, W, H, 4, spp, p, local.data_ptr(), stream); torch.cuda.synchronize()


def read(path):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    fin = [i for i, r in enumerate(rows) if "k_finalize" in r["Kernel_Name"]]
    a, b = fin[-2] + 1, fin[-1]                      # the last render of the run
    t0, prev = int(rows[a]["Start_Timestamp"]), None
    for r in rows[a:b + 1]:
        n = r["Kernel_Name"]
        m = re.search(r"k_[a-z_]+", n)
        name = m.group(0) if m else n[:24]
        if name == "k_trace": name += " resume" if re.search(r"k_trace<(true|false), true", n) else " first"
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print("%-16s start %9.1f us  duration %8.1f  gap %6.1f  workgroups %d" % (
            name, (s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])))
        prev = e


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--read": read(sys.argv[2])
    else: render()
