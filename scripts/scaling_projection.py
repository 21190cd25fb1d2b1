#!/usr/bin/env python3
"""Projected multi-GPU scaling of the bench workload from ONE device.

For G in {1, 2, 4, 8} every virtual rank r of G renders its share of the image tiles on this device,
alone, exactly as rank r of a G-GPU job would (same tiles, same passes, same kernels); the slowest rank
plus a modelled gather (12*W*H/G bytes over one 153 GB/s xGMI link) gives the projected step time.
Measured here: per-rank render times and the load balance.  Projected: the G-GPU rate (no other rank's
kernels run beside this one's; the real sweep is the driver's SCALE_rNN.json).

usage: python scripts/scaling_projection.py [--size 1024 --spp 256 --tris 100000 --reps 3] [--flags N]
"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import path_tracing_amd as hpt
from path_tracing_amd import scene_io


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--tris", type=int, default=100_000)
    ap.add_argument("--depth", type=int, default=4)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--flags", type=int, default=0)
    ap.add_argument("--worlds", default="1,2,4,8")
    ap.add_argument("--spass", type=int, default=0, help="samples per pass (0 = the library's choice)")
    args = ap.parse_args()
    W = H = args.size
    lights, spheres, tris = scene_io.cornell_with_sphere(args.tris)
    cam = scene_io.make_camera(scene_io.CORNELL_EYE, scene_io.CORNELL_LOOK, scene_io.CORNELL_UP, 50.0, W, H)
    scene = hpt.Scene(lights, spheres, tris)
    stream = torch.cuda.current_stream().cuda_stream
    rows = []
    t1 = None
    for G in [int(x) for x in args.worlds.split(",")]:
        per_rank = []
        for r in range(G):
            p = hpt.make_params(seed=1, rank=r, world=G, flags=args.flags, samples_per_pass=args.spass)
            n_local = hpt.local_pixels(W, H, p)
            local = torch.zeros((n_local, 3), dtype=torch.float32, device="cuda")
            scene.render_pt_device(cam, W, H, args.depth, args.spp, p, local.data_ptr(), stream)   # warm-up (workspace)
            torch.cuda.synchronize()
            best = 1e30
            for _ in range(args.reps):
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                scene.render_pt_device(cam, W, H, args.depth, args.spp, p, local.data_ptr(), stream)
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1))
            per_rank.append(best)
        worst = max(per_rank); mean = sum(per_rank) / G
        gather_ms = 0.0 if G == 1 else 12.0 * W * H / G / 153e9 * 1e3
        step = worst + gather_ms
        if t1 is None:
            t1 = step
        rows.append({"G": G, "rank_ms_max": worst, "rank_ms_mean": mean, "load_balance_max_over_mean": worst / mean,
                     "modelled_gather_ms": gather_ms, "projected_step_ms": step,
                     "projected_msamples_per_s": W * H * args.spp / step / 1e3,
                     "projected_efficiency": t1 / (G * step)})
        print(json.dumps(rows[-1]), flush=True)
    print(json.dumps({"workload": "bench default" if (args.size, args.spp, args.tris) == (1024, 256, 100_000) else vars(args),
                      "measured": "per-rank render times on one device (best of %d)" % args.reps,
                      "projected": "step time, rate, efficiency", "rows": rows}))


if __name__ == "__main__":
    main()
