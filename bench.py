#!/usr/bin/env python3
"""Benchmark of the MI355X path-tracing hot path (BASELINE.json metric).

One "step" = one full render of the workload: BASELINE.json configs[2] -- unidirectional PT
with NEE, Cornell-style box + ~100k-triangle tessellated sphere behind a BVH, 1024 x 1024,
256 spp, depth 4 -- with the scene and BVH already resident in HBM.  With N > 1 ranks (one
process per GPU, launched by torch.distributed.run) the image's tiles are dealt round-robin
to the ranks and rank 0 gathers the framebuffer over RCCL; total work is fixed ("strong").

Prints ONE JSON line on rank 0 (contract in the task statement); see DESIGN.md "Measurement"
for how roofline.achieved, roofline.traffic and cpu_baseline are defined.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md, chip-level parameters)
HBM_ACHIEVABLE_GBS = 6290.0  # measured float4 copy, same table


def cpu_baseline(lights, spheres, tris, W, H, depth, budget_s=20.0):
    """CPU baseline beside the GPU number (north_star: "next to cpu_bdpt.cpp timed on the same
    box's host cores").  /root/reference does not exist on the GPU box, so what is timed is the
    oracle's restatement of cpu_bdpt.cpp (oracle/bdpt_oracle.cpp, kind "port"; it replays the real
    cpu_bdpt.cpp bit for bit on input.txt, tests/test_bdpt_oracle.py) on a bounded window of the
    same scene, depth 4/4, spl 8, OpenMP over all host cores.  The PT-estimator port
    (oracle/pt_oracle.cpp, brute-force scans) is timed as well and reported under "pt_port"."""
    import oracle
    from path_tracing_amd import scene_io
    threads = os.cpu_count() or 1
    order = oracle.object_order(None, spheres, tris)
    win = (480, 320, 608, 384)                      # 128 x 64 window over the sphere's silhouette
    eye, look, up = scene_io.CORNELL_EYE, scene_io.CORNELL_LOOK, scene_io.CORNELL_UP
    t0 = time.perf_counter()
    _, st = oracle.bdpt_render(lights, spheres, tris, order, eye, look, up, 50.0, W, H, depth, depth, 1, 8, seed=1, window=win)
    t1 = time.perf_counter() - t0
    spp = int(max(1, min(64, budget_s / max(t1, 1e-3))))
    t0 = time.perf_counter()
    _, st = oracle.bdpt_render(lights, spheres, tris, order, eye, look, up, 50.0, W, H, depth, depth, spp, 8, seed=1, window=win)
    dt = time.perf_counter() - t0
    out = {"value": st["samples"] / dt / 1e6, "unit": "Msamples/s", "cores": threads, "kind": "port",
           "sample": "oracle/bdpt_oracle.cpp (restated cpu_bdpt.cpp estimator: eye paths connected to %d light vertices, "
                     "brute-force group scan of all %d triangles), %dx%d window %s of the 1024x1024 image, %d spp, spl 8, "
                     "%d samples / %d shadow rays in %.1f s, OpenMP %d threads"
                     % (8 * depth, len(tris), win[2] - win[0], win[3] - win[1], str(win), spp, st["samples"], st["shadow_rays"], dt, threads)}
    cam = scene_io.make_camera(eye, look, up, 50.0, W, H)
    t0 = time.perf_counter()
    _, sp1 = oracle.pt_render(lights, spheres, tris, cam, W, H, depth, 1, seed=1, window=win)
    t1 = time.perf_counter() - t0
    spp = int(max(1, min(64, 0.5 * budget_s / max(t1, 1e-3))))
    t0 = time.perf_counter()
    _, sp1 = oracle.pt_render(lights, spheres, tris, cam, W, H, depth, spp, seed=1, window=win)
    dt = time.perf_counter() - t0
    out["pt_port"] = {"value": sp1["samples"] / dt / 1e6, "unit": "Msamples/s", "cores": threads,
                      "sample": "oracle/pt_oracle.cpp (reference PT loop, brute-force scans), same window, %d spp, %d samples in %.1f s" % (spp, sp1["samples"], dt)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--tris", type=int, default=100_000)
    ap.add_argument("--depth", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--single-pipeline", action="store_true", help="one pass in flight at a time (HPT_FLAG_SINGLE_PIPELINE) in every step")
    ap.add_argument("--no-exclusive-step", action="store_true", help="skip the extra untimed single-pipeline step that measures the kernels' exclusive durations")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse the multi-rank path on a one-GPU box)")
    ap.add_argument("--pmc-file", default=os.path.join(ROOT, "profiles", "r01_pmc_traffic.json"))
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import path_tracing_amd as hpt
    from path_tracing_amd import distributed, scene_io

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback)")
    device_index = local_rank % ndev              # one GPU per rank; ranks share a GPU only in the gloo rehearsal
    torch.cuda.set_device(device_index)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    W = H = args.size
    lights, spheres, tris = scene_io.cornell_with_sphere(args.tris)
    cam = scene_io.make_camera(scene_io.CORNELL_EYE, scene_io.CORNELL_LOOK, scene_io.CORNELL_UP, 50.0, W, H)
    scene = hpt.Scene(lights, spheres, tris)               # upload + BVH build: outside the timed region
    stream = torch.cuda.current_stream().cuda_stream
    base = dict(seed=1, rank=rank, world=world)
    n_local = hpt.local_pixels(W, H, hpt.make_params(**base))
    local = torch.zeros((n_local, 3), dtype=torch.float32, device="cuda")
    image = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda") if rank == 0 else None

    def step(flags):
        p = hpt.make_params(flags=flags, **base)

        def render_local():
            scene.render_pt_device(cam, W, H, args.depth, args.spp, p, local.data_ptr(), stream)
            return local

        def untile(gathered):
            hpt.untile(gathered.data_ptr(), image.data_ptr(), W, H, hpt.make_params(**base), stream)
            return image

        return distributed.render_tiled(render_local, untile, rank, world)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    base_flags = hpt.FLAG_SINGLE_PIPELINE if args.single_pipeline else 0
    for _ in range(args.warmup):
        step(base_flags)
    fence()
    t0 = time.perf_counter()
    ext_ms, ext_n, tot_ms, res_ms, res_n = 0.0, 0, 0.0, 0.0, 0
    for _ in range(args.steps):
        step(base_flags | hpt.FLAG_TIME_KERNELS)        # HIP events around every launch, on the launch stream
        st = scene.stats()                 # waits for this rank's render
        ext_ms += st["ms_extend"] + st["ms_connect"]; ext_n += st["n_extend"] + st["n_connect"]; tot_ms += st["ms_total"]
        res_ms += st["ms_resume"]; res_n += st["n_resume"]
        shade_ms, connect_ms, other_ms = st["ms_shade"], st["ms_connect"], st["ms_other"]
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # untimed: the same render with one pass in flight at a time -- the kernels' durations when they have the
    # device to themselves (in the timed steps two pipelines share it, so a launch takes longer there)
    excl = None
    if not args.single_pipeline and not args.no_exclusive_step:
        step(hpt.FLAG_SINGLE_PIPELINE | hpt.FLAG_TIME_KERNELS)
        fence()
        excl = scene.stats()
    # untimed pass that counts the work of the same render (boxes / triangles / rays per kernel)
    step(hpt.FLAG_COUNT_WORK)
    fence()
    wc = scene.stats()

    if rank == 0:
        samples = W * H * args.spp
        value = samples * args.steps / dt / 1e6
        # dominant kernel: k_trace (closest-hit + any-hit BVH traversal).  One trace step per iteration =
        # the first launch (every ray, `split_budget` node steps) + the resume launch (the rays that need
        # more); "launch" below is that pair, its duration the sum of the two HIP-event brackets.
        # Algorithmic bytes per step: 32 B per child box slab-tested + 36 B per triangle tested + 44 B per
        # closest-hit ray (queue index, origin, direction in; hit record out) + 36 B per shadow ray (queue
        # index, origin|max, direction in), counted on the plain single-launch traversal of the same rays
        # (the restarts of the split are not algorithmic work) -- DESIGN.md "Kernels".
        ext_bytes = (32.0 * (wc["boxes_closest"] + wc["boxes_shadow"]) + 36.0 * (wc["tris_closest"] + wc["tris_shadow"])
                     + 44.0 * wc["closest_rays"] + 36.0 * wc["shadow_rays"])
        avg_first_ms = ext_ms / max(ext_n, 1)
        avg_resume_ms = res_ms / max(res_n, 1)
        avg_launch_ms = (ext_ms + res_ms) / max(ext_n, 1)
        launches_per_render = ext_n / max(args.steps, 1)
        bytes_per_launch = ext_bytes / max(launches_per_render, 1)
        achieved = bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else 0.0
        traffic = None; valu = None
        if os.path.exists(args.pmc_file):
            try:
                pmc = json.load(open(args.pmc_file))
                traffic = pmc.get("hbm_bytes_per_launch")
                valu = pmc.get("trace_valu_utilization")
            except Exception:
                traffic = None
        all_rays = wc["closest_rays"] + wc["shadow_rays"]
        per_sample = (32.0 * (wc["boxes_closest"] + wc["boxes_shadow"]) + 36.0 * (wc["tris_closest"] + wc["tris_shadow"])
                      + (28.0 + 128.0) * wc["path_iters"]) / max(wc["samples"], 1) + 12.0 / args.spp
        out = {
            "metric": "Msamples/s (paths/s) at 1024^2 x 256spp",
            "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[2]: unidirectional PT + NEE, Cornell-style box + tessellated sphere, "
                                   "%d triangles behind a BVH, %dx%d, %d spp, depth %d, 1 cone light"
                                   % (len(tris), W, H, args.spp, args.depth),
                       "parallelism": "image tiles 32x32 round-robin over %d rank(s), %s gather to rank 0" % (world, "RCCL" if args.backend == "nccl" else args.backend),
                       "seed": 1},
            "roofline": {"bound": "hbm", "kernel": "k_trace (closest-hit + any-hit BVH traversal; one step = first launch k_trace<false,false> + resume launch k_trace<false,true>)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "frac_of_measured_copy_peak": achieved / HBM_ACHIEVABLE_GBS, "traffic": traffic,
                         "avg_launch_ms": avg_launch_ms, "avg_first_launch_ms": avg_first_ms, "avg_resume_launch_ms": avg_resume_ms,
                         "launches_per_step": launches_per_render, "split_budget": st["split_budget"],
                         "long_ray_fraction_last_pass": st["long_rays_last_pass"] / max(st["traced_rays_last_pass"], 1),
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "pipelines_in_flight": 1 if args.single_pipeline else 2,
                         "exclusive": None if excl is None else {
                             "note": "same render, untimed, HPT_FLAG_SINGLE_PIPELINE: one pass in flight, each launch has the device to itself",
                             "ms_per_step": excl["ms_total"],
                             "avg_first_launch_ms": (excl["ms_extend"] + excl["ms_connect"]) / max(excl["n_extend"] + excl["n_connect"], 1),
                             "avg_resume_launch_ms": excl["ms_resume"] / max(excl["n_resume"], 1),
                             "achieved": bytes_per_launch / ((excl["ms_extend"] + excl["ms_connect"] + excl["ms_resume"]) / max(excl["n_extend"] + excl["n_connect"], 1) * 1e-3) / 1e9},
                         "valu_utilization_pmc": valu,
                         "note": "rank 0's kernels; bytes = 32*boxes + 36*tris + 44*closest rays + 36*shadow rays of this rank. "
                                 "Two passes of a render are in flight on two streams, so a launch shares the device with the other pipeline's kernels "
                                 "and its duration (hence `achieved`) is that of a shared device; `exclusive` has the single-pipeline figures. "
                                 "The BVH and triangles of this scene stay in L2 / Infinity Cache, so the algorithmic bytes are not "
                                 "HBM traffic (`traffic` is what the fabric saw, PMC) and frac can exceed 1; the kernel's binding limit "
                                 "is VALU issue (valu_utilization_pmc = SQ_INSTS_VALU x 4 cycles / SIMD cycles, from profiles/)"},
            "work": {"rays_per_sample": all_rays / max(wc["samples"], 1),
                     "boxes_per_ray": (wc["boxes_closest"] + wc["boxes_shadow"]) / max(all_rays, 1),
                     "tris_per_ray": (wc["tris_closest"] + wc["tris_shadow"]) / max(all_rays, 1),
                     "algorithmic_bytes_per_sample": per_sample,
                     "Mrays_per_s": all_rays * world / (dt / args.steps) / 1e6 if world == 1 else None,
                     "device_ms_per_step_rank0": tot_ms / args.steps,
                     "kernel_ms_last_step_rank0": {"trace": st["ms_extend"] + st["ms_connect"] + st["ms_resume"], "shade": shade_ms, "other": other_ms},
                     "bvh_nodes": wc["bvh_nodes"], "bvh_depth": wc["bvh_depth"], "ms_bvh_build": wc["ms_bvh_build"]},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(lights, spheres, tris, W, H, args.depth)
        print(json.dumps(out), flush=True)
    scene.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
