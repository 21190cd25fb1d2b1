#!/usr/bin/env python3
"""Benchmark of the MI355X path-tracing hot path (BASELINE.json metric).

One "step" = one full render of the workload: BASELINE.json configs[2] -- unidirectional PT
with NEE, Cornell-style box + ~100k-triangle tessellated sphere behind a BVH, 1024 x 1024,
256 spp, depth 4 -- with the scene and BVH already resident in HBM.  With N > 1 ranks (one
process per GPU) the image's tiles are dealt round-robin to the ranks and rank 0 gathers the
framebuffer over RCCL; total work is fixed ("strong").

`python bench.py --gpus N` works both ways: under `python -m torch.distributed.run ...` (RANK /
WORLD_SIZE in the environment) it is one rank; started plainly with N > 1 it starts that launcher
itself as a CHILD process, before anything touches the GPU, and relays its exit code.

Prints ONE JSON line on rank 0 (contract in the task statement); see DESIGN.md "Measurement"
for how roofline.achieved, roofline.traffic, verify and cpu_baseline are defined.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md, chip-level parameters)
HBM_ACHIEVABLE_GBS = 6290.0  # measured float4 copy, same table
# wave64 VALU instructions issue over 4 cycles on a SIMD (v_fma_f32 "one wave alone: 4", same guide); 256 CUs x 4
# SIMDs x 2.4 GHz / 4 = 614.4 G wave-instructions/s = 39.3 T lane-operations/s (the 157.3 TFLOP/s spec counts a
# packed FMA as 4 flops per lane)
VALU_PEAK_LANE_OPS = 256 * 4 * 2.4e9 / 4 * 64
LANE_OPS_PER_BOX = 20.0      # 6 fma + 6 min/max + 2 min3/max3 + scale, compare, select of one slab test (pt_kernels.hip, node step)
LANE_OPS_PER_TRI = 45.0      # Moeller-Trumbore with precomputed edges: 2 cross, 4 dot, reciprocal, 5 compares (pt_device_math.h)
WINDOW = (480, 320, 608, 384)   # 128 x 64 pixels over the sphere's silhouette: the cpu_baseline sample and the oracle check


def kernel_source_sha():
    """sha256 over the kernel sources: a PMC file in profiles/ is only quoted for the build it was taken from."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "path_tracing_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".cpp")) and os.path.isfile(os.path.join(d, f)):
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def cpu_baseline(lights, spheres, tris, W, H, depth, budget_s=20.0):
    """CPU baseline beside the GPU number (north_star: "next to cpu_bdpt.cpp timed on the same
    box's host cores").  /root/reference does not exist on the GPU box, so what is timed is the
    oracle's restatement of cpu_bdpt.cpp (oracle/bdpt_oracle.cpp, kind "port"; it replays the real
    cpu_bdpt.cpp bit for bit on input.txt, tests/test_bdpt_oracle.py) on a bounded window of the
    same scene, depth 4/4, spl 8, OpenMP over all host cores.  The PT-estimator port
    (oracle/pt_oracle.cpp, brute-force scans) is timed as well and reported under "pt_port"; its
    window image is returned for the bench's oracle check."""
    import oracle
    from path_tracing_amd import scene_io
    threads = os.cpu_count() or 1
    order = scene_io.object_order(None, spheres, tris)
    win = WINDOW
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
                     "brute-force group scan of all %d triangles), %dx%d window %s of the %dx%d image, %d spp, spl 8, "
                     "%d samples / %d shadow rays in %.1f s, OpenMP %d threads"
                     % (8 * depth, len(tris), win[2] - win[0], win[3] - win[1], str(win), W, H, spp, st["samples"], st["shadow_rays"], dt, threads)}
    cam = scene_io.make_camera(eye, look, up, 50.0, W, H)
    t0 = time.perf_counter()
    _, sp1 = oracle.pt_render(lights, spheres, tris, cam, W, H, depth, 1, seed=1, window=win)
    t1 = time.perf_counter() - t0
    spp = int(max(1, min(64, 0.5 * budget_s / max(t1, 1e-3))))
    t0 = time.perf_counter()
    ref, sp1 = oracle.pt_render(lights, spheres, tris, cam, W, H, depth, spp, seed=1, window=win)
    dt = time.perf_counter() - t0
    out["pt_port"] = {"value": sp1["samples"] / dt / 1e6, "unit": "Msamples/s", "cores": threads,
                      "sample": "oracle/pt_oracle.cpp (reference PT loop, brute-force scans), same window, %d spp, %d samples in %.1f s" % (spp, sp1["samples"], dt)}
    return out, ref[win[1]:win[3], win[0]:win[2]].copy(), spp


def spawn_ranks(n, argv):
    """Plain `python bench.py --gpus N`: start torch.distributed.run as a child process (never exec: this
    process must stay what the caller started) and hand back its exit code."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.run(cmd, env=env).returncode


def fanout_main(args):
    """`--fanout`: the same workload through hpt_multi_render_pt -- ONE process, one host thread per device, image tiles per
    device, ncclGather on the first device, host image out (what SURVEY 8(b) calls "multi-GPU fan-out is internal").  The
    timed region is the blocking call itself, so unlike the default mode it includes the copy of the image to host memory
    (12 MB at 1024^2); `value` stays whole-job samples per second."""
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    import path_tracing_amd as hpt
    from path_tracing_amd import scene_io
    ndev = hpt.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback)")
    n = args.gpus
    if n > ndev and args.fanout_exchange != 1:
        raise SystemExit("bench.py --fanout --gpus %d: %d device(s) visible (--fanout-exchange 1 lets ranks share a device)" % (n, ndev))
    ids = [i % ndev for i in range(n)]
    W = H = args.size
    lights, spheres, tris = scene_io.cornell_with_sphere(args.tris)
    cam = scene_io.make_camera(scene_io.CORNELL_EYE, scene_io.CORNELL_LOOK, scene_io.CORNELL_UP, 50.0, W, H)
    per_rank, gathers, totals = [], [], []
    with hpt.MultiScene(lights, spheres, tris, device_ids=ids, exchange=args.fanout_exchange) as ms:
        p = hpt.make_params(seed=1, flags=hpt.FLAG_SINGLE_PIPELINE if args.single_pipeline else 0)
        first = None
        for i in range(args.warmup):
            img = ms.render_pt(cam, W, H, args.depth, args.spp, p)
            if first is None:
                first = img
        t0 = time.perf_counter()
        for _ in range(args.steps):
            img = ms.render_pt(cam, W, H, args.depth, args.spp, p)
            t = ms.timing()
            per_rank.append(t["render_ms_per_device"]); gathers.append(t["gather_ms"]); totals.append(t["total_ms"])
        dt = time.perf_counter() - t0
    pr = np.array(per_rank)
    out = {"metric": "Msamples/s (paths/s) at 1024^2 x 256spp", "value": W * H * args.spp * args.steps / dt / 1e6, "unit": "Msamples/s",
           "n_gpus": n, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
           "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "configs[2]: unidirectional PT + NEE, Cornell-style box + tessellated sphere, %d triangles behind a BVH, %dx%d, "
                                  "%d spp, depth %d, 1 cone light" % (len(tris), W, H, args.spp, args.depth),
                      "parallelism": "ONE process, hpt_multi_render_pt: image tiles 32x32 round-robin over %d device rank(s) %s, %s to the first device, "
                                     "host image out" % (n, ids, "ncclGather (RCCL)" if args.fanout_exchange == 0 else "peer copies"), "seed": 1},
           "fanout": {"render_ms_per_rank_mean_over_steps": pr.mean(axis=0).tolist(), "render_ms_max_rank": float(pr.max(axis=1).mean()),
                      "render_ms_mean_rank": float(pr.mean()), "load_balance_max_over_mean": float((pr.max(axis=1) / pr.mean(axis=1)).mean()),
                      "gather_ms_per_step": gathers, "call_ms_per_step": totals,
                      "note": "device time of every rank's render (HIP events), of the exchange step, and host wall time of each blocking call; "
                              "the timed region includes the image copy to host memory"},
           "verify": {"image_finite_and_lit": bool(np.isfinite(img).all() and img.mean() > 0),
                      "timed_image_equals_first_warmup_image": bool(first is None or np.array_equal(img, first))}}
    real_stdout.write(json.dumps(out) + "\n")
    real_stdout.flush()
    if not all(out["verify"].values()):
        raise SystemExit("bench.py --fanout: verification failed: %s" % out["verify"])


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
    ap.add_argument("--no-count-step", action="store_true", help="skip the untimed work-counting render (profiling runs: the timed kernels only)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse the multi-rank path on a one-GPU box)")
    ap.add_argument("--pmc-file", default=os.path.join(ROOT, "profiles", "r03_pmc_traffic.json"))
    ap.add_argument("--fanout", action="store_true",
                    help="time the boundary's own one-process fan-out (hpt_multi_render_pt: one host thread per device, RCCL gather behind the "
                         "blocking call) over --gpus devices instead of one process per GPU")
    ap.add_argument("--fanout-exchange", type=int, default=0, help="--fanout: 0 = RCCL gather (distinct devices), 1 = peer copies "
                                                                   "(also lets several ranks share one device: a rehearsal on a one-GPU box)")
    args = ap.parse_args()

    if args.fanout:
        return fanout_main(args)
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))        # nothing has touched the GPU yet

    # ONE line on stdout: native libraries write there too (RCCL prints a version banner when NCCL_DEBUG=VERSION is in
    # the environment, as on the GPU boxes), so file descriptor 1 is pointed at stderr for the run and the JSON line
    # goes to the real stdout at the end
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    import path_tracing_amd as hpt
    from path_tracing_amd import distributed, scene_io

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback)")
    device_index = local_rank % ndev              # one GPU per rank; ranks share a GPU only in the gloo rehearsal
    torch.cuda.set_device(device_index)
    # the framebuffer gather goes through the process group at every N -- at N = 1 too, so the RCCL call is the
    # code the headline measurement runs
    comm_error = None
    try:
        if world > 1:
            if args.backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device_index))
            else:
                dist.init_process_group(args.backend, rank=rank, world_size=world)
        else:
            import socket
            s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
            kw = {"device_id": torch.device("cuda", device_index)} if args.backend == "nccl" else {}
            dist.init_process_group(args.backend, init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, **kw)
    except Exception as e:                          # N = 1 only: the render does not need the communicator
        if world > 1:
            raise
        comm_error = "%s: %s" % (type(e).__name__, e)

    W = H = args.size
    lights, spheres, tris = scene_io.cornell_with_sphere(args.tris)
    cam = scene_io.make_camera(scene_io.CORNELL_EYE, scene_io.CORNELL_LOOK, scene_io.CORNELL_UP, 50.0, W, H)
    scene = hpt.Scene(lights, spheres, tris)               # upload + BVH build: outside the timed region
    stream = torch.cuda.current_stream().cuda_stream
    base = dict(seed=1, rank=rank, world=world)
    n_local = hpt.local_pixels(W, H, hpt.make_params(**base))
    local = torch.zeros((n_local, 3), dtype=torch.float32, device="cuda")
    image = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda") if rank == 0 else None
    gather_events = []

    def step(flags, spp=None, timed_gather=False):
        p = hpt.make_params(flags=flags, **base)

        def render_local():
            scene.render_pt_device(cam, W, H, args.depth, spp or args.spp, p, local.data_ptr(), stream)
            return local

        def untile(gathered):
            hpt.untile(gathered.data_ptr(), image.data_ptr(), W, H, hpt.make_params(**base), stream)
            return image

        def on_gather(phase):
            if timed_gather:
                e = torch.cuda.Event(enable_timing=True); e.record(); gather_events.append(e)

        return distributed.render_tiled(render_local, untile, rank, world, always_collective=True, on_gather=on_gather)

    def fence():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    base_flags = hpt.FLAG_SINGLE_PIPELINE if args.single_pipeline else 0
    first_image = None
    for i in range(args.warmup):
        step(base_flags)
        if i == 0 and rank == 0:
            first_image = image.clone()
    fence()
    t0 = time.perf_counter()
    ext_ms, ext_n, tot_ms, res_ms, res_n, shd_ms, shd_n = 0.0, 0, 0.0, 0.0, 0, 0.0, 0
    step_render_ms = []                    # this rank's device time per step (HIP events first-to-last kernel)
    for _ in range(args.steps):
        step(base_flags | hpt.FLAG_TIME_KERNELS, timed_gather=True)        # HIP events around every launch, on the launch stream
        st = scene.stats()                 # waits for this rank's render
        step_render_ms.append(st["ms_total"])
        ext_ms += st["ms_extend"] + st["ms_connect"]; ext_n += st["n_extend"] + st["n_connect"]; tot_ms += st["ms_total"]
        res_ms += st["ms_resume"]; res_n += st["n_resume"]; shd_ms += st["ms_shade"]; shd_n += st["n_shade"]
        shade_ms, connect_ms, other_ms = st["ms_shade"], st["ms_connect"], st["ms_other"]
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    gather_ms = [gather_events[i].elapsed_time(gather_events[i + 1]) for i in range(0, len(gather_events) - 1, 2)]
    # every rank's render time of every step, collected after the timed region (not on the critical path)
    rank_ms = None
    if world > 1:
        mine = torch.tensor(step_render_ms, dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        everyone = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(everyone, mine)
        rank_ms = np.array([t.cpu().numpy() for t in everyone])          # [rank][step]
    timed_image = image.clone() if rank == 0 else None

    # ---- untimed: checks of what was just timed + the kernels' exclusive durations + the work counts ----
    verify = {}
    excl = None
    if not args.single_pipeline and not args.no_exclusive_step:
        # same render with one pass in flight at a time: the kernels' durations when they have the device to
        # themselves, and the image the two-pipeline render must reproduce bit for bit
        step(hpt.FLAG_SINGLE_PIPELINE | hpt.FLAG_TIME_KERNELS)
        fence()
        excl = scene.stats()
        if rank == 0:
            verify["timed_image_equals_single_pipeline_image"] = bool(torch.equal(timed_image, image))
    if rank == 0 and first_image is not None:
        verify["timed_image_equals_first_warmup_image"] = bool(torch.equal(timed_image, first_image))
    wc = None
    if not args.no_count_step:
        # the same render again, counting boxes / triangles / rays per kernel (plain single-launch traversal)
        step(hpt.FLAG_COUNT_WORK)
        fence()
        wc = scene.stats()
        if rank == 0:
            verify["timed_image_equals_counting_render"] = bool(torch.equal(timed_image, image))
    if rank == 0:
        verify["timed_image_finite_and_lit"] = bool(torch.isfinite(timed_image).all().item() and timed_image.mean().item() > 0)

    if rank == 0:
        samples = W * H * args.spp
        value = samples * args.steps / dt / 1e6
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
            "exchange": {"backend": dist.get_backend() if dist.is_initialized() else None,
                         "ranks_in_communicator": dist.get_world_size() if dist.is_initialized() else 0,
                         "gather_ms_per_step_rank0": float(np.mean(gather_ms)) if gather_ms else None,
                         "gather_ms_every_step_rank0": [float(x) for x in gather_ms],
                         "gather_bytes_per_rank": int(n_local) * 12, "error": comm_error,
                         "note": "dist.gather of the packed local framebuffers to rank 0 (every step, inside the timed region; at 1 rank it "
                                 "still goes through the communicator), HIP events on the render stream around the call"},
        }
        if rank_ms is not None:
            out["ranks"] = {"render_ms_per_rank_mean_over_steps": rank_ms.mean(axis=1).tolist(),
                            "render_ms_max_rank": float(rank_ms.max(axis=0).mean()), "render_ms_mean_rank": float(rank_ms.mean()),
                            "load_balance_max_over_mean": float((rank_ms.max(axis=0) / rank_ms.mean(axis=0)).mean()),
                            "render_ms_every_step": rank_ms.tolist(),
                            "note": "device time of each rank's render per step (HIP events, first to last kernel); the step's wall time is the "
                                    "slowest rank + the gather + the untile on rank 0"}
        if wc is not None:
            # dominant kernel: k_trace (closest-hit + any-hit BVH traversal).  One trace step per iteration = the first
            # launch (every ray, `split_budget` node steps) + the resume launch (the rays that need more); "launch" below
            # is that pair, its duration the sum of the two HIP-event brackets.  Algorithmic bytes per step:
            # 32 B per BVH node visited (one node = both children's boxes; a node visit slab-tests 2 boxes) + 36 B per
            # triangle tested + 44 B per closest-hit ray (queue index, origin, direction in; hit record out) + 36 B per
            # shadow ray, counted on the plain single-launch traversal of the same rays (the restarts of the split are
            # not algorithmic work).
            boxes = wc["boxes_closest"] + wc["boxes_shadow"]; tri_tests = wc["tris_closest"] + wc["tris_shadow"]
            ext_bytes = 32.0 * (boxes / 2.0) + 36.0 * tri_tests + 44.0 * wc["closest_rays"] + 36.0 * wc["shadow_rays"]
            avg_first_ms = ext_ms / max(ext_n, 1)
            avg_resume_ms = res_ms / max(res_n, 1)
            avg_launch_ms = (ext_ms + res_ms) / max(ext_n, 1)
            launches_per_render = ext_n / max(args.steps, 1)
            bytes_per_launch = ext_bytes / max(launches_per_render, 1)
            achieved = bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else 0.0
            lane_ops = LANE_OPS_PER_BOX * boxes + LANE_OPS_PER_TRI * tri_tests
            trace_s_per_render = (ext_ms + res_ms) / max(args.steps, 1) * 1e-3
            # PMC figures are never measured by this run: quoted only from a file taken from this very build and workload
            pmc = None
            default_workload = (args.size, args.tris, args.depth) == (1024, 100_000, 4) and args.spp >= 128
            if os.path.exists(args.pmc_file) and default_workload:
                try:
                    cand = json.load(open(args.pmc_file))
                    if cand.get("kernel_source_sha") == kernel_source_sha():
                        pmc = cand
                except Exception:
                    pmc = None
            excl_obj = None
            if excl is not None:
                e_first = (excl["ms_extend"] + excl["ms_connect"]) / max(excl["n_extend"] + excl["n_connect"], 1)
                e_step = (excl["ms_extend"] + excl["ms_connect"] + excl["ms_resume"]) / max(excl["n_extend"] + excl["n_connect"], 1)
                excl_obj = {"note": "same render, untimed, HPT_FLAG_SINGLE_PIPELINE: one pass in flight, each launch has the device to itself",
                            "ms_per_step": excl["ms_total"], "avg_first_launch_ms": e_first,
                            "avg_resume_launch_ms": excl["ms_resume"] / max(excl["n_resume"], 1),
                            "achieved": bytes_per_launch / (e_step * 1e-3) / 1e9 if e_step > 0 else None,
                            "avg_shade_launch_ms": excl["ms_shade"] / max(excl["n_shade"], 1)}
            out["roofline"] = {
                "bound": "hbm", "binding_limit": "valu issue (see roofline_valu): the scene's tree and triangles are L2 / Infinity-Cache resident, "
                                                 "so the HBM roofline the contract asks for is reported but is not what limits this kernel",
                "kernel": "k_trace (closest-hit + any-hit BVH traversal; one step = first launch k_trace<false,false> + resume launch k_trace<false,true>)",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "frac_of_measured_copy_peak": achieved / HBM_ACHIEVABLE_GBS,
                "traffic": pmc.get("hbm_bytes_per_launch") if pmc else None,
                "traffic_from": ({"file": os.path.relpath(args.pmc_file, ROOT), "kernel_source_sha": pmc.get("kernel_source_sha"),
                                  "measured_by_this_run": False} if pmc else
                                 "not quoted: no PMC file in profiles/ was taken from this build (sha %s) and workload" % kernel_source_sha()),
                "avg_launch_ms": avg_launch_ms, "avg_first_launch_ms": avg_first_ms, "avg_resume_launch_ms": avg_resume_ms,
                "launches_per_step": launches_per_render, "split_budget": st["split_budget"],
                "long_ray_fraction_last_pass": st["long_rays_last_pass"] / max(st["traced_rays_last_pass"], 1),
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "pipelines_in_flight": 1 if args.single_pipeline else 2,
                "exclusive": excl_obj,
                "round1_accounting": {
                    "achieved": (bytes_per_launch + 32.0 * (boxes / 2.0) / max(launches_per_render, 1)) / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else None,
                    "frac": (bytes_per_launch + 32.0 * (boxes / 2.0) / max(launches_per_render, 1)) / (avg_launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if avg_launch_ms > 0 else None,
                    "note": "for comparison with BENCH_r01 only: round 1's line billed 32 B per child box, i.e. 64 B per node visit (SURVEY 8d says 32 B per node); "
                            "its 0.81 is on that scale, and on a tree that needed 1.7 x the tests per ray"},
                "note": "rank 0's kernels; bytes = 32*nodes visited + 36*tris + 44*closest rays + 36*shadow rays of this rank (SURVEY 8d). "
                        "Two passes of a render are in flight on two streams, so a launch shares the device with the other pipeline's kernels "
                        "and its duration (hence `achieved`) is that of a shared device; `exclusive` has the single-pipeline figures."}
            out["roofline_valu"] = {
                "bound": "valu", "kernel": "k_trace, both launches",
                "box_tests_per_s": boxes / trace_s_per_render if trace_s_per_render > 0 else None,
                "triangle_tests_per_s": tri_tests / trace_s_per_render if trace_s_per_render > 0 else None,
                "achieved": lane_ops / trace_s_per_render / 1e12 if trace_s_per_render > 0 else None,
                "peak": VALU_PEAK_LANE_OPS / 1e12, "unit": "T lane-operations/s",
                "frac": lane_ops / trace_s_per_render / VALU_PEAK_LANE_OPS if trace_s_per_render > 0 else None,
                "valu_issue_utilization_pmc": pmc.get("trace_valu_utilization") if pmc else None,
                "active_lanes_pmc": pmc.get("trace_active_lanes") if pmc else None,
                "note": "useful arithmetic only: %g lane-operations per box slab test, %g per triangle test, of the plain traversal's counts, over the "
                        "time of the trace launches; peak = 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz (one non-packed VALU operation per lane and cycle; the "
                        "157.3 TFLOP/s spec counts a packed FMA as 4). Everything else the kernel issues (node decode, stack, selects, queue and refill "
                        "logic, restarts) is overhead by this measure." % (LANE_OPS_PER_BOX, LANE_OPS_PER_TRI)}
            shade_bytes = 156.0 * wc["path_iters"]       # 28 B material + 2 x 64 B path state per (path, bounce)
            shade_s = shd_ms / max(args.steps, 1) * 1e-3
            out["roofline_shade"] = {
                "bound": "hbm", "binding_limit": "valu issue and the latency of its gathers together (DESIGN.md section 5: a fifth fewer VALU "
                                                 "instructions changed its time by 1 % alone on the device, 3.3 % of the render with two pipelines)",
                "kernel": "k_shade",
                "achieved": shade_bytes / shade_s / 1e9 if shade_s > 0 else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": shade_bytes / shade_s / 1e9 / HBM_PEAK_GBS if shade_s > 0 else None,
                "avg_launch_ms": shd_ms / max(shd_n, 1), "launches_per_step": shd_n / max(args.steps, 1),
                "algorithmic_bytes_per_path_iteration": 156.0,
                "traffic_bytes_per_path_iteration_pmc": pmc.get("shade_bytes_per_path_iteration") if pmc else None,
                "valu_issue_utilization_pmc": pmc.get("shade_valu_utilization") if pmc else None}
            all_rays = wc["closest_rays"] + wc["shadow_rays"]
            per_sample = (32.0 * boxes / 2.0 + 36.0 * tri_tests + (28.0 + 128.0) * wc["path_iters"]) / max(wc["samples"], 1) + 12.0 / args.spp
            out["work"] = {"rays_per_sample": all_rays / max(wc["samples"], 1),
                           "nodes_per_ray": boxes / 2.0 / max(all_rays, 1),
                           "tris_per_ray": tri_tests / max(all_rays, 1),
                           "path_iterations_per_sample": wc["path_iters"] / max(wc["samples"], 1),
                           "algorithmic_bytes_per_sample": per_sample,
                           "Mrays_per_s": all_rays * world / (dt / args.steps) / 1e6 if world == 1 else None,
                           "device_ms_per_step_rank0": tot_ms / args.steps,
                           "kernel_ms_last_step_rank0": {"trace": st["ms_extend"] + st["ms_connect"] + st["ms_resume"], "shade": shade_ms, "other": other_ms},
                           "bvh_nodes": wc["bvh_nodes"], "bvh_depth": wc["bvh_depth"], "ms_bvh_build": wc["ms_bvh_build"]}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], ref_win, ref_spp = cpu_baseline(lights, spheres, tris, W, H, args.depth)
            # the oracle's window against the GPU render of the same samples (same seed, same spp)
            step(base_flags, spp=ref_spp)
            torch.cuda.synchronize()
            x0, y0, x1, y1 = WINDOW
            got = image[y0:y1, x0:x1].cpu().numpy()
            d = got.astype(np.float64) - ref_win.astype(np.float64)
            verify["oracle_window"] = {"window": list(WINDOW), "spp": ref_spp, "rmse": float(np.sqrt((d * d).mean())),
                                       "max_abs": float(np.abs(d).max()), "bit_identical": bool(np.array_equal(got, ref_win)),
                                       "oracle": "oracle/pt_oracle.cpp (brute-force scans), the pt_port render timed above"}
            # the roofline's numerator, checked: device work counts of a small counting render of this scene == the oracle's host walk of
            # the tree the device holds ("counted by the build's deterministic CPU restatement traversing the same BVH")
            import oracle
            cw = 256
            cam_small = scene_io.make_camera(scene_io.CORNELL_EYE, scene_io.CORNELL_LOOK, scene_io.CORNELL_UP, 50.0, cw, cw)
            small = scene.render_pt(cam_small, cw, cw, args.depth, 2, hpt.make_params(seed=1, flags=hpt.FLAG_COUNT_WORK))
            dev_counts = scene.stats()
            ref_small, host_counts = oracle.pt_render(lights, spheres, tris, cam_small, cw, cw, args.depth, 2, seed=1, bvh=scene.export_bvh())
            keys = ("closest_rays", "shadow_rays", "boxes_closest", "tris_closest", "boxes_shadow", "tris_shadow")
            verify["work_counts_equal_host_walk"] = {"render": "%dx%d x 2 spp of the bench scene" % (cw, cw),
                                                     "device": {k: int(dev_counts[k]) for k in keys}, "host_walk": {k: int(host_counts[k]) for k in keys},
                                                     "equal": all(int(dev_counts[k]) == int(host_counts[k]) for k in keys),
                                                     "images_bit_identical": bool(np.array_equal(small, ref_small))}
        out["verify"] = verify
        real_stdout.write(json.dumps(out) + "\n")
        real_stdout.flush()
    scene.close()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    bad = [k for k, v in verify.items() if v is False] if rank == 0 else []
    if rank == 0 and "work_counts_equal_host_walk" in verify and not (verify["work_counts_equal_host_walk"]["equal"] and verify["work_counts_equal_host_walk"]["images_bit_identical"]):
        bad.append("work_counts_equal_host_walk")
    if rank == 0 and (bad or ("oracle_window" in verify and not verify["oracle_window"]["rmse"] < 1e-3)):
        raise SystemExit("bench.py: verification failed: %s" % (bad or verify["oracle_window"]))


if __name__ == "__main__":
    main()
