"""world_size-2 (and 3) test of the multi-rank driver on CPU with the gloo backend: every rank
produces its packed local framebuffer, rank 0 gathers and un-tiles (path_tracing_amd/distributed.py
is the code bench.py runs with the nccl/RCCL backend; here the oracle stands in for the GPU render)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN, ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, W, H, tile, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle
        from path_tracing_amd import distributed, scene_io, tiling
        sc = scene_io.load_scene(os.path.join(GOLDEN, "scenes", "input.txt"))
        L, sp, tr = scene_io.flatten_for_pt(sc)
        cam = scene_io.camera_for(sc, W, H)

        def render_local():
            # stand-in for Scene.render_pt_device: the oracle renders, the host tiling map packs
            img, _ = oracle.pt_render(L, sp, tr, cam, W, H, 4, 2, seed=13, threads=2)
            return torch.from_numpy(tiling.tile_image(img, tile, rank, world))

        def untile(gathered):
            return tiling.untile_image(gathered.numpy(), W, H, tile, world)

        dist.barrier()
        image = distributed.render_tiled(render_local, untile, rank, world)
        if rank == 0:
            np.save(out_path, image)
        else:
            assert image is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,W,H,tile", [(2, 48, 40, 16), (3, 50, 37, 8)])
def test_gather_and_untile_over_gloo(tmp_path, oracle_mod, sio, world, W, H, tile):
    out = str(tmp_path / "img.npy")
    mp.spawn(_worker, args=(world, _free_port(), W, H, tile, out), nprocs=world, join=True)
    got = np.load(out)
    sc = sio.load_scene(os.path.join(GOLDEN, "scenes", "input.txt"))
    L, sp, tr = sio.flatten_for_pt(sc)
    ref, _ = oracle_mod.pt_render(L, sp, tr, sio.camera_for(sc, W, H), W, H, 4, 2, seed=13)
    assert np.array_equal(got, ref)
