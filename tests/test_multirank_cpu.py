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


def test_one_rank_gather_goes_through_the_backend_when_asked(tmp_path):
    """always_collective: the world-1 gather runs through the process group (bench.py does this with RCCL
    so that the collective is exercised at N = 1); without a group, or without the flag, it is a view."""
    from path_tracing_amd import distributed
    local = torch.arange(24, dtype=torch.float32).reshape(8, 3)
    assert distributed.gather_framebuffer(local, 0, 1).data_ptr() == local.data_ptr()
    assert distributed.gather_framebuffer(local, 0, 1, always_collective=True).data_ptr() == local.data_ptr()   # no group yet
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1)
    try:
        calls = []
        out = distributed.render_tiled(lambda: local, lambda g: g.clone(), 0, 1, always_collective=True, on_gather=calls.append)
        assert calls == ["begin", "end"]
        assert out.shape == (1, 8, 3) and torch.equal(out[0], local)
        g = distributed.gather_framebuffer(local, 0, 1, always_collective=True)
        assert g.data_ptr() != local.data_ptr() and torch.equal(g[0], local)
    finally:
        dist.destroy_process_group()


def test_bench_starts_its_own_ranks_as_a_child_process(monkeypatch):
    """`python bench.py --gpus N` without a launcher: torch.distributed.run is started as a CHILD (subprocess),
    before anything imports torch.cuda, with the rendezvous on 127.0.0.1; under a launcher (RANK set) it is a rank."""
    import importlib
    import subprocess
    bench = importlib.import_module("bench")
    seen = {}

    class Done:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        seen["cmd"] = cmd; seen["env"] = env
        return Done()

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.delenv("RANK", raising=False); monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7                                   # the child's exit code is relayed
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "2"] and os.path.basename(cmd[-5]) == "bench.py"
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


@pytest.mark.parametrize("G", [2, 4, 8])
def test_gather_bytes_cover_the_tiled_image(hpt, G):
    """What N ranks send to the root in one step: `exchange.gather_bytes_per_rank` (12 B per packed local framebuffer
    slot) x N == 12 B x the pixels of the image rounded up to whole 32 x 32 tiles and to an equal number of tiles per
    rank -- at config 5's 4096^2 and at a size that is not a multiple of the tile."""
    for W, H in ((4096, 4096), (1000, 700)):
        tile = 32
        tiles = ((W + tile - 1) // tile) * ((H + tile - 1) // tile)
        per_rank_tiles = (tiles + G - 1) // G
        n_local = [hpt.local_pixels(W, H, hpt.make_params(rank=r, world=G)) for r in range(G)]
        assert len(set(n_local)) == 1                                     # every rank's buffer has the same size: one fixed-size gather
        gather_bytes_per_rank = n_local[0] * 12
        assert gather_bytes_per_rank * G == 12 * per_rank_tiles * G * tile * tile
        assert gather_bytes_per_rank * G >= 12 * W * H
        if W % tile == 0 and H % tile == 0 and tiles % G == 0:
            assert gather_bytes_per_rank * G == 12 * W * H                # config 5: 201 326 592 B in all, 25 165 824 B per rank at G = 8
