"""One process per GPU: every rank renders its image tiles, rank 0 gathers the framebuffer.

Pixels are independent (reference src/pt_cu.cu:27-35), so the only exchange step of the path
is the framebuffer gather.  It is one fixed-size `gather` to rank 0 (RCCL over xGMI when the
backend is "nccl": each sender uses its own link to the root; nothing is ringed), followed
by the un-tiling scatter on the root.  The random streams are keyed by global pixel and
sample index, so the assembled image does not depend on the number of ranks.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def gather_framebuffer(local: torch.Tensor, rank: int, world: int, root: int = 0, always_collective: bool = False):
    """local: [n_local, 3] float32 tensor (same shape on every rank).  Returns on the root a
    [world, n_local, 3] tensor laid out [rank][local slot]; None elsewhere.

    world == 1 skips the collective unless `always_collective` is set and a process group exists:
    then the one-rank gather goes through the backend too (bench.py does this so that the RCCL
    gather is the code that runs at every N, including the N = 1 headline measurement)."""
    if world == 1 and not (always_collective and dist.is_available() and dist.is_initialized()):
        return local.unsqueeze(0)
    if dist.get_backend() == "gloo" and local.is_cuda:
        # rehearsal on a one-GPU box: gloo moves host memory
        host = local.cpu()
        if rank == root:
            out = torch.empty((world,) + tuple(host.shape), dtype=host.dtype)
            dist.gather(host, list(out.unbind(0)), dst=root)
            return out.to(local.device)
        dist.gather(host, None, dst=root)
        return None
    if rank == root:
        out = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
        dist.gather(local, list(out.unbind(0)), dst=root)
        return out
    dist.gather(local, None, dst=root)
    return None


def render_tiled(render_local_fn, untile_fn, rank: int, world: int, always_collective: bool = False, on_gather=None):
    """Driver shared by bench.py and the CPU tests.
    render_local_fn() -> this rank's packed local framebuffer tensor [n_local, 3];
    untile_fn(gathered [world, n_local, 3]) -> row-major image (root only);
    on_gather(phase) is called with "begin" / "end" around the exchange step (timing hooks)."""
    local = render_local_fn()
    if on_gather:
        on_gather("begin")
    gathered = gather_framebuffer(local, rank, world, always_collective=always_collective)
    if on_gather:
        on_gather("end")
    if gathered is None:
        return None
    return untile_fn(gathered)
