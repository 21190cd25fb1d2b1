"""Image-tile partition of the framebuffer across devices (host-side mirror).

The image is cut into tile x tile squares numbered row by row, every row rotated by its index (tile g
sits in row g // tiles_x at column (g % tiles_x + row) % tiles_x); device `rank` of `world` renders
tiles rank, rank+world, ... into a packed local framebuffer, which spreads its tiles diagonally over
the image instead of in columns.  Inside a tile, slots
run over 8x8 pixel blocks (one wave64 of primary rays = one 8x8 block).  Every rank's local
buffer has the same size ceil(ntiles/world) * tile^2 so the gather is a plain fixed-size
collective; slots past the image edge or past the last tile stay zero.

The device code uses the same map (csrc/pt_kernels.hip: tile_to_pixel, k_untile); these
numpy versions are the host logic used by the multi-rank driver and its CPU tests.
"""
from __future__ import annotations

import numpy as np


def tiling_dims(W: int, H: int, tile: int = 32, world: int = 1):
    if tile % 8 != 0 or tile <= 0:
        raise ValueError("tile must be a positive multiple of 8")
    tiles_x = (W + tile - 1) // tile
    tiles_y = (H + tile - 1) // tile
    ntiles = tiles_x * tiles_y
    n_local = ((ntiles + world - 1) // world) * tile * tile
    return tiles_x, tiles_y, ntiles, n_local


def local_to_pixel(W: int, H: int, tile: int, rank: int, world: int):
    """For every local slot p of `rank`: (x, y, valid)."""
    tiles_x, tiles_y, ntiles, n_local = tiling_dims(W, H, tile, world)
    p = np.arange(n_local, dtype=np.int64)
    ts2 = tile * tile
    lt, q = p // ts2, p % ts2
    gt = lt * world + rank
    ty = gt // tiles_x
    tx = (gt % tiles_x + ty) % tiles_x          # every tile row is rotated by its index (diagonal stripes per rank)
    sub, l = q >> 6, q & 63
    spr = tile >> 3
    bx, by = sub % spr, sub // spr
    x = tx * tile + bx * 8 + (l & 7)
    y = ty * tile + by * 8 + (l >> 3)
    valid = (gt < ntiles) & (x < W) & (y < H)
    return x, y, valid


def tile_image(image: np.ndarray, tile: int, rank: int, world: int) -> np.ndarray:
    """Row-major [H, W, C] image -> this rank's packed local buffer [n_local, C]."""
    H, W = image.shape[:2]
    x, y, valid = local_to_pixel(W, H, tile, rank, world)
    out = np.zeros((len(x),) + image.shape[2:], image.dtype)
    out[valid] = image[y[valid], x[valid]]
    return out


def untile_image(gathered: np.ndarray, W: int, H: int, tile: int, world: int) -> np.ndarray:
    """[world, n_local, C] packed local buffers -> row-major [H, W, C] image."""
    out = np.zeros((H, W) + gathered.shape[2:], gathered.dtype)
    for r in range(world):
        x, y, valid = local_to_pixel(W, H, tile, r, world)
        out[y[valid], x[valid]] = gathered[r][valid]
    return out
