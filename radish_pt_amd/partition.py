"""Host-side arithmetic of the screen-tile partition (mirrors makePixelMap / mapPixel / k_untile in csrc).

The frame is cut into `tile` x `tile` pixel tiles (row-major tile ids); tile t belongs to rank t % world and is that
rank's local tile t // world.  Each rank renders into a packed buffer [tiles_per_rank][tile*tile][3]; the all-gather
of those buffers is [world][tiles_per_rank][tile*tile][3], and `untile_indices` says where each frame pixel lives in it.
"""
import numpy as np


def tiles_per_rank(width, height, world, tile=64):
    tiles_x = (width + tile - 1) // tile
    tiles_y = (height + tile - 1) // tile
    return (tiles_x * tiles_y + world - 1) // world


def shard_elems(width, height, world, tile=64):
    """Pixels in one rank's packed buffer (including padding tiles and off-frame pixels of edge tiles)."""
    return tiles_per_rank(width, height, world, tile) * tile * tile


def rank_pixels(width, height, rank, world, tile=64):
    """(frame pixel index, packed index) pairs of the pixels rank `rank` renders."""
    tiles_x = (width + tile - 1) // tile
    y, x = np.mgrid[0:height, 0:width]
    tid = (y // tile) * tiles_x + (x // tile)
    mine = (tid % world) == rank
    packed = (tid // world) * (tile * tile) + (y % tile) * tile + (x % tile)
    return (y * width + x)[mine].astype(np.int64), packed[mine].astype(np.int64)


def untile_indices(width, height, world, tile=64):
    """Index into the gathered buffer (flattened over [world][tiles_per_rank][tile*tile]) for every frame pixel."""
    tiles_x = (width + tile - 1) // tile
    tpr = tiles_per_rank(width, height, world, tile)
    y, x = np.mgrid[0:height, 0:width]
    tid = (y // tile) * tiles_x + (x // tile)
    src = ((tid % world) * tpr + tid // world) * (tile * tile) + (y % tile) * tile + (x % tile)
    return src.reshape(-1).astype(np.int64)
