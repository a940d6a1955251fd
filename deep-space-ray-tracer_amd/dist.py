"""Screen-tile sharding across the GPUs of one node: who renders what, the gather, and the reassembly.

The reference has no multi-GPU code at all.  A pixel's value depends only on (x, y, W, seed, scene)
(src/gpu_render.cu:983-999), so the image shards into independent units with no exchange during rendering; the only
communication is bringing the finished tiles to rank 0.

Layout (shared with the kernel, render_kernel.hip ST_FETCH, and with dsrt_deinterleave_tiles):
  * the image is cut into T x T tiles (T a multiple of 8, default 8), numbered row-major from the top-left;
  * tile g belongs to rank g mod N -- interleaved, so every rank sees the same mix of station and background;
  * rank r stores its k-th tile (g = k*N + r) at bytes [k*T*T*3, (k+1)*T*T*3) of its compact buffer, rows of the tile top
    first, pixels outside the image left as they were;
  * every rank's buffer is padded to ceil(tiles/N) tiles so that one gather with equal counts works.
One process per GPU; torch.distributed ("nccl" is RCCL on ROCm; "gloo" in the CPU tests) carries the gather.
"""
import numpy as np


def tile_geometry(width, height, tile=8, world=1):
    tile = tile or 8
    tiles_x, tiles_y = -(-width // tile), -(-height // tile)
    total = tiles_x * tiles_y
    return {"tile": tile, "tiles_x": tiles_x, "tiles_y": tiles_y, "tiles_total": total, "tiles_per_shard_padded": -(-total // max(1, world))}


def shard_pixel_indices(width, height, rank, world, tile=8):
    """For rank's compact buffer: image-linear pixel index (row*W + x) of every compact pixel slot, -1 where the slot is
    padding or lies outside the image.  Shape (tiles_per_shard_padded * T * T,)."""
    g = tile_geometry(width, height, tile, world)
    T = g["tile"]
    out = np.full(g["tiles_per_shard_padded"] * T * T, -1, np.int64)
    iy, ix = np.divmod(np.arange(T * T), T)
    for k in range(g["tiles_per_shard_padded"]):
        t = k * world + rank
        if t >= g["tiles_total"]:
            break
        ty, tx = divmod(t, g["tiles_x"])
        x, row = tx * T + ix, ty * T + iy
        ok = (x < width) & (row < height)
        out[k * T * T:(k + 1) * T * T] = np.where(ok, row * width + x, -1)
    return out


def deinterleave_host(gathered, width, height, world, tile=8):
    """numpy mirror of dsrt_deinterleave_tiles: `gathered` is the concatenation of the ranks' compact rgb8 buffers."""
    g = tile_geometry(width, height, tile, world)
    per = g["tiles_per_shard_padded"] * g["tile"] * g["tile"]
    gathered = np.asarray(gathered, np.uint8).reshape(world, per, 3)
    image = np.zeros((height * width, 3), np.uint8)
    for r in range(world):
        idx = shard_pixel_indices(width, height, r, world, tile)
        ok = idx >= 0
        image[idx[ok]] = gathered[r][ok]
    return image.reshape(height, width, 3)


def gather_to_root(part, world, rank, dst=0):
    """One gather of the equal-sized compact buffers to rank `dst`; returns the concatenated tensor there, None elsewhere.
    `part` is a torch uint8 tensor on the device the process group uses."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return part
    if part.is_cuda and dist.get_backend() == "gloo":
        # rehearsal of the N-rank flow on fewer GPUs than ranks (bench.py DSRT_BENCH_REHEARSAL): gloo carries host copies
        host = part.cpu()
        bucket = [torch.empty_like(host) for _ in range(world)] if rank == dst else None
        dist.gather(host, bucket, dst=dst)
        return torch.cat(bucket).to(part.device) if rank == dst else None
    bucket = [torch.empty_like(part) for _ in range(world)] if rank == dst else None
    dist.gather(part, bucket, dst=dst)
    return torch.cat(bucket) if rank == dst else None
