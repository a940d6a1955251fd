"""Worker for tests/test_multi_rank_cpu.py: one rank of a gloo process group on the CPU.

What runs here is the N > 1 plumbing of bench.py without a GPU: the shard layout from the C ABI (dsrt_shard_layout), each
rank filling its compact tile buffer, the gather, and the reassembly.  The pixel values come from the CPU oracle (the
checker), rendered for the whole small image on every rank and then cut down to the rank's own tiles -- enough to prove
that tiles land where they should; the GPU-side equivalent is tests/test_gpu_parity.py::test_tile_shards_*.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import torch.distributed as dist
    import dsrt_amd as d
    from dsrt_amd import dist as shard
    from conftest import Oracle, load_world

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    W, H, spp, tile = 44, 27, 2, int(os.environ.get("DSRT_TILE", "8"))
    hs = load_world(d, "lights")
    cam = d.camera_look_at((0.0, 3.0, 9.0), (0.0, 2.0, 0.0), 45.0, W, H, spp, 6)
    scene = hs.view(cam, (0.32780463, -0.7564221, 0.5660121))
    full, _, _ = Oracle().render(scene, W, H, want_f32=False)

    desc = d.make_desc(W, H, spp, 6, tile_size=tile, shard_rank=rank, shard_count=world)
    lay = d.shard_layout(desc)
    geo = shard.tile_geometry(W, H, tile, world)
    assert lay["tiles_total"] == geo["tiles_total"] and lay["tiles_per_shard_padded"] == geo["tiles_per_shard_padded"]
    assert lay["rgb8_bytes_padded"] == geo["tiles_per_shard_padded"] * tile * tile * 3
    idx = shard.shard_pixel_indices(W, H, rank, world, tile)
    mine = int(((idx >= 0).reshape(-1, tile * tile).any(axis=1)).sum())
    assert mine == lay["tiles_this_shard"]
    part = np.full((idx.size, 3), 201, np.uint8)                      # padding keeps a sentinel value
    part[idx >= 0] = full.reshape(-1, 3)[idx[idx >= 0]]
    t = torch.from_numpy(part.reshape(-1).copy())
    got = shard.gather_to_root(t, world, rank)
    ok = True
    if rank == 0:
        image = shard.deinterleave_host(got.numpy(), W, H, world, tile)
        ok = bool(np.array_equal(image, full))
    flag = torch.tensor([1 if ok else 0])
    dist.broadcast(flag, src=0)
    dist.barrier()
    dist.destroy_process_group()
    print(json.dumps({"rank": rank, "ok": bool(flag.item()), "tiles": mine}))
    sys.exit(0 if flag.item() else 1)


if __name__ == "__main__":
    main()
