"""N > 1 path on the CPU: world_size 2 and 3 over gloo (no GPU): shard layout, gather, reassembly."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,tile", [(2, 8), (3, 16)])
def test_gloo_gather_reassembles_the_frame(dsrt, world, tile):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), DSRT_TILE=str(tile))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_gloo_worker.py")], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=240) for p in procs]
    for p, (out, err) in zip(procs, outs):
        assert p.returncode == 0, err[-2000:]
    reports = [json.loads([l for l in out.splitlines() if l.startswith("{")][-1]) for out, _ in outs]
    assert all(r["ok"] for r in reports)
    assert sum(r["tiles"] for r in reports) == -(-44 // tile) * -(-27 // tile)


def test_shard_index_map_covers_every_pixel_once(dsrt):
    from dsrt_amd import dist as shard
    for W, H, world, tile in ((44, 27, 2, 8), (1920, 1080, 8, 8), (200, 112, 3, 16), (9, 9, 4, 8)):
        seen = np.zeros(W * H, np.int32)
        for r in range(world):
            idx = shard.shard_pixel_indices(W, H, r, world, tile)
            lay = dsrt.shard_layout(dsrt.make_desc(W, H, 1, 1, tile_size=tile, shard_rank=r, shard_count=world))
            assert idx.size * 3 == lay["rgb8_bytes_padded"]
            np.add.at(seen, idx[idx >= 0], 1)
        assert (seen == 1).all(), (W, H, world, tile)


def test_eight_rank_layouts_of_the_benchmark_frame(dsrt):
    """BASELINE.json configs[3]/[4] at N = 8 without a GPU: tile counts, padding and per-rank ownership for 1920x1080 at the tile sizes
    the kernel accepts (the 8-GPU run itself is the driver's), and the frame dealing of the sequence job."""
    from dsrt_amd import dist as shard
    from dsrt_amd import sequence
    W, H, world = 1920, 1080, 8
    for tile, tiles in ((8, 240 * 135), (16, 120 * 68), (32, 60 * 34)):
        owned = []
        for r in range(world):
            lay = dsrt.shard_layout(dsrt.make_desc(W, H, 1, 1, tile_size=tile, shard_rank=r, shard_count=world))
            assert lay["tiles_total"] == tiles
            assert lay["tiles_per_shard_padded"] == -(-tiles // world)               # equal on every rank: one gather with equal counts
            assert lay["rgb8_bytes_padded"] == lay["tiles_per_shard_padded"] * tile * tile * 3
            owned.append(lay["tiles_this_shard"])
            idx = shard.shard_pixel_indices(W, H, r, world, tile)
            assert int(((idx >= 0).reshape(-1, tile * tile).any(axis=1)).sum()) == lay["tiles_this_shard"]
        assert sum(owned) == tiles and max(owned) - min(owned) <= 1
    # 32x8-style counts of SURVEY.md 8(e): 8,100 tiles -> 1,013 per rank, the last ranks own one fewer
    geo = shard.tile_geometry(1920, 1080 * 2, 32, 8)                                  # 60 x 68 tiles = 4,080; any ragged count pads up
    assert geo["tiles_per_shard_padded"] * 8 >= geo["tiles_total"]
    # sequence: 99 poses over 8 ranks, dealt round-robin; every frame exactly once, loads differ by at most one frame
    frames = list(range(99))
    dealt = [sequence.frame_assignment(frames, r, 8, "frames") for r in range(8)]
    assert sorted(f for part in dealt for f in part) == frames
    assert {len(p) for p in dealt} == {12, 13} and dealt[3][:3] == [3, 11, 19]
    assert sequence.frame_assignment(frames, 5, 8, "tiles") == frames                # tile split: every rank renders its tiles of every frame
    assert sequence.frame_assignment(frames, 0, 1, "frames") == frames
    # dealt by estimated cost (the approach: the nearest frame costs about 30 times the farthest): every frame still exactly once, the same
    # partition whichever rank computes it, and the ranks' estimated loads within a few per cent where round-robin leaves a quarter idle
    sep = [1787.0 - i * (1787.0 - 35.7) / 98.0 for i in frames]
    costs = [sequence.approach_cost(s_m, 50.0) for s_m in sep]
    by_cost = [sequence.frame_assignment(frames, r, 8, "frames", costs) for r in range(8)]
    assert sorted(f for part in by_cost for f in part) == frames and all(part == sorted(part) for part in by_cost)
    assert by_cost == [sequence.frame_assignment(list(frames), r, 8, "frames", list(costs)) for r in range(8)]

    def spread(parts):
        loads = [sum(costs[f] for f in part) for part in parts]
        return max(loads) / (sum(loads) / len(loads))
    assert spread(by_cost) < 1.03 < 1.15 < spread(dealt)
    with pytest.raises(ValueError):
        sequence.frame_assignment(frames, 0, 8, "frames", costs[:-1])


def test_multi_api_is_exported_and_fails_cleanly_without_a_gpu(dsrt):
    """dsrt_multi_* load and bind on a CPU-only machine (no compute call): creation reports the missing device, nothing crashes."""
    import ctypes as C
    for name in ("dsrt_multi_create", "dsrt_multi_destroy", "dsrt_multi_scene_upload", "dsrt_multi_render_frame", "dsrt_multi_render_sequence",
                 "dsrt_ctx_clone", "dsrt_microbench_gather"):
        assert hasattr(dsrt.lib, name), name
    h = C.c_void_p()
    devs = (C.c_int * 2)(0, 1)
    assert dsrt.lib.dsrt_multi_create(devs, 0, 1, C.byref(h)) == -1 and not h.value       # n < 1: DSRT_ERR_INVALID
    if dsrt.lib.dsrt_device_count() == 0:
        rc = dsrt.lib.dsrt_multi_create(devs, 2, 1, C.byref(h))
        assert rc < 0 and not h.value and dsrt.lib.dsrt_last_error()
    dsrt.lib.dsrt_multi_destroy(None)
