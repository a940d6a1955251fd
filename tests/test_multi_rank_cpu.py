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
