"""The HOST half of the certified second tree (include/dsrt.h: dsrt_ctx_set_certified_tree), without a GPU: what dsrt_scene_upload prepares from a scene that carries the
reference's tree -- which triangles the reference walk can never reach (a zero-thickness box on their root-to-leaf path: bbox_hit's `t_max <= t_min` holds with
equality, src/gpu_render.cu:312), every triangle's leaf box on the reference tree, and a widened binned-SAH tree over the reachable triangles -- checked against an
independent numpy walk of the same reference tree and against the structural properties the kernel's certificate relies on."""
import ctypes as C

import numpy as np
import pytest

from conftest import load_world


def _probe(dsrt, hs):
    n_tris = len(hs.arrays()["tris"])
    counts, pad = (C.c_int * 6)(), C.c_float()
    unreachable = np.zeros(n_tris, np.uint8)
    leaf_box = np.zeros((n_tris, 6), np.float32)
    nodes = np.zeros(max(1, 2 * n_tris), dsrt.capi.NODE_DTYPE)
    order = np.zeros(max(1, n_tris), np.int32)
    rc = dsrt.lib.dsrt_host_scene_second_tree_probe(hs._h, counts, C.byref(pad), unreachable.ctypes.data, leaf_box.ctypes.data, nodes.ctypes.data, len(nodes), order.ctypes.data, len(order))
    assert rc == 0, dsrt.lib.dsrt_last_error()
    c = list(counts)
    return c, pad.value, unreachable, leaf_box, nodes[:c[3]], order[:c[2]]


def _reference_facts(arr):
    """Independent walk of the reference tree: per triangle, its leaf's box and whether a zero-thickness box lies on its path."""
    nodes, idx = arr["nodes"], arr["idx"]
    n = len(arr["tris"])
    dead_tri, box = np.zeros(n, np.uint8), np.zeros((n, 6), np.float32)
    stack = [(0, False)]
    while stack:
        i, dead = stack.pop()
        nd = nodes[i]
        dead = dead or bool(((nd["bbox_max"] - nd["bbox_min"]) == 0).any())
        if nd["tri_count"] > 0:
            t = idx[nd["tri_offset"]:nd["tri_offset"] + nd["tri_count"]]
            box[t, :3], box[t, 3:] = nd["bbox_min"], nd["bbox_max"]
            if dead:
                dead_tri[t] = 1
        else:
            stack += [(int(nd["left"]), dead), (int(nd["right"]), dead)]
    return dead_tri, box


@pytest.mark.parametrize("world", ["station_3k", "mixed", "quirks", "textured"])
def test_second_tree_preparation(dsrt, world):
    hs = load_world(dsrt, world)
    arr = hs.arrays()
    tris = arr["tris"]
    counts, pad, unreachable, leaf_box, nodes, order = _probe(dsrt, hs)
    want_dead, want_box = _reference_facts(arr)
    assert counts[0] == len(tris) and counts[1] == int(want_dead.sum())
    assert np.array_equal(unreachable, want_dead)
    assert np.array_equal(leaf_box.view(np.uint32), want_box.view(np.uint32))
    # the second tree holds every REACHABLE triangle exactly once and no other
    assert counts[2] == len(order) == len(tris) - int(want_dead.sum())
    assert sorted(order.tolist()) == np.flatnonzero(want_dead == 0).tolist()
    root = arr["nodes"][0]
    extent = float((root["bbox_max"] - root["bbox_min"]).max())
    assert pad == np.float32(extent) * np.float32(1.0 / 65536.0) and pad > 0
    # structure: a proper binary tree in pre-order whose leaves partition `order`, every node box containing its children's, every leaf box containing its triangles
    # widened by the pad (what makes the walk conservative against the rounding of the slab arithmetic)
    seen_leaf_entries, stack, height = 0, [(0, 1)], 0
    visited = np.zeros(len(nodes), bool)
    while stack:
        i, level = stack.pop()
        assert not visited[i]
        visited[i] = True
        nd = nodes[i]
        height = max(height, level)
        if nd["tri_count"] > 0:
            assert nd["tri_count"] <= 4
            t = order[nd["tri_offset"]:nd["tri_offset"] + nd["tri_count"]]
            v = tris["v"][t].reshape(-1, 3)
            assert (nd["bbox_min"] <= v.min(axis=0) - np.float32(pad) * np.float32(0.999)).all() and (nd["bbox_max"] >= v.max(axis=0) + np.float32(pad) * np.float32(0.999)).all()
            seen_leaf_entries += int(nd["tri_count"])
        else:
            for child in (int(nd["left"]), int(nd["right"])):
                assert 0 < child < len(nodes)
                assert (nodes[child]["bbox_min"] >= nd["bbox_min"]).all() and (nodes[child]["bbox_max"] <= nd["bbox_max"]).all()
                stack.append((child, level + 1))
    assert visited.all() and seen_leaf_entries == len(order) and height == counts[4]
    assert counts[5] == 1                                            # the test worlds' spheres sit beside their meshes


def test_the_mixed_scene_has_a_panel_the_reference_tree_never_reaches(dsrt):
    """a leaf of coplanar axis-aligned triangles has a zero-thickness box, which bbox_hit never passes.  `mixed` has such a panel: it must be in the unreachable
    set (and therefore absent from the second tree), or the second tree would render geometry the reference does not."""
    counts, _, unreachable, _, _, _ = _probe(dsrt, load_world(dsrt, "mixed"))
    assert counts[1] > 0 and counts[1] == int(unreachable.sum())


def test_a_scene_with_a_far_sphere_keeps_the_reference_tree_only(dsrt):
    capi = dsrt.capi
    hs0 = load_world(dsrt, "station_3k")
    arr = hs0.arrays()
    sph = np.zeros(1, capi.SPHERE_DTYPE)
    sph["center"], sph["radius"] = (0.0, -1.0e5, 0.0), 9.0e4        # a "ground" far below a 100 m station: rays would start 100 extents away
    hs = dsrt.HostScene().add_arrays(tris=arr["tris"], spheres=sph, mats=arr["mats"])
    hs.build_bvh()
    counts, _, _, _, _, _ = _probe(dsrt, hs)
    assert counts[5] == 0
    near = sph.copy()
    near["center"], near["radius"] = (0.0, 40.0, 0.0), 5.0
    hs2 = dsrt.HostScene().add_arrays(tris=arr["tris"], spheres=near, mats=arr["mats"])
    hs2.build_bvh()
    assert _probe(dsrt, hs2)[0][5] == 1
