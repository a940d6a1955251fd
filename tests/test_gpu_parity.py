"""HIP path (through the C ABI) vs the CPU oracle on the same inputs.  Bar: bit-exact -- identical rgb8 bytes AND identical
fp32 bit patterns of the pre-quantisation colour, i.e. per-pixel L-infinity = 0 <= the 1e-3 BASELINE.json asks for.

The oracle is oracle/dsrt_oracle.c (restatement of src/gpu_render.cu; pinning status in oracle/dsrt_oracle.h).
"""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ASSETS, GOLDEN, load_world
from test_oracle import CASES, SUN

pytestmark = pytest.mark.gpu

LINF_TOLERANCE = 1e-3          # BASELINE.json north_star; we assert 0 and report against this


def _scene(dsrt, name):
    world, cam_args, spp = CASES[name]
    hs = load_world(dsrt, world)
    W, H = cam_args[3], cam_args[4]
    cam = dsrt.camera_look_at(cam_args[0], cam_args[1], cam_args[2], W, H, spp, cam_args[5])
    return hs, hs.view(cam, SUN), W, H, spp, cam_args[5]


def test_device_math_is_bit_identical_to_host(dsrt, gpu_ctx, oracle):
    r1 = (np.arange(0, 1 << 24, 13, dtype=np.uint32).astype(np.float32)) / np.float32(16777216.0)
    phi = (np.float32(2.0) * np.float32(3.14159265358979323846)) * r1
    extra = np.float32(np.random.default_rng(3).uniform(-50, 50, 50000))
    x = np.concatenate([phi, extra])
    for fn, name in ((0, "sinf"), (1, "cosf")):
        got = gpu_ctx.selftest_math(fn, x)
        f = getattr(oracle.lib, "dsrt_oracle_" + name)
        want = np.array([f(float(v)) for v in x[::7]], np.float32)
        assert np.array_equal(got[::7].view(np.uint32), want.view(np.uint32)), name
    base = np.concatenate([np.float32(np.random.default_rng(4).uniform(0, 10, 40000)), np.float32([0.0, 1.0, 10.0, 1e-30, 0.25])])
    for y in (0.5, 1.0 / 2.2, 5.0, 1.0, 2.0, 0.4):
        got = gpu_ctx.selftest_math(2, base, np.float32(y))
        want = np.array([oracle.lib.dsrt_oracle_powf(float(v), float(np.float32(y))) for v in base], np.float32)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), y


@pytest.mark.parametrize("name", sorted(CASES))
def test_render_matches_oracle_bit_for_bit(dsrt, gpu_ctx, oracle, name):
    hs, scene, W, H, spp, depth = _scene(dsrt, name)
    want_rgb, want_f32, want_cnt = oracle.render(scene, W, H)
    gpu_ctx.upload(scene)
    # counting build in reference-equivalent mode (no any-hit early-out): work counters must equal the oracle's exactly
    rgb, f32, st = gpu_ctx.render_to_host(dsrt.make_desc(W, H, spp, depth, collect_counters=2), want_f32=True)
    linf = float(np.abs(f32 - want_f32).max())
    assert linf <= LINF_TOLERANCE
    assert np.array_equal(rgb, want_rgb), f"{(rgb != want_rgb).any(axis=2).sum()} pixels differ, Linf={linf}"
    assert np.array_equal(f32.view(np.uint32), want_f32.view(np.uint32))
    for key in ("samples", "rays", "primary_hits", "box_fetches", "nodes_entered", "internal_entered", "tri_tests", "hit_updates",
                "sphere_tests", "shaded_hits", "tex_fetches", "max_stack"):
        assert getattr(st, key) == want_cnt[key], (key, getattr(st, key), want_cnt[key])
    # production build (unchecked, any-hit shadow rays): same bytes, never more work
    rgb2, f32b, _ = gpu_ctx.render_to_host(dsrt.make_desc(W, H, spp, depth), want_f32=True)
    assert np.array_equal(rgb2, want_rgb) and np.array_equal(f32b.view(np.uint32), want_f32.view(np.uint32))
    _, _, st1 = gpu_ctx.render_to_host(dsrt.make_desc(W, H, spp, depth, collect_counters=1))
    assert st1.rays == want_cnt["rays"] and st1.tri_tests <= want_cnt["tri_tests"] and st1.nodes_entered <= want_cnt["nodes_entered"]


@pytest.mark.parametrize("entries", [0, 8])
def test_short_stack_and_spill_give_identical_images(dsrt, gpu_ctx, oracle, entries):
    hs, scene, W, H, spp, depth = _scene(dsrt, "station_near")
    want_rgb, _, want_cnt = oracle.render(scene, W, H)
    gpu_ctx.upload(scene)
    rgb, _, st = gpu_ctx.render_to_host(dsrt.make_desc(W, H, spp, depth, collect_counters=1, stack_entries=entries))
    assert np.array_equal(rgb, want_rgb)
    assert st.lds_stack_entries == 8 and st.max_stack == want_cnt["max_stack"] and want_cnt["max_stack"] > 8
    assert st.stack_spills > 0              # this scene's walks go deeper than the LDS part: the global spill strip was really used


def test_pose_frame_config_c2_shape(dsrt, gpu_ctx, oracle, tmp_path):
    # BASELINE.json configs[1] at its own size and sample count: frame 0 of the pose file, 640x360 @ 64 spp (20k-triangle stand-in mesh;
    # the real ISS OBJ is not available).  Frame 0 is almost all background, so the oracle finishes in seconds;
    # frame 98 (fills the view) is checked at a reduced size on the same resident scene.
    from dsrt_amd import meshgen
    obj = tmp_path / "iss_20k.obj"
    meshgen.generate(obj, 20000)
    hs = dsrt.HostScene().add_obj(obj)
    hs.build_bvh()
    poses = dsrt.read_pose_file(os.path.join(GOLDEN, "rendezvous_1s_dt0_01s.txt"))
    uploaded = False
    for idx, (W, H, spp) in ((0, (640, 360, 64)), (98, (160, 90, 8))):
        fr = dsrt.pose_to_frame(poses[idx])
        cam = dsrt.frame_camera(fr, 40.0, W, H, spp, 50)
        scene = hs.view(cam, tuple(fr.sun_dir_model))
        if not uploaded:
            gpu_ctx.upload(scene)
            uploaded = True
        else:
            gpu_ctx.set_camera_sun(cam, tuple(fr.sun_dir_model))       # scene stays resident; only camera + sun change
        want_rgb, want_f32, cnt = oracle.render(scene, W, H)
        rgb, f32, _ = gpu_ctx.render_to_host(dsrt.make_desc(W, H, spp, 50), want_f32=True)
        assert cnt["primary_hits"] > 0
        assert np.array_equal(rgb, want_rgb) and np.array_equal(f32.view(np.uint32), want_f32.view(np.uint32)), idx


def test_tile_shards_reassemble_to_the_full_image(dsrt, gpu_ctx, oracle):
    import torch
    hs, scene, W, H, spp, depth = _scene(dsrt, "station_near")
    want_rgb, _, _ = oracle.render(scene, W, H)
    gpu_ctx.upload(scene)
    for world, tile in ((3, 8), (2, 16), (5, 8)):
        lay = dsrt.shard_layout(dsrt.make_desc(W, H, spp, depth, tile_size=tile, shard_count=world))
        gathered = torch.zeros(world * lay["rgb8_bytes_padded"], dtype=torch.uint8, device="cuda")
        for rank in range(world):
            d = dsrt.make_desc(W, H, spp, depth, tile_size=tile, shard_rank=rank, shard_count=world)
            part = gathered[rank * lay["rgb8_bytes_padded"]:(rank + 1) * lay["rgb8_bytes_padded"]]
            gpu_ctx.render(d, part.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
        image = torch.zeros(H * W * 3, dtype=torch.uint8, device="cuda")
        gpu_ctx.deinterleave(dsrt.make_desc(W, H, spp, depth, tile_size=tile, shard_count=world), gathered.data_ptr(), image.data_ptr(),
                             stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert np.array_equal(image.cpu().numpy().reshape(H, W, 3), want_rgb), (world, tile)
        # the host-side mirror of the layout (used by the gloo tests and for debugging) agrees with the kernels
        from dsrt_amd import dist as shard
        assert np.array_equal(shard.deinterleave_host(gathered.cpu().numpy(), W, H, world, tile), want_rgb)


def test_ragged_image_sizes(dsrt, gpu_ctx, oracle):
    # width/height not multiples of the 8x8 work item: edge items are clipped, nothing is written out of bounds
    hs = load_world(dsrt, "lights")
    gpu_ctx_uploaded = False
    for W, H in ((37, 21), (8, 2), (2, 2), (65, 9)):
        cam = dsrt.camera_look_at((0.0, 3.0, 9.0), (0.0, 2.0, 0.0), 45.0, W, H, 4, 12)
        scene = hs.view(cam, SUN)
        if not gpu_ctx_uploaded:
            gpu_ctx.upload(scene)
            gpu_ctx_uploaded = True
        else:
            gpu_ctx.set_camera_sun(cam, SUN)
        want_rgb, _, _ = oracle.render(scene, W, H)
        rgb, _, _ = gpu_ctx.render_to_host(dsrt.make_desc(W, H, 4, 12, checked=1))
        assert np.array_equal(rgb, want_rgb), (W, H)


def test_empty_scene_and_argument_errors(dsrt, gpu_ctx):
    hs = dsrt.HostScene()
    hs.build_bvh()
    cam = dsrt.camera_look_at((0, 0, 5), (0, 0, 0), 40.0, 32, 16, 2, 5)
    gpu_ctx.upload(hs.view(cam, SUN))
    rgb, _, st = gpu_ctx.render_to_host(dsrt.make_desc(32, 16, 2, 5, collect_counters=1))
    assert not rgb.any() and st.samples == 32 * 16 * 2 and st.rays == st.samples and st.box_fetches == 0
    with pytest.raises(dsrt.DsrtError):
        gpu_ctx.render_to_host(dsrt.make_desc(1, 16, 2, 5))
    with pytest.raises(dsrt.DsrtError):
        gpu_ctx.render_to_host(dsrt.make_desc(32, 16, 2, 5, tile_size=12))
    with pytest.raises(dsrt.DsrtError):
        gpu_ctx.render_to_host(dsrt.make_desc(32, 16, 2, 5, stack_entries=12))
    fresh = dsrt.Context(0)
    with pytest.raises(dsrt.DsrtError):
        fresh.render_to_host(dsrt.make_desc(32, 16, 2, 5))      # no scene uploaded
    fresh.close()


def test_drop_in_entry_points(dsrt, oracle, tmp_path):
    # build_gpu_scene -> gpu_render_scene -> free_gpu_scene, the three calls of src/main.cpp:405-428, C forms.
    hs, scene, W, H, spp, depth = _scene(dsrt, "mixed")
    want_rgb, _, _ = oracle.render(scene, W, H)
    dev = dsrt.GPUScene()
    sun = (C.c_float * 3)(*SUN)
    rc = dsrt.lib.dsrt_build_gpu_scene(hs._h, C.byref(scene.camera), sun, C.byref(dev))
    assert rc == 0, dsrt.lib.dsrt_last_error()
    assert dev.num_triangles == scene.num_triangles and dev.seed == 1337 and dev.params.gamma == 2.0
    assert dev.triangles and dev.triangles != scene.triangles          # device copies in the reference layouts
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        dsrt.lib.gpu_render_scene(C.byref(dev), W, H)
        data = open("image_gpu.ppm", "rb").read()
    finally:
        os.chdir(cwd)
    header = f"P6\n{W} {H}\n255\n".encode()
    assert data.startswith(header) and data[len(header):] == want_rgb.tobytes()
    # the reference calls this per frame with the same geometry: the second call finds the converted scene of the first (content
    # fingerprint) and must give the same file; a camera change in the header must still be honoured ...
    os.chdir(tmp_path)
    try:
        os.remove("image_gpu.ppm")
        dsrt.lib.gpu_render_scene(C.byref(dev), W, H)
        again = open("image_gpu.ppm", "rb").read()
        cam2 = dsrt.camera_look_at((9.0, 7.0, 24.0), (0.0, 1.0, 0.0), 40.0, W, H, spp, depth)
        dev.camera = cam2
        dev.params.samples_per_pixel = spp
        dsrt.lib.gpu_render_scene(C.byref(dev), W, H)
        moved = open("image_gpu.ppm", "rb").read()
    finally:
        os.chdir(cwd)
    assert again == data
    want2, _, _ = oracle.render(hs.view(cam2, SUN), W, H)
    assert moved[len(header):] == want2.tobytes()
    dsrt.lib.dsrt_free_gpu_scene(C.byref(dev))
    assert not dev.triangles and dev.num_triangles == 0 and not dev.bvh_nodes
    # ... and different geometry behind a fresh header must not be mistaken for the cached scene
    hs3, scene3, W3, H3, spp3, depth3 = _scene(dsrt, "lights")
    want3, _, _ = oracle.render(scene3, W3, H3)
    dev3 = dsrt.GPUScene()
    assert dsrt.lib.dsrt_build_gpu_scene(hs3._h, C.byref(scene3.camera), sun, C.byref(dev3)) == 0
    os.chdir(tmp_path)
    try:
        dsrt.lib.gpu_render_scene(C.byref(dev3), W3, H3)
        third = open("image_gpu.ppm", "rb").read()
    finally:
        os.chdir(cwd)
    dsrt.lib.dsrt_free_gpu_scene(C.byref(dev3))
    assert third[len(f"P6\n{W3} {H3}\n255\n".encode()):] == want3.tobytes()


def test_full_size_frame_properties_and_oracle_rows(dsrt, gpu_ctx, oracle, tmp_path):
    """BASELINE.json's metric size (1920x1080 @ 1000 spp, pose frame 98) on a 100k-triangle stand-in mesh.

    The oracle cannot render 2 G samples in test time, so the whole frame is checked through properties that do not depend
    on size -- idempotence (same bytes twice), independence from the order pixels are handed out in (costliest-first vs
    natural), tile shards reassembling to the same frame -- and three full-width rows (1920 px x 1000 spp each) are checked
    bit for bit against the oracle."""
    import threading
    import torch
    from dsrt_amd import meshgen
    obj = tmp_path / "iss_100k.obj"
    meshgen.generate(obj, 100000)
    hs = dsrt.HostScene().add_obj(obj)
    hs.build_bvh()
    poses = dsrt.read_pose_file(os.path.join(GOLDEN, "rendezvous_1s_dt0_01s.txt"))
    W, H, spp = 1920, 1080, 1000
    fr = dsrt.pose_to_frame(poses[98])
    cam = dsrt.frame_camera(fr, 40.0, W, H, spp, 50)
    scene = hs.view(cam, tuple(fr.sun_dir_model))
    gpu_ctx.upload(scene)
    first, f32, st = gpu_ctx.render_to_host(dsrt.make_desc(W, H, spp, 50), want_f32=True)
    again, _, _ = gpu_ctx.render_to_host(dsrt.make_desc(W, H, spp, 50))
    natural, _, _ = gpu_ctx.render_to_host(dsrt.make_desc(W, H, spp, 50, tune=(0, 0, 0, 1)))
    assert np.array_equal(first, again) and np.array_equal(first, natural)
    assert (first.max(axis=2) > 0).mean() > 0.2            # the station fills a good part of this frame

    world = 4
    lay = dsrt.shard_layout(dsrt.make_desc(W, H, spp, 50, shard_count=world))
    gathered = torch.zeros(world * lay["rgb8_bytes_padded"], dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    for rank in range(world):
        part = gathered[rank * lay["rgb8_bytes_padded"]:(rank + 1) * lay["rgb8_bytes_padded"]]
        gpu_ctx.render(dsrt.make_desc(W, H, spp, 50, shard_rank=rank, shard_count=world), part.data_ptr(), stream=stream)
    image = torch.zeros(H * W * 3, dtype=torch.uint8, device="cuda")
    gpu_ctx.deinterleave(dsrt.make_desc(W, H, spp, 50, shard_count=world), gathered.data_ptr(), image.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    assert np.array_equal(image.cpu().numpy().reshape(H, W, 3), first)

    rows = [540, 97, 1003]                                   # kernel rows (0 = bottom): centre, low, high
    want = {}

    def run(y):
        want[y] = oracle.render(scene, W, H, y, y + 1)
    threads = [threading.Thread(target=run, args=(y,)) for y in rows]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for y in rows:
        rgb, ref32, _ = want[y]
        r = H - 1 - y
        assert np.array_equal(first[r], rgb[r]), f"row {y}: {(first[r] != rgb[r]).any(axis=1).sum()} pixels differ"
        assert np.array_equal(f32[r].view(np.uint32), ref32[r].view(np.uint32))


def test_philox_stream_equals_rocrand(gpu_ctx):
    # rng_mode 1 draws from rocRAND's Philox4x32-10: our stateless form vs rocrand_init + rocrand() on the device
    for seed, sub in ((1337, 0), (1337, 123456789012), (0xDEADBEEFDEADBEEF, 2**40 + 5), (0, 2**63)):
        ours, theirs = gpu_ctx.selftest_philox(seed, sub, 64)
        assert np.array_equal(ours, theirs), (seed, sub)
    a, _ = gpu_ctx.selftest_philox(1337, 1, 8)
    b, _ = gpu_ctx.selftest_philox(1337, 2, 8)
    assert not np.array_equal(a, b)


@pytest.mark.parametrize("name", ["station_near", "lights", "c1_spheres"])
def test_rng_mode1_is_deterministic_shard_invariant_and_statistically_mode0(dsrt, gpu_ctx, oracle, name):
    """rng_mode 1 (one Philox sub-sequence per (pixel, sample)) is NOT bit-identical to the reference's LCG mode by design;
    it must be reproducible, independent of sharding / hand-out order, and the same image up to Monte-Carlo noise."""
    import torch
    hs, scene, W, H, _, depth = _scene(dsrt, name)
    spp = 256
    gpu_ctx.upload(scene)
    m0, f0, _ = gpu_ctx.render_to_host(dsrt.make_desc(W, H, spp, depth), want_f32=True)
    m1, f1, st = gpu_ctx.render_to_host(dsrt.make_desc(W, H, spp, depth, rng_mode=1, collect_counters=1), want_f32=True)
    again, _, _ = gpu_ctx.render_to_host(dsrt.make_desc(W, H, spp, depth, rng_mode=1, tune=(0, 0, 0, 1)))
    assert np.array_equal(m1, again)                                   # same bytes, whatever the order of work
    assert st.samples == W * H * spp
    assert not np.array_equal(m0, m1)
    d = f1.astype(np.float64) - f0.astype(np.float64)
    assert abs(d.mean()) < 2e-3                                        # no bias ...
    assert np.abs(d).mean() < 0.03                                     # ... and pixel differences at the noise level of 256 spp
    # shards
    world = 3
    lay = dsrt.shard_layout(dsrt.make_desc(W, H, spp, depth, shard_count=world))
    gathered = torch.zeros(world * lay["rgb8_bytes_padded"], dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    for rank in range(world):
        part = gathered[rank * lay["rgb8_bytes_padded"]:(rank + 1) * lay["rgb8_bytes_padded"]]
        gpu_ctx.render(dsrt.make_desc(W, H, spp, depth, shard_rank=rank, shard_count=world, rng_mode=1), part.data_ptr(), stream=stream)
    image = torch.zeros(H * W * 3, dtype=torch.uint8, device="cuda")
    gpu_ctx.deinterleave(dsrt.make_desc(W, H, spp, depth, shard_count=world), gathered.data_ptr(), image.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    assert np.array_equal(image.cpu().numpy().reshape(H, W, 3), m1)


def test_empty_tile_culling_is_exact(dsrt, gpu_ctx, oracle):
    """Tiles whose sample footprint (grown by a pixel) cannot reach the root box are skipped and left black.  The frame must be
    the oracle's, byte for byte, with and without the culling; and the culling must actually remove tiles on a far view,
    none when the camera sits inside the root box, and none when the scene has spheres."""
    import torch
    hs = load_world(dsrt, "station_3k")
    uploaded = False
    views = [((-0.7, 0.0, 260.0), (0.0, 0.0, 0.0), 200, 112, True),       # far: a small station in a black frame
             ((60.0, 45.0, 120.0), (30.0, 10.0, 0.0), 157, 83, True),     # off-centre, ragged size
             ((0.0, 0.0, 300.0), (250.0, 0.0, 0.0), 96, 64, True),        # station almost out of view at the image edge
             ((0.5, 0.2, 0.3), (10.0, 0.0, 0.0), 96, 64, False)]          # camera inside the root box: nothing can be culled
    for lookfrom, lookat, W, H, expect_culled in views:
        cam = dsrt.camera_look_at(lookfrom, lookat, 40.0, W, H, 8, 12)
        scene = hs.view(cam, SUN)
        if not uploaded:
            gpu_ctx.upload(scene)
            uploaded = True
        else:
            gpu_ctx.set_camera_sun(cam, SUN)
        want_rgb, want_f32, _ = oracle.render(scene, W, H)
        rgb, f32, st = gpu_ctx.render_to_host(dsrt.make_desc(W, H, 8, 12), want_f32=True)
        rgb_all, f32_all, st_all = gpu_ctx.render_to_host(dsrt.make_desc(W, H, 8, 12, tune=(0, 0, 0, 2)), want_f32=True)
        assert st_all.tiles_culled == 0 and st.tiles_total == st_all.tiles_total == ((W + 7) // 8) * ((H + 7) // 8)
        assert (st.tiles_culled > 0) == expect_culled, (lookfrom, st.tiles_culled)
        assert np.array_equal(rgb, want_rgb) and np.array_equal(rgb_all, want_rgb), lookfrom
        assert np.array_equal(f32.view(np.uint32), want_f32.view(np.uint32)) and np.array_equal(f32_all.view(np.uint32), want_f32.view(np.uint32))
        # rng_mode 1 goes through the same pre-pass: culled or not, same bytes
        m1, _, s1 = gpu_ctx.render_to_host(dsrt.make_desc(W, H, 8, 12, rng_mode=1))
        m1_all, _, _ = gpu_ctx.render_to_host(dsrt.make_desc(W, H, 8, 12, rng_mode=1, tune=(0, 0, 0, 2)))
        assert np.array_equal(m1, m1_all) and s1.tiles_culled == st.tiles_culled
        # sharded, compact output
        world, tile = 3, 8
        lay = dsrt.shard_layout(dsrt.make_desc(W, H, 8, 12, tile_size=tile, shard_count=world))
        gathered = torch.full((world * lay["rgb8_bytes_padded"],), 77, dtype=torch.uint8, device="cuda")
        for rank in range(world):
            part = gathered[rank * lay["rgb8_bytes_padded"]:(rank + 1) * lay["rgb8_bytes_padded"]]
            gpu_ctx.render(dsrt.make_desc(W, H, 8, 12, tile_size=tile, shard_rank=rank, shard_count=world), part.data_ptr(),
                           stream=torch.cuda.current_stream().cuda_stream)
        image = torch.zeros(H * W * 3, dtype=torch.uint8, device="cuda")
        gpu_ctx.deinterleave(dsrt.make_desc(W, H, 8, 12, tile_size=tile, shard_count=world), gathered.data_ptr(), image.data_ptr(),
                             stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert np.array_equal(image.cpu().numpy().reshape(H, W, 3), want_rgb), lookfrom
    # spheres are tested outside the BVH: a scene that has any is never culled
    hs2 = load_world(dsrt, "mixed")
    cam = dsrt.camera_look_at((3.0, 6.0, 140.0), (0.0, 2.0, 0.0), 40.0, 96, 64, 4, 10)
    scene = hs2.view(cam, SUN)
    gpu_ctx.upload(scene)
    want_rgb, _, _ = oracle.render(scene, 96, 64)
    rgb, _, st = gpu_ctx.render_to_host(dsrt.make_desc(96, 64, 4, 10))
    assert st.tiles_culled == 0 and np.array_equal(rgb, want_rgb)


def test_scheduling_switches_never_change_a_pixel(dsrt, gpu_ctx, oracle):
    """Probe-refined tile order (active from 256 spp), shadow-ray helpers, culling, natural order: every combination gives the
    oracle's bytes.  256 spp on a small image: most lanes are idle, so nearly every shadow ray is traced by a helper lane."""
    hs, scene, W, H, _, depth = _scene(dsrt, "station_near")
    W, H, spp = 96, 54, 256
    cam = dsrt.camera_look_at((12.0, 9.0, 38.0), (0.0, 0.0, 0.0), 40.0, W, H, spp, depth)
    scene = hs.view(cam, SUN)
    want_rgb, want_f32, _ = oracle.render(scene, W, H)
    gpu_ctx.upload(scene)
    for flags in (0, 1, 2, 4, 8, 12, 5, 14, 32, 36, 63 - 1, 64):        # include/dsrt.h, DSRT_TUNE_*
        rgb, f32, _ = gpu_ctx.render_to_host(dsrt.make_desc(W, H, spp, depth, tune=(0, 0, 0, flags)), want_f32=True)
        assert np.array_equal(rgb, want_rgb) and np.array_equal(f32.view(np.uint32), want_f32.view(np.uint32)), flags
    # bits the ABI does not define are refused, not ignored
    for flags in (128, 1 << 23, -(1 << 31)):
        with pytest.raises(dsrt.DsrtError):
            gpu_ctx.render_to_host(dsrt.make_desc(W, H, spp, depth, tune=(0, 0, 0, flags)))
    # the development switches (dsrt_dev_set_experiment) are scheduling only as well; undefined bits are refused
    try:
        for xp in (64, 1 << 20):
            dsrt.set_experiment(xp)
            rgb, f32, _ = gpu_ctx.render_to_host(dsrt.make_desc(W, H, spp, depth), want_f32=True)
            assert np.array_equal(rgb, want_rgb) and np.array_equal(f32.view(np.uint32), want_f32.view(np.uint32)), xp
    finally:
        dsrt.set_experiment(0)
    with pytest.raises(dsrt.DsrtError):
        dsrt.set_experiment(1 << 5)
    # rng_mode 1 sums samples as integers, so neither the scheduling switches nor sample stealing (+16 switches it off) may move a bit
    a, _, _ = gpu_ctx.render_to_host(dsrt.make_desc(W, H, spp, depth, rng_mode=1))
    for flags in (12, 16, 28, 1 + 16, 2):
        b, _, _ = gpu_ctx.render_to_host(dsrt.make_desc(W, H, spp, depth, rng_mode=1, tune=(0, 0, 0, flags)))
        assert np.array_equal(a, b), flags
    try:
        for xp in (1 << 31, 16 << 8, 1 << 28):                      # background pixels one item each; 16 slices per heavy pixel; 64-sample light items
            dsrt.set_experiment(xp)
            b, _, _ = gpu_ctx.render_to_host(dsrt.make_desc(W, H, spp, depth, rng_mode=1))
            assert np.array_equal(a, b), xp
    finally:
        dsrt.set_experiment(0)


def test_cli_renders_pose_frames_like_the_library(dsrt, oracle, tmp_path):
    """deep-space-ray-tracer_amd/dsrt_render (the reference's main.cpp flow: pose file -> frames in --output_dir) end to end:
    two frames of the committed pose file on the small station; frame 98's PPM equals the oracle's image, its PNG twin decodes
    to the same bytes, and the non-parity --fast run produces a frame of the right shape."""
    import subprocess
    import zlib
    import struct
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "deep-space-ray-tracer_amd", "dsrt_render")
    assert os.path.exists(exe), "build the CLI with `make tools`"
    obj = os.path.join(ASSETS, "station_3k.obj")
    poses_txt = os.path.join(GOLDEN, "rendezvous_1s_dt0_01s.txt")
    W, H, spp = 160, 90, 8
    common = [exe, "--obj", obj, "--input_txt", poses_txt, "--width", str(W), "--height", str(H), "--spp", str(spp), "--frame", "97", "--frames", "2"]
    out = tmp_path / "frames"
    r = subprocess.run(common + ["--output_dir", str(out)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Loaded 99 poses." in r.stdout and "=== Frame 98 ===" in r.stdout
    raw = (out / "frame_0098.ppm").read_bytes()
    head = f"P6\n{W} {H}\n255\n".encode()
    assert raw.startswith(head)
    got = np.frombuffer(raw[len(head):], np.uint8).reshape(H, W, 3)
    # the same frame through the library + oracle
    hs = dsrt.HostScene().add_obj(obj)
    hs.build_bvh()
    poses = dsrt.read_pose_file(poses_txt)
    fr = dsrt.pose_to_frame(poses[98])
    cam = dsrt.frame_camera(fr, 40.0, W, H, spp, 50)
    want, _, _ = oracle.render(hs.view(cam, tuple(fr.sun_dir_model)), W, H)
    assert np.array_equal(got, want)
    # PNG output
    out2 = tmp_path / "png"
    r = subprocess.run(common + ["--output_dir", str(out2), "--png"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    png = (out2 / "frame_0098.png").read_bytes()
    pos, idat = 8, b""
    while pos < len(png):
        n, kind = struct.unpack(">I4s", png[pos:pos + 8])
        if kind == b"IDAT":
            idat += png[pos + 8:pos + 8 + n]
        pos += 12 + n
    rows = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(H, 1 + W * 3)
    assert np.array_equal(rows[:, 1:].reshape(H, W, 3), want)
    # --certified-tree: the same bytes as the plain run
    out4 = tmp_path / "certified"
    r = subprocess.run(common + ["--output_dir", str(out4), "--certified-tree"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    for f in sorted(os.listdir(out)):
        assert open(out4 / f, "rb").read() == open(out / f, "rb").read(), f
    # fast mode: runs, right size, statistically the same picture
    out3 = tmp_path / "fast"
    r = subprocess.run(common + ["--output_dir", str(out3), "--fast"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    fast = np.frombuffer((out3 / "frame_0098.ppm").read_bytes()[len(head):], np.uint8).reshape(H, W, 3)
    assert abs(fast.astype(float).mean() - want.astype(float).mean()) < 3.0
    # the tree built on the GPU: same flow, a picture of the same level
    out4 = tmp_path / "lbvh"
    r = subprocess.run(common + ["--output_dir", str(out4), "--bvh", "lbvh"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "BVH built on the GPU" in r.stdout, r.stdout + r.stderr
    gpu_tree = np.frombuffer((out4 / "frame_0098.ppm").read_bytes()[len(head):], np.uint8).reshape(H, W, 3)
    assert abs(gpu_tree.astype(float).mean() - want.astype(float).mean()) < 3.0
    # usage errors
    assert subprocess.run([exe], capture_output=True).returncode == 2


def test_randomised_cameras_sizes_and_switches_match_the_oracle(dsrt, gpu_ctx, oracle):
    """A bounded fuzz over what the scheduling layers see: cameras near, far, inside and beside the station, ragged sizes, sample
    counts on both sides of the probe threshold, depths, tile sizes, and random combinations of the scheduling switches.  The
    expected image is always the oracle's, byte for byte (rng_mode 0), and rng_mode 1 must not depend on the switches."""
    rng = np.random.default_rng(20251004)
    worlds = {name: load_world(dsrt, name) for name in ("station_3k", "mixed", "textured")}
    failures = []
    for trial in range(36):
        name = ("station_3k", "station_3k", "mixed", "textured")[trial % 4]
        hs = worlds[name]
        W, H = int(rng.integers(9, 140)), int(rng.integers(5, 90))
        spp = int(rng.choice([1, 3, 8, 32, 128, 160]))
        if W * H * spp > 400000:
            spp = max(1, 400000 // (W * H))
        depth = int(rng.choice([1, 2, 5, 12, 50]))
        dist = float(rng.choice([0.3, 2.0, 15.0, 45.0, 200.0, 900.0]))
        direction = rng.normal(size=3)
        direction /= np.linalg.norm(direction)
        lookfrom = tuple(float(v) for v in direction * dist)
        lookat = tuple(float(v) for v in rng.normal(size=3) * (0.0 if trial % 3 else dist * 0.4))
        cam = dsrt.camera_look_at(lookfrom, lookat, float(rng.choice([20.0, 40.0, 75.0])), W, H, spp, depth)
        sun = tuple(float(v) for v in rng.normal(size=3))
        scene = hs.view(cam, sun)
        want, want_f32, _ = oracle.render(scene, W, H)
        gpu_ctx.upload(scene)
        flags = int(rng.choice([0, 0, 2, 4, 8, 12, 6, 1]))
        tile = int(rng.choice([0, 0, 16]))
        rgb, f32, _ = gpu_ctx.render_to_host(dsrt.make_desc(W, H, spp, depth, tile_size=tile, tune=(0, 0, 0, flags)), want_f32=True)
        if not (np.array_equal(rgb, want) and np.array_equal(f32.view(np.uint32), want_f32.view(np.uint32))):
            failures.append((trial, name, W, H, spp, depth, lookfrom, flags, tile, int((rgb != want).any(axis=2).sum())))
        if trial % 6 == 0:
            a, _, _ = gpu_ctx.render_to_host(dsrt.make_desc(W, H, spp, depth, rng_mode=1))
            b, _, _ = gpu_ctx.render_to_host(dsrt.make_desc(W, H, spp, depth, rng_mode=1, tune=(0, 0, 0, 14 + 16 * (trial % 12 == 0))))
            if not np.array_equal(a, b):
                failures.append((trial, "rng_mode 1 depends on scheduling switches"))
    assert not failures, failures


def test_bench_two_rank_flow_rehearsed_on_one_gpu(tmp_path):
    """bench.py's N > 1 path (tile shards, equal-sized compact buffers, one gather, de-interleave on rank 0, max-over-ranks
    timing, JSON line) with two ranks sharing GPU 0 and gloo carrying host copies (DSRT_BENCH_REHEARSAL): rank 0 must
    reassemble exactly the image a whole-frame render gives."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DSRT_BENCH_REHEARSAL="1", MASTER_ADDR="127.0.0.1")
    # `python bench.py --gpus 2` from a plain interpreter -- the way the driver calls it for N = 1 -- must start the two-rank job itself
    # (one process per rank through torch.distributed.run) and hand back its exit code and its JSON line, never run one GPU silently
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--tris", "20000", "--width", "322", "--height", "190",
           "--spp", "16", "--no-cpu"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["unit"] == "Msamples/s"
    assert out["rehearsal"]["reassembled_image_equals_whole_frame_render"] is True
    assert out["value"] > 0 and out["roofline"]["achieved_algorithmic_GBps"] > 0 and out["roofline"]["kernel_ms"] > 0
    assert out["rng_mode_1_same_sharding"]["value"] > 0              # both generators are reported for the sharded step
    # what scales, readable without any other document: the three forms against their own one-GPU times measured in the same run
    sd = out["scaling_detail"]
    assert sd["n_gpus"] == 2 and set(sd["speedup_vs_1gpu"]) == {"single_launch_rng_mode_0", "single_launch_rng_mode_1", "batch_launch_rng_mode_0"}
    assert all(v and v > 0 for v in sd["speedup_vs_1gpu"].values()), sd


def test_bench_line_ties_the_headline_frame_to_the_oracle(tmp_path):
    """`python bench.py` at N = 1 (a small frame here): the oracle rows rendered for the cpu_baseline leg are compared with the same rows of the
    image of the timed GPU step, and the line says so -- `parity_rows.mismatched == 0`."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0", "--tris", "20000", "--width", "322", "--height", "190", "--spp", "16",
           "--no-extras", "--no-pmc", "--cpu-budget", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    pr = out["parity_rows"]
    assert pr["rows"] >= 8 and pr["pixels"] == pr["rows"] * 322 and pr["mismatched"] == 0 and pr["lit_pixels"] > 100, pr
    assert out["cpu_baseline"]["kind"] == "port" and out["cpu_baseline"]["value"] > 0


def test_headline_mesh_rows_match_the_oracle(dsrt, gpu_ctx, oracle):
    """The bench's own workload -- the 1,000,000-triangle stand-in, pose frame 98, 1920x1080, max_depth 50 -- at 48 samples: twelve rows spread
    over the image against the oracle, rgb8 and fp32 bit patterns.  The tree of this mesh needs 18 stack entries: 10 spill levels behind the
    8-entry LDS stack, which the small test meshes never reach."""
    import threading
    from dsrt_amd import meshgen
    obj = f"/tmp/dsrt_bench_station_v{meshgen.VERSION}_1000000.obj"
    if not os.path.exists(obj):
        tmp = obj + f".{os.getpid()}.tmp"
        meshgen.write_obj(meshgen.build_station(1000000), tmp, mtl_name=os.path.basename(obj)[:-4] + ".mtl")
        os.replace(tmp, obj)
    hs = dsrt.HostScene().add_obj(obj)
    hs.build_bvh()
    assert hs.stack_need > 8
    poses = dsrt.read_pose_file(os.path.join(GOLDEN, "rendezvous_1s_dt0_01s.txt"))
    W, H, spp = 1920, 1080, 48
    fr = dsrt.pose_to_frame(poses[98])
    cam = dsrt.frame_camera(fr, 40.0, W, H, spp, 50)
    scene = hs.view(cam, tuple(fr.sun_dir_model))
    scene.params.samples_per_pixel = spp
    gpu_ctx.upload(scene)
    rgb, f32, st = gpu_ctx.render_to_host(dsrt.make_desc(W, H, spp, 50), want_f32=True)
    rows = [45 + 90 * k for k in range(12)]
    want = np.zeros((H, W, 3), np.uint8)
    want32 = np.zeros((H, W, 3), np.float32)
    cnt = [(C.c_uint64 * len(oracle.COUNTER_NAMES))() for _ in rows]
    threads = [threading.Thread(target=oracle.lib.dsrt_oracle_render_rows, args=(C.byref(scene), W, H, y, y + 1, want.ctypes.data, want32.ctypes.data, cnt[i]))
               for i, y in enumerate(rows)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    img_rows = [H - 1 - y for y in rows]
    assert (want[img_rows].max(axis=2) > 0).sum() > 2000
    assert np.array_equal(rgb[img_rows], want[img_rows]), f"{(rgb[img_rows] != want[img_rows]).any(axis=2).sum()} pixels differ"
    assert np.array_equal(f32[img_rows].view(np.uint32), want32[img_rows].view(np.uint32))
    # the counting build on this tree: the LDS stack's spill path is taken, and its maximum depth is the oracle's
    _, _, sc = gpu_ctx.render_to_host(dsrt.make_desc(W, H, 2, 50, collect_counters=2))
    assert sc.stack_spills > 0 and sc.max_stack > 8
    # and EVERY pixel of the frame at 4 samples (16 row bands in parallel; the C call releases the GIL), counters included
    spp = 4
    cam = dsrt.frame_camera(fr, 40.0, W, H, spp, 50)
    scene = hs.view(cam, tuple(fr.sun_dir_model))
    scene.params.samples_per_pixel = spp
    gpu_ctx.set_camera_sun(cam, tuple(fr.sun_dir_model))
    rgb, f32, st = gpu_ctx.render_to_host(dsrt.make_desc(W, H, spp, 50, collect_counters=2), want_f32=True)
    want = np.zeros((H, W, 3), np.uint8)
    want32 = np.zeros((H, W, 3), np.float32)
    bands = [(H * k // 16, H * (k + 1) // 16) for k in range(16)]
    cnt = [(C.c_uint64 * len(oracle.COUNTER_NAMES))() for _ in bands]
    threads = [threading.Thread(target=oracle.lib.dsrt_oracle_render_rows, args=(C.byref(scene), W, H, y0, y1, want.ctypes.data, want32.ctypes.data, cnt[i]))
               for i, (y0, y1) in enumerate(bands)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert np.array_equal(rgb, want), f"{(rgb != want).any(axis=2).sum()} pixels differ"
    assert np.array_equal(f32.view(np.uint32), want32.view(np.uint32))
    tot = {n: sum(int(c[i]) for c in cnt) for i, n in enumerate(oracle.COUNTER_NAMES)}
    for name in ("samples", "rays", "nodes_entered", "internal_entered", "tri_tests", "hit_updates", "box_fetches", "shaded_hits"):
        assert getattr(st, name) == tot[name], (name, getattr(st, name), tot[name])
    assert st.max_stack == max(int(c[oracle.COUNTER_NAMES.index("max_stack")]) for c in cnt)
    rgb2, _, _ = gpu_ctx.render_to_host(dsrt.make_desc(W, H, spp, 50))                    # and the production build
    assert np.array_equal(rgb2, want)


def test_whole_1080p_frames_match_the_oracle(dsrt, gpu_ctx, oracle, tmp_path):
    """Every pixel of two production-size frames against the oracle (threads over row bands; the C call releases the GIL):
    the near frame at 72 samples (probe-refined order, both queues, helpers in the tail) and a far frame at 16 (most tiles
    culled, heavy items spread one or two per wave, nearly every shadow ray traced by an idle lane)."""
    import threading
    from dsrt_amd import meshgen
    obj = tmp_path / "iss_60k.obj"
    meshgen.generate(obj, 60000)
    hs = dsrt.HostScene().add_obj(obj)
    hs.build_bvh()
    poses = dsrt.read_pose_file(os.path.join(GOLDEN, "rendezvous_1s_dt0_01s.txt"))
    W, H = 1920, 1080
    uploaded = False
    for frame, spp in ((98, 72), (70, 16)):
        fr = dsrt.pose_to_frame(poses[frame])
        cam = dsrt.frame_camera(fr, 40.0, W, H, spp, 50)
        scene = hs.view(cam, tuple(fr.sun_dir_model))
        if uploaded:
            gpu_ctx.set_camera_sun(cam, tuple(fr.sun_dir_model))
        else:
            gpu_ctx.upload(scene)
            uploaded = True
        rgb, f32, st = gpu_ctx.render_to_host(dsrt.make_desc(W, H, spp, 50), want_f32=True)
        want = np.zeros((H, W, 3), np.uint8)
        want32 = np.zeros((H, W, 3), np.float32)
        n_threads = min(16, len(os.sched_getaffinity(0)))
        edges = np.linspace(0, H, n_threads * 6 + 1).astype(int)
        bands = list(zip(edges[:-1], edges[1:]))
        lock = threading.Lock()

        def worker():
            while True:
                with lock:
                    if not bands:
                        return
                    y0, y1 = bands.pop()
                part, part32, _ = oracle.render(scene, W, H, int(y0), int(y1))
                r0, r1 = H - int(y1), H - int(y0)                    # kernel row y is image row H-1-y
                want[r0:r1] = part[r0:r1]
                want32[r0:r1] = part32[r0:r1]
        threads = [threading.Thread(target=worker) for _ in range(n_threads)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert np.array_equal(rgb, want), f"frame {frame}: {(rgb != want).any(axis=2).sum()} pixels differ"
        assert np.array_equal(f32.view(np.uint32), want32.view(np.uint32)), frame
        if frame == 70:
            assert st.tiles_culled > 0.8 * st.tiles_total
        else:
            # the priority switch only exists on a chip full of heavy work, i.e. at this size: scheduling only, so the same bits
            for flags in (32,):
                again, again32, _ = gpu_ctx.render_to_host(dsrt.make_desc(W, H, spp, 50, tune=(0, 0, 0, flags)), want_f32=True)
                assert np.array_equal(again, want) and np.array_equal(again32.view(np.uint32), want32.view(np.uint32)), flags


def _station_scene(dsrt, tmp_path, tris):
    from dsrt_amd import meshgen
    obj = tmp_path / f"iss_{tris}.obj"
    if not obj.exists():
        meshgen.generate(obj, tris)
    hs = dsrt.HostScene().add_obj(obj)
    hs.build_bvh()
    poses = dsrt.read_pose_file(os.path.join(GOLDEN, "rendezvous_1s_dt0_01s.txt"))
    return obj, hs, poses


def test_sequence_flow_frames_in_flight_match_the_oracle(dsrt, gpu_ctx, oracle, tmp_path):
    """BASELINE.json configs[4], the flow of `bench.py --sequence` (deep-space-ray-tracer_amd/sequence.py; the reference's frame loop is
    src/main.cpp:310-431): the scene resident, 4 frames in flight on separate streams and contexts that SHARE the scene, images
    copied to pinned host memory -- over 7 poses including the far end (0), the middle (70) and the near end (98), every frame
    compared byte for byte with the oracle.  Then the same frames as batch launches (sequence.render_batches) and through the library's
    one-process path (dsrt_multi_render_sequence: three ranks, here all on GPU 0, frames dealt round-robin, each rank's as batch launches)."""
    from dsrt_amd import sequence
    _, hs, poses = _station_scene(dsrt, tmp_path, 20000)
    W, H, spp, depth = 320, 180, 16, 50
    frames = [0, 30, 70, 90, 96, 97, 98]

    def frame(i):
        fr = dsrt.pose_to_frame(poses[i])
        return dsrt.frame_camera(fr, 40.0, W, H, spp, depth), tuple(fr.sun_dir_model)

    want = {}
    for i in frames:
        cam, sun = frame(i)
        want[i] = oracle.render(hs.view(cam, sun), W, H, want_f32=False)[0]
    assert want[98].max() > 0 and want[0].max() > 0
    cam0, sun0 = frame(frames[0])
    gpu_ctx.upload(hs.view(cam0, sun0))
    for rng_mode in (0, 1):
        got = sequence.render_frames(dsrt, gpu_ctx, frame, frames, W, H, spp, depth, inflight=4, rng_mode=rng_mode)
        assert sorted(got) == frames
        if rng_mode == 0:
            for i in frames:
                assert np.array_equal(got[i], want[i]), f"frame {i}: {(got[i] != want[i]).any(axis=2).sum()} pixels differ"
        else:                                             # mode 1: not the reference stream, but the same frame alone and in flight
            for i in (0, 98):
                cam, sun = frame(i)
                gpu_ctx.set_camera_sun(cam, sun)
                alone, _, _ = gpu_ctx.render_to_host(dsrt.make_desc(W, H, spp, depth, rng_mode=1))
                assert np.array_equal(got[i], alone), i
    batched = sequence.render_batches(dsrt, gpu_ctx, frame, frames[::-1], W, H, spp, depth, per_launch=3)      # three launches, two contexts in turn
    assert sorted(batched) == frames
    for i in frames:
        assert np.array_equal(batched[i], want[i]), f"batched: frame {i}"
    multi = dsrt.Multi([0, 0, 0], frames_in_flight=2)
    assert not multi.uses_rccl                           # ranks share a device here: copies stand in for the collective
    multi.upload(hs.view(cam0, sun0))
    cams, suns = zip(*[frame(i) for i in frames])
    imgs, sec = multi.render_sequence(dsrt.make_desc(W, H, spp, depth), list(cams), list(suns))
    assert sec > 0
    for i, img in zip(frames, imgs):
        assert np.array_equal(img, want[i]), f"multi: frame {i}"
    multi.close()


def test_batch_launch_gives_every_frame_its_own_image(dsrt, gpu_ctx, oracle, tmp_path):
    """dsrt_render_batch: the frames of a sequence as ONE launch -- a single queue that runs through the frames, a lane going from a pixel of
    one frame straight to the next -- over 7 poses from the far end to the near end, plus a ragged size.  rng_mode 0: every frame is the
    oracle's image byte for byte (uint8 and float); rng_mode 1: every frame is what dsrt_render gives for that camera alone.  At 64 and 300
    samples, so that with the larger count a pixel is a chain long enough for several frames' work to be in flight under it."""
    import torch
    _, hs, poses = _station_scene(dsrt, tmp_path, 20000)
    depth = 50
    frames = [0, 30, 70, 90, 96, 97, 98]
    for W, H, spp in ((320, 180, 64), (150, 85, 300)):
        def frame(i):
            fr = dsrt.pose_to_frame(poses[i])
            return dsrt.frame_camera(fr, 40.0, W, H, spp, depth), tuple(fr.sun_dir_model)
        cams, suns = zip(*[frame(i) for i in frames])
        gpu_ctx.upload(hs.view(cams[0], suns[0]))
        n = len(frames)
        rgb = torch.zeros(n * H * W * 3, dtype=torch.uint8, device="cuda")
        f32 = torch.zeros(n * H * W * 3, dtype=torch.float32, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream
        st = gpu_ctx.render_batch(dsrt.make_desc(W, H, spp, depth), list(cams), list(suns), rgb.data_ptr(), f32.data_ptr(), stream=stream, want_stats=True)
        assert st.kernel_ms > 0 and st.device_flags == 0
        got = rgb.cpu().numpy().reshape(n, H, W, 3)
        got32 = f32.cpu().numpy().reshape(n, H, W, 3)
        for k, i in enumerate(frames):
            want, want32, _ = oracle.render(hs.view(cams[k], suns[k]), W, H)
            assert np.array_equal(got[k], want), f"{W}x{H}@{spp} frame {i}: {(got[k] != want).any(axis=2).sum()} pixels differ"
            assert np.array_equal(got32[k].view(np.uint32), want32.view(np.uint32)), i
        gpu_ctx.render_batch(dsrt.make_desc(W, H, spp, depth, rng_mode=1), list(cams), list(suns), rgb.data_ptr(), stream=stream, want_stats=True)
        got1 = rgb.cpu().numpy().reshape(n, H, W, 3)
        for k, i in enumerate(frames):
            gpu_ctx.set_camera_sun(cams[k], suns[k])
            alone, _, _ = gpu_ctx.render_to_host(dsrt.make_desc(W, H, spp, depth, rng_mode=1))
            assert np.array_equal(got1[k], alone), f"rng_mode 1, frame {i}"
    # production size: 12 poses at 1920x1080 in one launch (25 M pixels: output indices far beyond one frame's), each against its own launch
    W, H, spp = 1920, 1080, 8
    big = list(range(0, 99, 9)) + [98]

    def frame_big(i):
        fr = dsrt.pose_to_frame(poses[i])
        return dsrt.frame_camera(fr, 40.0, W, H, spp, depth), tuple(fr.sun_dir_model)
    cams, suns = zip(*[frame_big(i) for i in big])
    rgb = torch.zeros(len(big) * H * W * 3, dtype=torch.uint8, device="cuda")
    gpu_ctx.render_batch(dsrt.make_desc(W, H, spp, depth), list(cams), list(suns), rgb.data_ptr(), stream=stream, want_stats=True)
    got = rgb.cpu().numpy().reshape(len(big), H, W, 3)
    for k, i in enumerate(big):
        gpu_ctx.set_camera_sun(cams[k], suns[k])
        alone, _, _ = gpu_ctx.render_to_host(dsrt.make_desc(W, H, spp, depth))
        assert np.array_equal(got[k], alone), f"1080p batch, frame {i}"
    assert got[-1].max() > 0
    # sharded batch: every rank renders ITS tiles of all the frames as one pool (the split that scales a sequence over a node); three ranks'
    # compact buffers, frame by frame through the gather layout and the de-interleave, are the whole-frame images
    W, H, spp = 150, 85, 64
    small = [0, 70, 90, 98]

    def frame_small(i):
        fr = dsrt.pose_to_frame(poses[i])
        return dsrt.frame_camera(fr, 40.0, W, H, spp, depth), tuple(fr.sun_dir_model)
    cams, suns = zip(*[frame_small(i) for i in small])
    world = 3
    lay = dsrt.shard_layout(dsrt.make_desc(W, H, spp, depth, shard_rank=0, shard_count=world))
    part_bytes = lay["rgb8_bytes_padded"]
    for rng_mode in (0, 1):
        parts = []
        for r in range(world):
            buf = torch.zeros(len(small) * part_bytes, dtype=torch.uint8, device="cuda")
            gpu_ctx.render_batch(dsrt.make_desc(W, H, spp, depth, shard_rank=r, shard_count=world, rng_mode=rng_mode), list(cams), list(suns), buf.data_ptr(),
                                 stream=stream, want_stats=True)
            parts.append(buf)
        for k, i in enumerate(small):
            gathered = torch.cat([p[k * part_bytes:(k + 1) * part_bytes] for p in parts])
            image = torch.zeros(W * H * 3, dtype=torch.uint8, device="cuda")
            gpu_ctx.deinterleave(dsrt.make_desc(W, H, spp, depth, shard_rank=0, shard_count=world), gathered.data_ptr(), image.data_ptr(), stream=stream)
            torch.cuda.synchronize()
            gpu_ctx.set_camera_sun(cams[k], suns[k])
            alone, _, _ = gpu_ctx.render_to_host(dsrt.make_desc(W, H, spp, depth, rng_mode=rng_mode))
            assert np.array_equal(image.cpu().numpy().reshape(H, W, 3), alone), f"sharded batch, rng_mode {rng_mode}, frame {i}"
    # other kinds of scene and tile: spheres + lights (no BVH culling), textures and mixed materials, 16-pixel tiles; three views each in one
    # launch, every view against the oracle
    for name in ("lights", "textured", "mixed", "c1_spheres"):
        world, cam_args, spp = CASES[name]
        hs2 = load_world(dsrt, world)
        W, H, depth2 = cam_args[3], cam_args[4], cam_args[5]
        views = [dsrt.camera_look_at(tuple(np.float64(cam_args[0]) * k), cam_args[1], cam_args[2], W, H, spp, depth2) for k in (1.0, 1.35, 0.8)]
        gpu_ctx.upload(hs2.view(views[0], SUN))
        rgb = torch.zeros(len(views) * H * W * 3, dtype=torch.uint8, device="cuda")
        f32 = torch.zeros(len(views) * H * W * 3, dtype=torch.float32, device="cuda")
        gpu_ctx.render_batch(dsrt.make_desc(W, H, spp, depth2, tile_size=16), views, [SUN] * len(views), rgb.data_ptr(), f32.data_ptr(), stream=stream, want_stats=True)
        got = rgb.cpu().numpy().reshape(len(views), H, W, 3)
        got32 = f32.cpu().numpy().reshape(len(views), H, W, 3)
        for k, v in enumerate(views):
            want, want32, _ = oracle.render(hs2.view(v, SUN), W, H)
            assert np.array_equal(got[k], want) and np.array_equal(got32[k].view(np.uint32), want32.view(np.uint32)), (name, k)
    # what a batch cannot be: counted, checked, or bigger than its 32-bit indices
    with pytest.raises(dsrt.DsrtError):
        gpu_ctx.render_batch(dsrt.make_desc(W, H, spp, depth, collect_counters=1), list(cams), list(suns), rgb.data_ptr())


def test_rccl_gather_runs_on_the_one_gpu_there_is(dsrt, gpu_ctx):
    """What of config 4's collective can execute on a one-GPU box: RCCL itself.  (1) the library's own path -- a one-rank communicator
    (ncclCommInitAll) and one ncclGather through it, checked byte for byte (dsrt_selftest_rccl_gather); (2) the path bench.py takes --
    torch.distributed's `nccl` backend (= RCCL) with world size 1 and dist.gather of a shard buffer.  The N-rank gather needs N devices;
    its receive layout, padding and de-interleave are what the 8-rank layout tests (one GPU, copies) and the gloo tests cover."""
    import torch
    import torch.distributed as dist
    dsrt.selftest_rccl_gather(0, 1 << 20)
    dsrt.selftest_rccl_gather(0, 777_601)                            # one rank's padded share of a 1080p frame, odd size
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
        dist.init_process_group("nccl", rank=0, world_size=1)
        try:
            part = torch.arange(777_600, dtype=torch.int64, device="cuda").to(torch.uint8)
            bucket = [torch.empty_like(part)]
            dist.gather(part, bucket, dst=0)
            torch.cuda.synchronize()
            assert torch.equal(bucket[0], part)
        finally:
            dist.destroy_process_group()


def test_eight_ranks_at_the_1080p_layout(dsrt, gpu_ctx, oracle, tmp_path):
    """BASELINE.json configs[3]'s layout on one GPU: 1920x1080 in 8x8 tiles is 32,400 tiles, 4,050 per rank with 8 ranks (equal here;
    the padded case is covered at 1918x1078 -> 32,400 tiles too but ragged edges, and by the CPU layout tests).  Eight shards rendered
    one after the other and de-interleaved must be the whole-frame render; the library's one-process path with eight ranks (all on GPU 0)
    must give the same image; three full rows are checked against the oracle."""
    import threading
    import torch
    _, hs, poses = _station_scene(dsrt, tmp_path, 60000)
    fr = dsrt.pose_to_frame(poses[98])
    for (W, H) in ((1920, 1080), (1918, 1078)):
        spp, depth, world = 2, 50, 8
        cam = dsrt.frame_camera(fr, 40.0, W, H, spp, depth)
        sun = tuple(fr.sun_dir_model)
        scene = hs.view(cam, sun)
        gpu_ctx.upload(scene)
        whole, _, _ = gpu_ctx.render_to_host(dsrt.make_desc(W, H, spp, depth))
        lay = dsrt.shard_layout(dsrt.make_desc(W, H, spp, depth, shard_count=world))
        tiles = ((W + 7) // 8) * ((H + 7) // 8)
        assert lay["tiles_total"] == tiles and lay["tiles_per_shard_padded"] == -(-tiles // world)
        gathered = torch.full((world * lay["rgb8_bytes_padded"],), 9, dtype=torch.uint8, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream
        for rank in range(world):
            part = gathered[rank * lay["rgb8_bytes_padded"]:(rank + 1) * lay["rgb8_bytes_padded"]]
            gpu_ctx.render(dsrt.make_desc(W, H, spp, depth, shard_rank=rank, shard_count=world), part.data_ptr(), stream=stream)
        image = torch.zeros(H * W * 3, dtype=torch.uint8, device="cuda")
        gpu_ctx.deinterleave(dsrt.make_desc(W, H, spp, depth, shard_count=world), gathered.data_ptr(), image.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        assert np.array_equal(image.cpu().numpy().reshape(H, W, 3), whole), (W, H)
        multi = dsrt.Multi([0] * world)
        multi.upload(scene)
        img, ms, sec = multi.render_frame(dsrt.make_desc(W, H, spp, depth), cam, sun)
        assert len(ms) == world and all(m > 0 for m in ms) and sec > 0
        assert np.array_equal(img, whole), (W, H)
        multi.close()
        if (W, H) == (1920, 1080):
            rows, want = [540, 3, 1076], {}

            def run(y):
                want[y] = oracle.render(scene, W, H, y, y + 1, want_f32=False)[0]
            threads = [threading.Thread(target=run, args=(y,)) for y in rows]
            for t in threads:
                t.start()
            for t in threads:
                t.join()
            for y in rows:
                r = H - 1 - y
                assert np.array_equal(whole[r], want[y][r]), y


def test_cpp_boundary_main_flow(dsrt, oracle, tmp_path):
    """The reference's main() flow through the C++ entry points: triangle_mesh -> hittable_list -> camera -> dsrt::build_gpu_scene ->
    gpu_render_scene -> dsrt::free_gpu_scene (deep-space-ray-tracer_amd/tools/main_flow_driver.cpp mirrors src/main.cpp:238-260, 399-428),
    twice in one process as the reference does per frame.  The PPM it leaves must be the oracle's image of that frame."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "deep-space-ray-tracer_amd", "main_flow_driver")
    assert os.path.exists(exe), "build it with `make tools`"
    obj = os.path.join(ASSETS, "station_3k.obj")
    poses_txt = os.path.join(GOLDEN, "rendezvous_1s_dt0_01s.txt")
    W, H, spp = 160, 90, 8
    out = tmp_path / "frame_0098.ppm"
    r = subprocess.run([exe, obj, poses_txt, "98", str(W), str(H), str(spp), "50", str(out), "2"], capture_output=True, text=True, timeout=300, cwd=tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    raw = out.read_bytes()
    head = f"P6\n{W} {H}\n255\n".encode()
    assert raw.startswith(head)
    hs = dsrt.HostScene().add_obj(obj)
    hs.build_bvh()
    poses = dsrt.read_pose_file(poses_txt)
    fr = dsrt.pose_to_frame(poses[98])
    cam = dsrt.frame_camera(fr, 40.0, W, H, spp, 50)
    want, _, _ = oracle.render(hs.view(cam, tuple(fr.sun_dir_model)), W, H)
    assert np.array_equal(np.frombuffer(raw[len(head):], np.uint8).reshape(H, W, 3), want)


def test_gather_calibration_kernel_runs(dsrt, gpu_ctx):
    """dsrt_microbench_gather (the roofline's calibration kernel): every mode gathers the records it says it does, at a plausible rate."""
    rates = {}
    for mode in (0, 1, 2):
        r = dsrt.microbench_gather(mode, dependent=False, live_lanes=64, pad_valu=0, table_bytes=1 << 20, iters=200)
        assert r["records"] == 64.0 * 200 * (r["records"] / (64.0 * 200)) and r["ms"] > 0
        rates[mode] = r["Grecords_per_s"]
        half = dsrt.microbench_gather(mode, dependent=True, live_lanes=27, pad_valu=16, table_bytes=1 << 20, iters=100)
        assert 0.2 < half["records"] / (r["records"] / 200 * 100) < 0.7          # about 27 of 64 lanes take part
    assert all(5.0 < v < 5000.0 for v in rates.values()), rates
    with pytest.raises(dsrt.DsrtError):
        dsrt.microbench_gather(3)


class DeviceMatkat:
    """The render kernel's own helpers (csrc/device_math.h) through dsrt_selftest_devkat: 12 words in, 12 words out per case."""

    def __init__(self, ctx):
        self.ctx = ctx

    @staticmethod
    def _pack(rows, **cols):
        n = rows
        a = np.zeros((n, 12), np.float32)
        for k, (col, val) in cols.items():
            val = np.asarray(val)
            if val.dtype in (np.uint32, np.int32):
                val = val.astype(np.uint32).view(np.float32)
            a[:, col:col + (val.shape[1] if val.ndim == 2 else 1)] = val.reshape(n, -1)
        return a

    def reflect(self, v, n):
        return self.ctx.selftest_devkat(0, self._pack(len(v), v=(0, v), n=(3, n)))[:, :3]

    def refract(self, v, n, eta):
        return self.ctx.selftest_devkat(1, self._pack(len(v), v=(0, v), n=(3, n), eta=(6, eta)))[:, :3]

    def metal(self, d, n, fuzz, state):
        out = self.ctx.selftest_devkat(2, self._pack(len(d), v=(0, d), n=(3, n), fuzz=(6, fuzz), state=(7, state.astype(np.uint32))))
        return out[:, :3], out[:, 3] != 0.0, out[:, 4].copy().view(np.uint32)

    def dielectric(self, d, n, front, ref_idx, state):
        out = self.ctx.selftest_devkat(3, self._pack(len(d), v=(0, d), n=(3, n), ref=(6, ref_idx), state=(7, state.astype(np.uint32)), front=(8, front.astype(np.uint32))))
        return out[:, :3], out[:, 4].copy().view(np.uint32)

    def onb(self, n):
        out = self.ctx.selftest_devkat(4, self._pack(len(n), v=(0, n)))
        return out[:, 0:3], out[:, 3:6], out[:, 6:9]

    def schlick(self, cos, ratio):
        return self.ctx.selftest_devkat(5, self._pack(len(cos), c=(0, cos), r=(1, ratio)))[:, 0]


def test_device_material_helpers_match_the_reference_known_answers(dsrt, gpu_ctx, oracle):
    """The kernel's reflect / refract / scatter_metal / scatter_dielectric / build_onb / schlick (csrc/device_math.h, the functions path_machine.h
    calls) against what the reference's host code computed (tests/golden/ref_matkat.json), and bit for bit against the oracle's copies."""
    from test_oracle import OracleMatkat, check_material_known_answers
    orc = OracleMatkat(oracle)
    dev = DeviceMatkat(gpu_ctx)
    check_material_known_answers(dev, orc.normalize)
    # and, on inputs no golden covers (fuzz > 0: the rejection loop feeds the direction; the refracting branch, which draws): device == oracle, bits
    rng = np.random.default_rng(5)
    n = 256
    d = rng.normal(size=(n, 3)).astype(np.float32) * 30
    nn = rng.normal(size=(n, 3)).astype(np.float32)
    nn /= np.linalg.norm(nn, axis=1, keepdims=True).astype(np.float32)
    fuzz = rng.uniform(-0.2, 1.3, n).astype(np.float32)
    state = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    a, b = dev.metal(d, nn, fuzz, state), orc.metal(d, nn, fuzz, state)
    assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    front = rng.integers(0, 2, n).astype(np.int32)
    ref_idx = rng.choice(np.array([1.5, 1.33, 2.4, 0.0, -1.0, np.inf, 0.7], np.float32), n)
    a, b = dev.dielectric(d, nn, front, ref_idx, state), orc.dielectric(d, nn, front, ref_idx, state)
    assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)) and np.array_equal(a[1], b[1])
    assert 16 < (b[1] != state).sum() < n                                     # both the drawing and the non-drawing branch occur


@pytest.mark.parametrize("scale", [1.0e-15, 1.0 / 64.0, 4096.0, 1.0e9])
def test_scaled_scenes_match_the_oracle_bit_for_bit(dsrt, gpu_ctx, oracle, scale):
    """The node visit orders its two children by 2 d instead of the reference's d (fma(-2, o, lo + hi) = 2 (0.5 (lo + hi) - o), render_kernel.hip): an identity
    of IEEE arithmetic as long as no intermediate is subnormal or overflows.  The station at 1/64 of its size, 4096 times and a billion times its size (coordinates
    up to 5e10, products up to 1e22), camera moved with it: image bits and every work counter -- which depend on the ORDER in which children are visited -- equal the
    oracle's, which computes d the reference's way."""
    from conftest import ASSETS
    hs = dsrt.HostScene().add_obj(os.path.join(ASSETS, "station_3k.obj"), scale=scale)
    hs.build_bvh()
    W, H, spp, depth = 160, 90, 8, 50
    cam = dsrt.camera_look_at(tuple(c * scale for c in (12.0, 9.0, 38.0)), (0.0, 0.0, 0.0), 40.0, W, H, spp, depth)
    scene = hs.view(cam, SUN)
    want_rgb, want_f32, want_cnt = oracle.render(scene, W, H)
    gpu_ctx.upload(scene)
    rgb, f32, st = gpu_ctx.render_to_host(dsrt.make_desc(W, H, spp, depth, collect_counters=2), want_f32=True)
    assert np.array_equal(rgb, want_rgb) and np.array_equal(f32.view(np.uint32), want_f32.view(np.uint32))
    for key in ("rays", "box_fetches", "nodes_entered", "internal_entered", "tri_tests", "hit_updates", "max_stack"):
        assert getattr(st, key) == want_cnt[key], (key, getattr(st, key), want_cnt[key])
    if scale >= 1.0:
        assert int((rgb.max(axis=2) > 0).sum()) > 0.02 * W * H       # (at 1/64 the reference's absolute epsilons -- t_min 1e-3, |det| 1e-8 -- eat most hits: compared all the same)


def test_checked_build_flags_a_corrupt_node_reference_instead_of_hanging(dsrt, gpu_ctx, oracle):
    """A child reference in 0 .. 61 belongs to no class of device_layout.h (leaf < 0, none 62, pop 63, internal >= 64): a lane holding one would count as walking for
    ever.  Upload validates every index, so only memory corruption can produce one -- which is what the bounds-checked build exists for: it must come back with
    kFlagBadNodeRef (status bit 1), not run into the launch timeout.  The hook overwrites the root record's left child reference (word 12) in the resident scene."""
    hs, scene, W, H, spp, depth = _scene(dsrt, "station_near")
    gpu_ctx.upload(scene)
    good, _, _ = gpu_ctx.render_to_host(dsrt.make_desc(W, H, spp, depth, checked=1))
    for bad in (5, 61, 0):
        old = gpu_ctx.poke_node_word(12, bad)
        try:
            with pytest.raises(dsrt.DsrtError) as e:
                gpu_ctx.render_to_host(dsrt.make_desc(W, H, spp, depth, checked=1, tune=(0, 0, 0, 1)))      # (no pre-pass: its any-hit walk is an unchecked kernel)
            assert e.value.code == -7 and "0x1" in str(e.value)             # DSRT_ERR_DEVICE_FLAG, kFlagBadNodeRef
        finally:
            gpu_ctx.poke_node_word(12, old)
    again, _, _ = gpu_ctx.render_to_host(dsrt.make_desc(W, H, spp, depth, checked=1))
    assert np.array_equal(good, again)
    want, _, _ = oracle.render(scene, W, H)
    assert np.array_equal(good, want)
