"""BASELINE.json configs[4] as one job: every pose of the file rendered once, the scene resident across frames.

The reference's frame loop (src/main.cpp:310-431) rebuilds the BVH, re-uploads the whole scene and renders one frame at a time.
Here only camera and sun change per frame (dsrt_scene_set_camera_sun) and `inflight` frames are in flight at once on one GPU,
each on its own HIP stream with its own context (dsrt_ctx_clone: the contexts SHARE the resident scene, they differ in camera,
queue words and scratch) and its own device + pinned host image.  Why several frames in flight: with the reference's one LCG
stream per pixel a frame ends in a tail of a few spp-sample serial chains, and the next frames' workgroups fill the CUs that
tail leaves idle.

`render_batches` goes one step further (dsrt_render_batch): the frames of a launch are ONE pool of work -- a lane that finishes a pixel of
one frame goes straight on to the next frame -- so no chip time is lost to any frame's tail and no hardware queue limits how many frames
overlap: the 99-pose sequence runs at 56 frames/s in rng_mode 0 on one MI355X against 40 with 16 separate launches in flight.

With more than one rank (one process per GPU, torch.distributed) the job splits one of two ways:
  split="tiles"   (default) every rank renders ITS interleaved screen tiles of EVERY frame, as sharded batch launches: the rank's tiles of all
                  the frames of a launch are one pool of work, so loads are equal by construction and a frame's serial chains run under the
                  other frames' bulk; one gather per launch for all its frames, de-interleave on rank 0.
  split="frames"  the no-collective alternative: whole poses are dealt to ranks (by estimated cost, or round-robin); bound by the nearest
                  frame itself at 8 ranks.
Neither split has run on more than one GPU yet; the speed-ups quoted in DESIGN.md section 5 are projections from single-GPU shard probes.
"""
import numpy as np


def approach_cost(sep_m, scene_radius_m):
    """Relative cost of a frame of the approach: 1 / (distance to the model's centre + the model's radius).  Measured on the stand-in station
    at 1080p x 250 (tools/frame_costs_probe.py): dealing frames by this estimate keeps 8 ranks within 5-9 % of each other, round-robin
    dealing -- the nearest frames cost 30 times the farthest, and one rank gets the nearest -- leaves 28 % of the node idle."""
    return 1.0 / (float(sep_m) + float(scene_radius_m))


def frame_assignment(frame_ids, rank, world, split, costs=None):
    """The frames this rank works on: all of them (tile split: every rank renders its tiles of every frame), or its share of whole frames.
    Without `costs` frames are dealt round-robin; with costs (one number per frame, any unit) longest-processing-time first: frames in
    order of falling cost, each to the rank with the least so far (ties: lowest rank) -- the same answer on every rank, no communication."""
    frame_ids = list(frame_ids)
    if split == "tiles" or world == 1:
        return frame_ids
    if costs is None:
        return frame_ids[rank::world]
    costs = [float(c) for c in costs]
    if len(costs) != len(frame_ids):
        raise ValueError("one cost per frame")
    load = [0.0] * world
    mine = []
    for k in sorted(range(len(frame_ids)), key=lambda k: (-costs[k], k)):
        r = min(range(world), key=lambda r: (load[r], r))
        load[r] += costs[k]
        if r == rank:
            mine.append(frame_ids[k])
    return sorted(mine)


class FramePipeline:
    """K frames in flight on one GPU.  `submit(i, camera, sun)` queues frame i on the next slot (waiting for that slot's previous
    frame first); `drain()` waits for everything.  `on_frame(i, host_uint8_tensor)` is called when a frame's image has landed in
    pinned host memory (the tensor is reused by later frames: copy it if you keep it)."""

    def __init__(self, d, ctx, W, H, spp, depth, inflight=4, rng_mode=0, device=None, shard=None, on_frame=None, tune=(0, 0, 0, 0)):
        import torch
        self.torch, self.d = torch, d
        self.W, self.H, self.spp, self.depth, self.rng_mode = W, H, spp, depth, rng_mode
        self.tune = tuple(tune)                                       # scheduling knobs of DsrtRenderDesc (experiments; none changes a pixel)
        self.dev = device if device is not None else torch.device("cuda", ctx.device)
        self.K = max(1, int(inflight))
        self.ctxs = [ctx] + [ctx.clone() for _ in range(1, self.K)]
        with torch.cuda.device(self.dev):
            self.streams = [torch.cuda.Stream() for _ in range(self.K)]
        self.shard = shard                                            # None or (rank, world, gather_fn)
        n_img = W * H * 3
        if shard:
            rank, world, _ = shard
            lay = d.shard_layout(d.make_desc(W, H, spp, depth, shard_rank=rank, shard_count=world))
            self.parts = [torch.zeros(lay["rgb8_bytes_padded"], dtype=torch.uint8, device=self.dev) for _ in range(self.K)]
        root = (not shard) or shard[0] == 0
        self.images = [torch.zeros(n_img, dtype=torch.uint8, device=self.dev) for _ in range(self.K)] if root else None
        self.host = [torch.empty(n_img, dtype=torch.uint8).pin_memory() for _ in range(self.K)] if root else None
        self.pending = [None] * self.K
        self.on_frame = on_frame
        self.n = 0

    def _retire(self, slot):
        # Only a consumer of the pinned image needs the host to wait for a slot: without one, stream order alone protects the slot's
        # buffers (its next render queues behind its last copy), and the host must NOT block here -- while it waited for one long
        # near frame, the slots whose short far frames had finished would stand empty.
        if self.pending[slot] is not None:
            if self.on_frame is not None and self.host is not None:
                self.streams[slot].synchronize()
                self.on_frame(self.pending[slot], self.host[slot])
            self.pending[slot] = None

    def submit(self, frame_index, camera, sun_dir):
        torch, d = self.torch, self.d
        slot = self.n % self.K
        self.n += 1
        self._retire(slot)
        c, stream = self.ctxs[slot], self.streams[slot]
        c.set_camera_sun(camera, tuple(sun_dir))
        if self.shard:
            rank, world, gather = self.shard
            desc = d.make_desc(self.W, self.H, self.spp, self.depth, shard_rank=rank, shard_count=world, rng_mode=self.rng_mode, tune=self.tune)
        else:
            desc = d.make_desc(self.W, self.H, self.spp, self.depth, rng_mode=self.rng_mode, tune=self.tune)
        with torch.cuda.stream(stream):                               # everything of this frame is ordered on its slot's stream
            target = self.parts[slot] if self.shard else self.images[slot]
            c.render(desc, target.data_ptr(), stream=stream.cuda_stream)
            if self.shard:
                flat = gather(self.parts[slot], world, rank)
                if rank == 0:
                    c.deinterleave(desc, flat.data_ptr(), self.images[slot].data_ptr(), stream=stream.cuda_stream)
            if self.host is not None:
                self.host[slot].copy_(self.images[slot], non_blocking=True)
        self.pending[slot] = frame_index

    def drain(self):
        for k in range(self.K):
            self._retire((self.n + k) % self.K)                       # oldest first: frames are handed over in submission order
        for s in self.streams:
            s.synchronize()

    def close(self):
        self.drain()
        for c in self.ctxs[1:]:
            c.close()
        self.ctxs = self.ctxs[:1]


def render_frames(d, ctx, hs_frame, frame_ids, W, H, spp, depth, inflight=4, rng_mode=0, keep=True):
    """Convenience for tests and tools: render `frame_ids` through a FramePipeline on ctx's GPU; `hs_frame(i)` returns
    (camera, sun_dir) of frame i.  Returns {frame id: H x W x 3 uint8 array} (or {} with keep=False)."""
    out = {}

    def got(i, host):
        if keep:
            out[i] = host.numpy().reshape(H, W, 3).copy()
    pipe = FramePipeline(d, ctx, W, H, spp, depth, inflight=inflight, rng_mode=rng_mode, on_frame=got)
    for i in frame_ids:
        cam, sun = hs_frame(i)
        pipe.submit(i, cam, sun)
    pipe.close()
    return out


def render_batches(d, ctx, hs_frame, frame_ids, W, H, spp, depth, per_launch=32, rng_mode=0, keep=True, on_frame=None):
    """Render `frame_ids` through dsrt_render_batch on ctx's GPU, `per_launch` frames per launch, two contexts (ctx and a clone sharing its
    scene) taking the launches in turn so that one launch's images travel to pinned host memory while the next renders.  Frames are taken
    in the order given (put the costliest -- nearest -- first: the job then ends on short chains).  Returns {frame id: H x W x 3 uint8}."""
    import torch
    ids = list(frame_ids)
    dev = torch.device("cuda", ctx.device)
    ctxs = [ctx, ctx.clone()]
    with torch.cuda.device(dev):
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    n_img = W * H * 3
    bufs = [torch.zeros(per_launch * n_img, dtype=torch.uint8, device=dev) for _ in range(2)]
    host = [torch.empty(per_launch * n_img, dtype=torch.uint8).pin_memory() for _ in range(2)]
    desc = d.make_desc(W, H, spp, depth, rng_mode=rng_mode)
    held = [[], []]
    out = {}

    def retire(slot):
        if held[slot]:
            streams[slot].synchronize()
            for q, i in enumerate(held[slot]):
                img = host[slot][q * n_img:(q + 1) * n_img]
                if on_frame is not None:
                    on_frame(i, img)
                if keep:
                    out[i] = img.numpy().reshape(H, W, 3).copy()
            held[slot] = []

    try:
        for k in range(0, len(ids), per_launch):
            group = ids[k:k + per_launch]
            slot = (k // per_launch) % 2
            retire(slot)
            cams, suns = zip(*[hs_frame(i) for i in group])
            with torch.cuda.stream(streams[slot]):
                ctxs[slot].render_batch(desc, list(cams), [tuple(s) for s in suns], bufs[slot].data_ptr(), stream=streams[slot].cuda_stream)
                host[slot][:len(group) * n_img].copy_(bufs[slot][:len(group) * n_img], non_blocking=True)
            held[slot] = group
        n_launches = (len(ids) + per_launch - 1) // per_launch
        retire(n_launches % 2)                    # oldest first: the last launch sits in slot (n_launches - 1) % 2, the one before it in the other
        retire((n_launches + 1) % 2)
    finally:
        for st in streams:
            st.synchronize()
        ctxs[1].close()
    return out
