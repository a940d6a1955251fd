#!/usr/bin/env python3
"""bench.py -- Msamples/s of the per-pixel x spp sampling loop on the ISS-mesh frame BASELINE.json quotes
(1920x1080 @ 1000 spp, max_depth 50), on 1..N MI355X of one node.

A "step" is one complete render of the frame: every pixel, every sample, scene already resident in HBM in traversal layout
(upload, BVH build and OBJ parsing are outside the timed region, as BASELINE.md section 3 prescribes; the reference redoes them per
frame -- their cost is reported in `setup`).  For N > 1 the image is sharded by interleaved 8x8 screen tiles (tile t -> rank t mod N),
each rank renders its tiles into a compact buffer, one RCCL gather brings them to rank 0 and a small kernel restores image order -- all
inside the step.  `--gpus N` without a torch.distributed launcher starts one (one rank per GPU) as a child process and exits with its
code; `--single-process` instead drives the N GPUs from this one process through the library's own RCCL path (dsrt_multi_*).

Mesh: the real ISS OBJ is not available, so unless --obj is given the procedural stand-in from
deep-space-ray-tracer_amd/meshgen.py is generated (--tris, default 1,000,000 triangles).  Pose: --frame of the reference's
rendezvous_1s_dt0_01s.txt (tests/golden/ copy).  The default frame is 98 (camera 35.7 m from the station, which fills the view);
frame 0 (1787 m, 99.9 % of the tiles provably empty and culled) is reported in `extras` as what it is: a culling rate.

Tree: by default rays walk the CERTIFIED SECOND TREE (include/dsrt.h, dsrt_ctx_set_certified_tree): a SAH tree whose every answer the kernel certifies against the
reference's own median-split tree, re-walking that tree when it cannot -- the reference's bytes with a third fewer node visits.  The line proves it on the run itself:
`reference_kernel_fixture` compares the image of the last timed step with the image the reference's own kernel rendered of this frame (tests/golden/), `parity_rows`
with the CPU oracle's rows; `extras.reference_walk_only` times the same frame with every ray on the reference tree (`--tree reference` makes that the headline).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and, at N = 1, `cpu_baseline`.

roofline.  The path is branchy fp32 vector arithmetic over a scene that lives in L2 / Infinity Cache; the counters say it is bound
by vector-ALU ISSUE (and, second, by the L1's request rate), not by HBM.  The object therefore carries
  bound / achieved / peak / frac   the binding resource: VALU wave-instructions per second of the render kernel (PMC SQ_INSTS_VALU
                                   of this very run, rocprofv3 child pass) against the chip's issue PEAK, 1 wave64 VALU instruction
                                   per SIMD per 2 cycles (the guide's SIMD-32 figure; `peak_basis` says so in the line).
  issue_costs, mix                 what this device charges per instruction class, calibrated live (dsrt_microbench_valu), and what
                                   the kernel's own instruction mix therefore allows: most of its stream (packed fp32, compares,
                                   selects, min/max) issues at half the simple-op rate, so `valu_busy_estimate` -- issue cycles
                                   spent / issue cycles available -- is the fraction that says how close to issue-bound it runs
  useful_lane_frac                 frac x valu_lane_occupancy: lanes doing useful work per peak lane-slot
  l1                               16-byte L1 requests per second (PMC TCP_TOTAL_CACHE_ACCESSES) against the gather ceiling of the
                                   calibration kernel run live (dsrt_microbench_gather: same launch shape, every lane gathering random
                                   64-byte records as 4 x 16 B from an L2-resident table)
  achieved_algorithmic, hbm        SURVEY.md section 8(d)'s algorithmic bytes per launch / kernel time (counts every re-read the caches
                                   serve, so it is NOT bounded by the HBM peak), and the HBM-side figure: FETCH_SIZE / WRITE_SIZE of this
                                   run (separate PMC passes, gfx950 read-side correction) / kernel time / 8 TB/s
"""
import argparse
import csv
import ctypes as C
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
VALU_PEAK_CYCLES_PER_WAVE_INSTR = 2.0   # /opt/skills/guides/MI355X_MICROARCH.md: a wave64 VALU instruction issues over 2 cycles on a SIMD-32 once a second wave
                                        # is on the SIMD.  Measured here (dsrt_microbench_valu, profiles/r03/valu_issue_costs.md): v_add / v_mul / v_fma / v_mov /
                                        # v_and 2.3-2.45 cycles; v_pk_*, v_cmp, v_cndmask_e64, v_min / v_max, shifts 4.1-4.4; v_rcp / v_sqrt 8.1-8.5.  Round 2
                                        # priced the kernel against 4 cycles (cross-dependent fma chains): right for this kernel's MIX, wrong as the chip's peak.
# VALU instruction classes of the production kernel's three hot loops: static counts (tools/isa_mix.py --loops on this build) weighted by the loops'
# shares of the issued stream, node loop 0.57 / leaf pass 0.21 / advance 0.22 (trip counts of the counting build x VALU per trip against SQ_INSTS_VALU;
# profiles/r03/isa_mix.txt): share of simple-rate, half-rate and quarter-rate instructions
KERNEL_VALU_CLASS_SHARES = {"simple": 0.26, "half": 0.67, "quarter": 0.07}
RENDER_KERNEL = "dsrt_render_kernel<8, false, false, true, 0"       # (prefix: the LEAN instantiation appends a template argument)


def algorithmic_bytes(st, pixels):
    """B = 24*n_box + 16*n_enter + 40*n_tri + 44*n_upd + 48*shaded + 24*sphere tests + 3*pixels,
    with the counters of the kernel that was actually run (any-hit shadow rays included)."""
    return (24 * st.box_fetches + 16 * st.nodes_entered + 40 * st.tri_tests + 44 * st.hit_updates + 48 * st.shaded_hits +
            24 * st.sphere_tests + 12 * st.tex_fetches + 3 * pixels)


def host_cores():
    """Cores this process may really use: the affinity mask, the cgroup quota, and never more than the 16 a one-GPU box grants."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 16))


def cpu_baseline(d, scene, W, H, spp, budget_s, gpu_rgb8=None):
    """The CPU oracle (oracle/dsrt_oracle.c, kind "port") on a bounded sample of the SAME frame: single rows at full spp, spread evenly
    over the whole image height (a stratified sample: centre rows alone would over-weight the station), one row per thread per round,
    rounds added until the time budget is used up.  The rows' pixels are KEPT: with `gpu_rgb8` (the H x W x 3 image of the timed GPU step,
    on the host) they are compared byte for byte -- outside any timed region -- and the result is returned as `parity_rows`, which ties the
    headline number to bytes the oracle agrees with on the very mesh and frame it was measured on."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import Oracle
    orc = Oracle()
    cores = host_cores()
    order, step = [], H
    seen = set()
    while step >= 1:                                   # rows at H/2, then H/4 and 3H/4, ...: any prefix is spread over the image
        for y in range(step // 2, H, max(1, step)):
            if y not in seen:
                seen.add(y)
                order.append(y)
        step //= 2
    cpu_img = np.zeros((H, W, 3), np.uint8)            # dsrt_oracle_render_rows writes kernel row y into image row H - 1 - y; threads touch disjoint rows

    def one_round(rows):
        cnt = [(C.c_uint64 * len(Oracle.COUNTER_NAMES))() for _ in rows]
        jobs = [threading.Thread(target=orc.lib.dsrt_oracle_render_rows, args=(C.byref(scene), W, H, y, y + 1, cpu_img.ctypes.data, None, cnt[i]))
                for i, y in enumerate(rows)]
        for th in jobs:
            th.start()
        for th in jobs:
            th.join()
        return len(jobs)

    rows, pos = 0, 0
    t0 = time.perf_counter()
    while pos < len(order):
        rows += one_round(order[pos:pos + cores])
        pos += cores
        dt = time.perf_counter() - t0
        if dt >= budget_s:
            break
    dt = time.perf_counter() - t0
    base = {"value": rows * W * spp / dt / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": f"{rows} full rows spread evenly over the image height (rows {sorted(order[:rows])[:3]}...), same frame at {W}x{H}x{spp} "
                      f"({rows * W} pixels), {dt:.1f} s wall"}
    parity = None
    if gpu_rgb8 is not None:
        img_rows = [H - 1 - y for y in order[:rows]]
        diff = (cpu_img[img_rows] != gpu_rgb8[img_rows]).any(axis=2)
        parity = {"rows": rows, "pixels": int(rows * W), "mismatched": int(diff.sum()), "lit_pixels": int((cpu_img[img_rows].max(axis=2) > 0).sum()),
                  "compared": "rgb8 bytes of the oracle's rows against the same rows of the GPU image of the last timed step (rng_mode 0)"}
    return base, parity


def reference_kernel_on_this_gpu(d, ctx, part, stream, obj_path, fr, frame_idx, W, H, spp, depth):
    """The reference's OWN render kernel on this GPU, next to ours: oracle/_ref/ref_gpu is the reference's loader, build_gpu_scene and gpu_render_scene
    (src/gpu_render.cu), translated CUDA -> HIP by the image's hipify-perl and compiled with hipcc from the sources where they lie (oracle/Makefile; a
    checker binary like oracle/_ref/ref_host, prebuilt, run as a child process).  Same mesh, pose and size as the headline at a reduced sample count (its rate
    does not depend on it: profiles/r03/reference_kernel_hipified_headline_frame.json has the full 1000 spp).  Two things come out of it, both outside every
    timed region: its rate -- what "a hipify of src/gpu_render.cu" does on an MI355X -- and the number of pixels in which its image differs from this
    library's in math_mode 1 (the same kernels with the device math library's sinf / cosf / powf, as that build has them)."""
    import numpy as np
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_gpu")
    if not os.path.exists(exe):
        return None
    tmp = tempfile.mkdtemp(prefix="dsrt_refgpu_", dir="/tmp")
    try:
        world = os.path.join(tmp, "bench.world")
        with open(world, "w") as f:
            f.write(f"obj {obj_path}\n")
        cam_from, sun = [repr(float(v)) for v in fr.cam_in_model], [repr(float(v)) for v in fr.sun_dir_model]
        r = subprocess.run([exe, world, str(W), str(H), str(spp), str(depth), *cam_from, "0", "0", "0", "40", *sun, os.path.join(tmp, "ref.ppm"), "1"], cwd=tmp,
                           capture_output=True, text=True, timeout=600)
        ref = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        ms = ref["gpu_render_scene_second_call_ms"]
        ours_ms = {}
        for mode in (0, 1):
            dsc = d.make_desc(W, H, spp, depth, math_mode=mode)
            ctx.render(dsc, part.data_ptr(), stream=stream, want_stats=True)
            ours_ms[mode] = ctx.render(dsc, part.data_ptr(), stream=stream, want_stats=True).kernel_ms
        our_img = part[:W * H * 3].cpu().numpy().reshape(H, W, 3)                      # the math_mode 1 image
        data = open(os.path.join(tmp, "ref.ppm"), "rb").read()
        header = f"P6\n{W} {H}\n255\n".encode()
        ref_img = np.frombuffer(data[len(header):], np.uint8).reshape(H, W, 3)
        return {"what": "the reference's loader + build_gpu_scene + gpu_render_scene, hipify-perl + hipcc (oracle/_ref/ref_gpu), second call on a resident scene, "
                        "copy-back and PPM write included", "workload": f"{ref['triangles']} triangles, pose frame {frame_idx}, {W}x{H} @ {spp} spp, max_depth {depth}",
                "gpu_render_scene_ms": ms, "Msamples/s": W * H * spp / ms / 1e3, "build_gpu_scene_ms": ref["build_gpu_scene_ms"],
                "this_library_kernel_ms_same_frame": {"math_mode_0": ours_ms[0], "math_mode_1": ours_ms[1]}, "this_library_is_faster_by": ms / ours_ms[0],
                "image_comparison": {"pixels": W * H, "lit_pixels": int((ref_img.max(axis=2) > 0).sum()), "differing_pixels": int((ref_img != our_img).any(axis=2).sum()),
                                     "compared": "the reference kernel's image_gpu.ppm against this library in math_mode 1 (the device math library's sinf / cosf / powf, as the "
                                                 "reference's build has them; math_mode 0, the default, uses include/dsrt_detmath.h for those three and equals the CPU "
                                                 "oracle: parity_rows), rgb8 bytes"}}
    except Exception as e:  # noqa: BLE001 -- a baseline that cannot run is reported as such, it never stops the bench
        return {"error": str(e)[:300]}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def book_baseline(obj_path, fr, cores, tsv_path=None):
    """The book-style CPU render BASELINE.json names: the reference's own hittable_list / sphere / triangle_mesh / material
    classes (compiled from /root/reference into oracle/_ref/book_render, prebuilt) under our book-style pixel loop.
      * config C1 (RTIOW three-sphere scene): 200x112 @ 16 spp as configured, and a THREAD SWEEP at 800x448 @ 64 spp (22.9 M samples,
        a run of seconds) over 1, 2, 4, ... cores written as `num_threads<TAB>duration_ns` (the schema of the reference's
        scripts/performance.py:20-22); one process per core on disjoint row bands -- the classes draw from rand(), whose
        process-wide lock stops threads of one process from scaling;
      * the ISS mesh: triangle_mesh::hit is a linear scan over all triangles, so a 24x14 @ 1 spp corner of the bench frame is
        timed on one core and the rate is quoted as measured (linear in samples)."""
    exe = os.path.join(ROOT, "oracle", "_ref", "book_render")
    if not os.path.exists(exe):
        return None
    assets = os.path.join(ROOT, "tests", "golden", "assets")

    def c1(W, H, spp, procs):
        t0 = time.perf_counter()
        ps = [subprocess.Popen([exe, "c1_spheres.world", str(W), str(H), str(spp), "50", "1", "-2", "2", "1", "0", "0", "-1", "20", "0.3", "-0.8", "0.5", "-",
                                str(H * i // procs), str(H * (i + 1) // procs)], cwd=assets, stdout=subprocess.PIPE, text=True) for i in range(procs)]
        outs = [json.loads(p.communicate(timeout=600)[0]) for p in ps]
        wall = time.perf_counter() - t0
        return wall, sum(o["samples"] for o in outs), max(o["seconds"] for o in outs)

    wall, samples, slowest = c1(200, 112, 16, cores)
    out = {"kind": "reference classes + our book-style loop (oracle/book_render_driver.cpp)",
           "c1": {"workload": "RTIOW 3 spheres + ground (tests/golden/assets/c1_spheres.world), 200x112 @ 16 spp, max_depth 50", "cores": cores,
                  "Msamples/s": samples / slowest / 1e6, "wall_s": wall}}
    sweep, n = [], 1
    while n <= cores:
        w, s, _ = c1(800, 448, 64, n)
        sweep.append({"num_threads": n, "duration_ns": int(w * 1e9), "Msamples/s": s / w / 1e6})
        n *= 2
    out["c1_thread_sweep"] = {"workload": "same scene, 800x448 @ 64 spp, max_depth 50, one process per core", "rows": sweep}
    if tsv_path:
        try:
            os.makedirs(os.path.dirname(tsv_path), exist_ok=True)
            with open(tsv_path, "w") as f:
                f.write("num_threads\tduration_ns\n")
                for r in sweep:
                    f.write(f"{r['num_threads']}\t{r['duration_ns']}\n")
            out["c1_thread_sweep"]["tsv"] = os.path.relpath(tsv_path, ROOT)
        except OSError:
            pass
    try:
        world = f"/tmp/dsrt_book_{os.getpid()}.world"
        with open(world, "w") as f:
            f.write(f"obj {obj_path}\n")
        sun = [str(v) for v in fr.sun_dir_model]
        cam = [str(v) for v in fr.cam_in_model]
        res = subprocess.run([exe, world, "24", "14", "1", "50", "1", *cam, "0", "0", "0", "40", *sun], stdout=subprocess.PIPE, text=True, timeout=240)
        r = json.loads(res.stdout)
        out["iss_mesh"] = {"workload": "bench mesh and pose, 24x14 @ 1 spp, max_depth 50 (triangle_mesh::hit scans every triangle per ray)", "cores": 1,
                           "Msamples/s": r["msamples_per_s"], "seconds": r["seconds"]}
        os.remove(world)
    except Exception as e:  # noqa: BLE001 -- a baseline that cannot run is reported as such, it never stops the bench
        out["iss_mesh"] = {"error": str(e)[:200]}
    return out


# ------------------------------------------------------------------------------------------------------------------
# PMC counters of THIS workload, measured in this run: bench.py starts itself under rocprofv3 as a child process, one pass per
# counter set (never combined with tracing; FETCH_SIZE and WRITE_SIZE in passes of their own, as the guide prescribes).
# ------------------------------------------------------------------------------------------------------------------
PMC_PASSES = [["FETCH_SIZE"], ["WRITE_SIZE"],
              ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_INSTS_VMEM_RD", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "GRBM_GUI_ACTIVE"],
              ["TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum", "TCC_HIT_sum", "TCC_MISS_sum"]]


def pmc_counters(workload_args, timeout_s=240):
    """{counter: value} for the LAST dispatch of the production render kernel in a child run of this workload, or None."""
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None
    out, base = {}, tempfile.mkdtemp(prefix="dsrt_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for i, names in enumerate(PMC_PASSES):
        d = os.path.join(base, f"p{i}")
        cmd = [rocprof, "--pmc", *names, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__), "--pmc-child", *workload_args]
        try:
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s, check=True)
        except (subprocess.SubprocessError, OSError):
            continue
        per_dispatch = {}
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Kernel_Name"].replace(" ", "").startswith("void" + RENDER_KERNEL.replace(" ", "")) or RENDER_KERNEL.replace(" ", "") in r["Kernel_Name"].replace(" ", ""):
                    per_dispatch.setdefault(int(r["Dispatch_Id"]), {}).setdefault(r["Counter_Name"], 0.0)
                    per_dispatch[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
        if per_dispatch:
            out.update(per_dispatch[max(per_dispatch)])
    shutil.rmtree(base, ignore_errors=True)
    return out or None


def spawn_distributed(n):
    """`--gpus N` from a plain interpreter: start the N-rank job (one process per GPU) as a child and exit with its code.  Nothing has
    touched the GPU in this process yet (torch is not even imported)."""
    port = 29500 + os.getpid() % 2000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__), *sys.argv[1:]]
    sys.exit(subprocess.call(cmd, env=dict(os.environ, MASTER_ADDR="127.0.0.1")))


def run_sequence(args, d, ctx, poses, frame_scene, shard_mod, W, H, spp, depth, rank, world, dev, n_tris, mesh_name, hs_nodes_root_box):
    """BASELINE.json configs[4]: the whole pose file as one job (deep-space-ray-tracer_amd/sequence.py)."""
    import torch
    import torch.distributed as dist
    from dsrt_amd import sequence
    frames = [i for i in range(len(poses)) if not d.pose_to_frame(poses[i]).skipped]
    # whole frames are dealt by estimated cost (nearest poses cost 30 times the farthest), not round-robin: sequence.approach_cost
    nodes = hs_nodes_root_box
    radius = 0.5 * max(nodes[1][a] - nodes[0][a] for a in range(3))
    costs = [sequence.approach_cost(d.pose_to_frame(poses[i]).sep_m, radius) for i in frames] if args.deal == "cost" else None
    mine = sequence.frame_assignment(frames, rank, world, args.split, costs)
    shard = (rank, world, shard_mod.gather_to_root) if (world > 1 and args.split == "tiles") else None
    if args.batch > 0:
        return run_sequence_batched(args, d, ctx, frames, mine, frame_scene, W, H, spp, depth, rank, world, dev, n_tris, mesh_name, shard_mod, args.split == "tiles")
    pipe = sequence.FramePipeline(d, ctx, W, H, spp, depth, inflight=args.inflight, rng_mode=args.rng_mode, device=dev, shard=shard, tune=(0, 0, 0, args.tune3))

    def go(ids):
        for i in ids:
            fr, cam, _ = frame_scene(i)
            pipe.submit(i, cam, tuple(fr.sun_dir_model))
        pipe.drain()

    go([mine[0] if mine else frames[0]] * pipe.K)                     # warm-up of every slot (not timed)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    go(mine)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    pipe.close()
    if rank == 0:
        print(json.dumps({
            "metric": f"frames/s (pose sequence, {W}x{H}, scene resident, frames in flight on separate streams, images copied to pinned host memory)",
            "value": len(frames) / dt, "unit": "frames/s", "n_gpus": world, "steps": len(frames), "warmup": pipe.K,
            "ms_per_step": dt / len(frames) * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic", "msamples_per_s": len(frames) * W * H * spp / dt / 1e6,
            "config": {"workload": f"{mesh_name}: {n_tris} triangles, all {len(frames)} poses of rendezvous_1s_dt0_01s.txt, {W}x{H} @ {spp} spp, "
                                   f"max_depth {depth}, rng_mode {args.rng_mode}", "frames": len(frames), "spp": spp, "rng_mode": args.rng_mode, "bvh": args.bvh,
                       "frames_in_flight": pipe.K, "split": args.split if world > 1 else "single GPU",
                       "parallelism": ("poses dealt round-robin to ranks, no data-path collective" if args.split == "frames" else
                                       "every frame sharded by 8x8 tiles over all ranks, one RCCL gather per frame") if world > 1 else "one GPU",
                       "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")}}), flush=True)


def run_sequence_batched(args, d, ctx, frames, mine, frame_scene, W, H, spp, depth, rank, world, dev, n_tris, mesh_name, shard_mod, tiles):
    """configs[4] through dsrt_render_batch, launches of --batch frames each, nearest (costliest) poses first; two contexts on two streams
    take the launches in turn, so one launch's last chains run under the next launch's bulk and its images go to pinned host memory meanwhile.
    split "frames": this rank's poses (dealt by cost), no collective.  split "tiles" (N > 1): EVERY pose, this rank's interleaved tiles of each --
    a sharded batch launch -- then one gather per launch for all its frames and the de-interleave on rank 0: the ranks' loads are equal by
    construction and every frame's serial chains are spread over all GPUs as well as hidden under the other frames."""
    import torch
    import torch.distributed as dist
    sharded = tiles and world > 1
    desc = d.make_desc(W, H, spp, depth, rng_mode=args.rng_mode, tune=tuple(int(v) for v in args.tune012.split(":")) + (args.tune3,), shard_rank=rank if sharded else 0,
                       shard_count=world if sharded else 0)
    part = d.shard_layout(desc)["rgb8_bytes_padded"] if sharded else W * H * 3
    B = max(1, min(args.batch, ((1 << 32) - 1) // ((part // 3) * (16 if args.rng_mode == 1 else 1))))      # 32-bit work-item numbers inside a launch (include/dsrt.h)
    ids = sorted(mine, reverse=True)
    groups = [ids[k:k + B] for k in range(0, len(ids), B)]
    cams = {i: frame_scene(i) for i in ids}
    ctxs = [ctx, ctx.clone()]
    with torch.cuda.device(dev):
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    bufs = [torch.zeros(B * part, dtype=torch.uint8, device=dev) for _ in range(2)]
    root = rank == 0
    images = [torch.zeros(B * W * H * 3, dtype=torch.uint8, device=dev) for _ in range(2)] if sharded and root else bufs
    host = [torch.empty(B * W * H * 3, dtype=torch.uint8).pin_memory() for _ in range(2)] if root or not sharded else None

    def go(gs):
        for k, g in enumerate(gs):
            slot = k % 2
            with torch.cuda.stream(streams[slot]):
                ctxs[slot].render_batch(desc, [cams[i][1] for i in g], [tuple(cams[i][0].sun_dir_model) for i in g], bufs[slot].data_ptr(),
                                        stream=streams[slot].cuda_stream)
                if sharded:
                    flat = shard_mod.gather_to_root(bufs[slot][:len(g) * part], world, rank)      # the one collective of the launch
                    if root:
                        ctxs[slot].deinterleave_batch(desc, len(g), flat.data_ptr(), images[slot].data_ptr(), stream=streams[slot].cuda_stream)
                if host is not None:
                    n = len(g) * W * H * 3
                    host[slot][:n].copy_(images[slot][:n], non_blocking=True)
        for st in streams:
            st.synchronize()

    go([groups[-1][:2]] * 2 if groups else [])                        # warm-up of both slots (not timed)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    go(groups)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    check = None
    if sharded and root and groups:                                   # the last launch's first frame against a launch of its own
        g = groups[-1]
        whole = torch.zeros(W * H * 3, dtype=torch.uint8, device=dev)
        ctx.set_camera_sun(cams[g[0]][1], tuple(cams[g[0]][0].sun_dir_model))
        ctx.render(d.make_desc(W, H, spp, depth, rng_mode=args.rng_mode), whole.data_ptr(), stream=torch.cuda.current_stream().cuda_stream, want_stats=True)
        slot = (len(groups) - 1) % 2
        check = bool(torch.equal(whole, images[slot][:W * H * 3]))
    ctxs[1].close()
    if rank == 0:
        print(json.dumps({
            "metric": f"frames/s (pose sequence, {W}x{H}, scene resident, frames rendered as batch launches, images copied to pinned host memory)",
            "value": len(frames) / dt, "unit": "frames/s", "n_gpus": world, "steps": len(frames), "warmup": 4,
            "ms_per_step": dt / len(frames) * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic", "msamples_per_s": len(frames) * W * H * spp / dt / 1e6,
            "config": {"workload": f"{mesh_name}: {n_tris} triangles, all {len(frames)} poses of rendezvous_1s_dt0_01s.txt, {W}x{H} @ {spp} spp, "
                                   f"max_depth {depth}, rng_mode {args.rng_mode}", "frames": len(frames), "spp": spp, "rng_mode": args.rng_mode, "bvh": args.bvh,
                       "frames_per_launch": B, "launches_per_rank": len(groups), "split": ("tiles" if sharded else "frames") if world > 1 else "single GPU",
                       "frames_dealt_by": args.deal if not sharded else None, "reassembled_frame_equals_its_own_launch": check,
                       "parallelism": ("every rank renders its interleaved 8x8 tiles of every pose as one pool; one gather per launch" if sharded else
                                       "poses dealt to ranks by estimated cost, no data-path collective") if world > 1 else "one GPU"}}), flush=True)


def fixture_check(rgb, fixture_name, args, obj, W, H, spp, depth):
    """Compares an image of THE HEADLINE CONFIGURATION with the record the reference's own kernel left for exactly that frame in tests/golden/<fixture_name>
    (tests/golden/make_ref_gpu_fixtures.py: oracle/_ref/ref_gpu or ref_gpu_detmath, run once on an MI355X).  Data only: nothing of oracle/ is executed."""
    import hashlib
    path = os.path.join(ROOT, "tests", "golden", fixture_name)
    key = f"station/{args.tris}/frame{args.frame:02d}/{W}x{H}x{spp}"
    rec = {"file": "tests/golden/" + fixture_name, "key": key, "equal": None}
    if args.obj or args.bvh != "median" or depth != 50 or not os.path.exists(path):
        rec["why_not_compared"] = "not the fixture's configuration (procedural mesh, median tree, depth 50) or fixture file absent"
        return rec
    entry = json.load(open(path)).get("entries", {}).get(key)
    if entry is None:
        rec["why_not_compared"] = "the fixture file has no image of this configuration"
        return rec
    h = hashlib.sha256()
    with open(obj, "rb") as f:
        for block in iter(lambda: f.read(1 << 22), b""):
            h.update(block)
    if h.hexdigest() != entry["job"]["obj_sha256"]:
        rec["why_not_compared"] = "the mesh file differs from the one the fixture was rendered from"
        return rec
    mine = hashlib.sha256(rgb.tobytes()).hexdigest()
    rec.update({"equal": mine == entry["image"]["sha256"], "image_sha256": mine, "reference_kernel_image_sha256": entry["image"]["sha256"], "lit_pixels": entry["image"]["lit"],
                "pixels": W * H, "reference_kernel_ms_when_the_fixture_was_made": entry.get("reference_report", {}).get("gpu_render_scene_ms")})
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=1000)
    ap.add_argument("--depth", type=int, default=50)
    ap.add_argument("--frame", type=int, default=98)
    ap.add_argument("--tris", type=int, default=1000000)
    ap.add_argument("--obj", type=str, default="")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra single-GPU measurements (far frame 0; rng_mode 1; SAH tree; drop-in entry point)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 child passes (roofline fields that need counters become null)")
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    ap.add_argument("--stack-entries", type=int, default=0)
    ap.add_argument("--sequence", action="store_true", help="config 5: render every pose of the file once (default 250 spp) and report frames/s")
    ap.add_argument("--split", choices=["frames", "tiles"], default="tiles",
                    help="--sequence on N > 1 GPUs: tiles = every rank renders its interleaved tiles of EVERY pose as one pool, one gather per launch (default: "
                         "equal loads; projected from single-GPU shard probes, never yet measured on N GPUs: 5.8x at 8 ranks in rng_mode 0); frames = whole poses dealt to "
                         "ranks by estimated cost, no collective (projected 5.0x)")
    ap.add_argument("--rng-mode", type=int, default=0)
    ap.add_argument("--inflight", type=int, default=16, help="--sequence: frames in flight at once per GPU (separate streams; contexts share the scene)")
    ap.add_argument("--bvh", choices=["median", "sah", "lbvh"], default="median",
                    help="median = the reference's tree (parity; the headline). sah = non-parity fast mode (SURVEY.md 8(f) n4), labelled in the output")
    ap.add_argument("--single-process", action="store_true", help="N > 1: drive all GPUs from this process through dsrt_multi_* (library-side RCCL gather)")
    ap.add_argument("--deal", choices=["cost", "round-robin"], default="cost", help="--sequence on N GPUs, --split frames: how whole frames are dealt to ranks")
    ap.add_argument("--batch", type=int, default=99, help="--sequence: render the poses through dsrt_render_batch, this many frames per launch (0 = one launch per frame, --inflight of them overlapping)")
    ap.add_argument("--tune012", type=str, default="0:0:0", help="--sequence: DsrtRenderDesc.tune[0..2] = min_walk_iters:advance_budget:leaf_ratio4 (development aid)")
    ap.add_argument("--tune3", type=int, default=0, help="--sequence: DsrtRenderDesc.tune[3], the DSRT_TUNE_* switches of include/dsrt.h (development switches: env DSRT_EXPERIMENT)")
    ap.add_argument("--tree", choices=["certified", "reference"], default="certified",
                    help="certified (default): rays walk the certified second tree (dsrt_ctx_set_certified_tree: a SAH tree whose every answer the kernel certifies against the "
                         "reference's median-split tree, re-walking that tree when it cannot) -- the reference's bytes, checked against the reference kernel's own image in this "
                         "line; reference: every ray walks the reference tree (the plain walk; also timed as extras.reference_walk_only)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.tree == "certified" and args.bvh == "median":
        os.environ.setdefault("DSRT_CERTIFIED_TREE", "1")          # contexts the library creates itself (dsrt_multi_*, the drop-in gpu_render_scene) follow the same choice
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world == 1 and not args.single_process and not args.pmc_child:
        spawn_distributed(args.gpus)                                   # never returns
    # one hardware queue per frame in flight (the HIP runtime maps streams onto 4 by default); must be set before HIP starts
    os.environ.setdefault("GPU_MAX_HW_QUEUES", str(max(4, args.inflight)))

    import torch
    import torch.distributed as dist
    import dsrt_amd as d
    from dsrt_amd import meshgen
    from dsrt_amd import dist as shard_mod

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # DSRT_BENCH_REHEARSAL=1: run the N-rank flow with every rank on GPU 0 and gloo carrying host copies -- a functional rehearsal of
    # the multi-GPU path on a one-GPU box (timings mean nothing; rank 0 checks the reassembled image against a whole-frame render)
    rehearsal = world > 1 and os.environ.get("DSRT_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    if world > 1:
        if world != args.gpus:
            sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    def all_reduce(t, op=None):
        kw = {} if op is None else {"op": op}
        if rehearsal:
            h = t.cpu()
            dist.all_reduce(h, **kw)
            t.copy_(h)
        else:
            dist.all_reduce(t, **kw)
    n_gpus = args.gpus if args.single_process else world
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    stream = torch.cuda.current_stream().cuda_stream

    # ---- scene: mesh -> flatten -> BVH (host), upload + re-layout (device).  Not in the timed region; reported in `setup`. ----
    W, H, spp, depth = args.width, args.height, args.spp, args.depth
    if args.sequence and spp == 1000:
        spp = 250                                   # BASELINE.json configs[4]
    setup = {}
    if args.obj:
        obj, mesh_name = args.obj, os.path.basename(args.obj)
    else:
        obj = f"/tmp/dsrt_bench_station_v{meshgen.VERSION}_{args.tris}.obj"
        if rank == 0 and not os.path.exists(obj):
            tmp = obj + f".{os.getpid()}.tmp"
            meshgen.write_obj(meshgen.build_station(args.tris), tmp, mtl_name=os.path.basename(obj)[:-4] + ".mtl")
            os.replace(tmp, obj)
        if world > 1:
            dist.barrier()
        mesh_name = f"procedural ISS-like stand-in (meshgen.py v{meshgen.VERSION}), target {args.tris} triangles"
    t0 = time.perf_counter()
    hs = d.HostScene().add_obj(obj)
    hs.lbvh_device = local_rank
    setup["obj_parse_flatten_s"] = time.perf_counter() - t0
    if hs.texture_failures:
        setup["textures_not_decoded"] = hs.texture_failures[:8]     # the reference's stb_image would have decoded these: NOT the reference's picture
    t0 = time.perf_counter()
    hs.build_bvh(args.bvh)
    setup[f"bvh_build_{args.bvh}_host_s"] = time.perf_counter() - t0
    if args.bvh != "median":
        mesh_name += f" [NON-PARITY {args.bvh.upper()} BVH]"
    poses = d.read_pose_file(os.path.join(ROOT, "tests", "golden", "rendezvous_1s_dt0_01s.txt"))

    def frame_scene(idx):
        fr = d.pose_to_frame(poses[idx])
        cam = d.frame_camera(fr, 40.0, W, H, spp, depth)
        return fr, cam, hs.view(cam, tuple(fr.sun_dir_model))

    fr, cam, scene = frame_scene(args.frame)
    n_tris = scene.num_triangles

    if args.single_process and n_gpus > 1:
        multi = d.Multi(list(range(n_gpus)), frames_in_flight=args.inflight if args.sequence else 1)
        multi.upload(scene)
        desc = d.make_desc(W, H, spp, depth, rng_mode=args.rng_mode)
        if args.sequence:
            frames = [i for i in range(len(poses)) if not d.pose_to_frame(poses[i]).skipped]
            cams, suns = [], []
            for i in frames:
                f_i, c_i, _ = frame_scene(i)
                cams.append(c_i)
                suns.append(tuple(f_i.sun_dir_model))
            multi.render_sequence(desc, cams[:n_gpus * args.inflight], suns[:n_gpus * args.inflight], want_images=False)      # warm-up
            _, sec = multi.render_sequence(desc, cams, suns, want_images=False)
            print(json.dumps({"metric": "frames/s (pose sequence, one host process, every GPU its tiles of every pose as batch launches)", "value": len(frames) / sec, "unit": "frames/s",
                              "n_gpus": n_gpus, "steps": len(frames), "warmup": n_gpus * args.inflight, "ms_per_step": sec / len(frames) * 1e3, "higher_is_better": True,
                              "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                              "config": {"workload": f"{mesh_name}: {n_tris} triangles, {len(frames)} poses, {W}x{H} @ {spp} spp, rng_mode {args.rng_mode}",
                                         "parallelism": "dsrt_multi_render_sequence: sharded batch launches, one gather per launch"}}), flush=True)
            return
        for _ in range(args.warmup):
            multi.render_frame(desc, cam, tuple(fr.sun_dir_model))
        t0 = time.perf_counter()
        per_rank = []
        for _ in range(args.steps):
            _, ms, _ = multi.render_frame(desc, cam, tuple(fr.sun_dir_model))
            per_rank.append(ms)
        dt = time.perf_counter() - t0
        print(json.dumps({"metric": "Msamples/s (ISS-mesh frame, 1920x1080 @1000spp path-traced samples per second, whole job)", "value": W * H * spp * args.steps / dt / 1e6,
                          "unit": "Msamples/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
                          "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": f"{mesh_name}: {n_tris} triangles, pose frame {args.frame}, {W}x{H} @ {spp} spp, max_depth {depth}, rng_mode {args.rng_mode}",
                                     "parallelism": f"one host process, dsrt_multi_render_frame: 8x8 tiles interleaved over {n_gpus} GPUs, ncclGather to GPU 0 (RCCL: {multi.uses_rccl}), "
                                                    "image copied to host inside the step", "kernel_ms_per_rank_last_step": per_rank[-1]}}), flush=True)
        return

    ctx = d.Context(local_rank)
    certified = args.tree == "certified" and args.bvh == "median"
    if certified:
        ctx.set_certified_tree(True)
    t0 = time.perf_counter()
    ctx.upload(scene)
    torch.cuda.synchronize()
    setup["pack_and_upload_s"] = time.perf_counter() - t0
    if certified:
        setup["pack_and_upload_includes"] = "the SAH build of the certified second tree and both trees' re-layout"

    if args.pmc_child:                               # the run rocprofv3 watches: this workload's kernel, twice, nothing else
        part = torch.zeros(W * H * 3, dtype=torch.uint8, device=dev)
        dsc = d.make_desc(W, H, spp, depth, stack_entries=args.stack_entries, rng_mode=args.rng_mode)
        ctx.render(dsc, part.data_ptr(), stream=stream, want_stats=True)
        ctx.render(dsc, part.data_ptr(), stream=stream, want_stats=True)
        return

    if args.sequence:
        root = hs.arrays()["nodes"][0]
        root_box = ([float(v) for v in root["bbox_min"]], [float(v) for v in root["bbox_max"]])
        run_sequence(args, d, ctx, poses, frame_scene, shard_mod, W, H, spp, depth, rank, world, dev, n_tris, mesh_name, root_box)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    shard = n_gpus if n_gpus > 1 else 0
    desc = d.make_desc(W, H, spp, depth, shard_rank=rank if shard else 0, shard_count=shard, stack_entries=args.stack_entries)
    lay = d.shard_layout(desc)
    part = torch.zeros(lay["rgb8_bytes_padded"] if shard else W * H * 3, dtype=torch.uint8, device=dev)
    image = torch.zeros(W * H * 3, dtype=torch.uint8, device=dev) if shard and rank == 0 else part

    kernel_ms, last_stats = [], []
    step_desc = [desc]

    def step(collect=True):
        st = ctx.render(step_desc[0], part.data_ptr(), stream=stream, want_stats=True)
        if collect:
            kernel_ms.append(st.kernel_ms)
            last_stats[:] = [st]
        if shard:
            flat = shard_mod.gather_to_root(part, world, rank)          # the one collective of the step (RCCL gather)
            if rank == 0:
                ctx.deinterleave(desc, flat.data_ptr(), image.data_ptr(), stream=stream)

    def timed(steps, warmup):
        kernel_ms.clear()
        for _ in range(warmup):
            step(collect=False)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            all_reduce(t, dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    def all_ranks_can(local_action, what):
        """Runs `local_action` (no collective inside: a render that makes the library allocate this mode's buffers) on every rank and agrees on the
        outcome BEFORE anybody enters a collective: a rank that failed alone would leave the others waiting in the gather for ever."""
        try:
            local_action()
            mine_ok = 1.0
        except Exception as e:  # noqa: BLE001
            print(f"bench.py rank {rank}: {what}: {e}", file=sys.stderr, flush=True)
            mine_ok = 0.0
        if world == 1:
            return mine_ok == 1.0
        agreed = torch.tensor([mine_ok], dtype=torch.float64, device=dev)
        all_reduce(agreed, dist.ReduceOp.MIN)
        return float(agreed.item()) == 1.0

    if not all_ranks_can(lambda: ctx.render(desc, part.data_ptr(), stream=stream, want_stats=True), "first render of the headline configuration"):
        if world > 1:
            dist.destroy_process_group()
        sys.exit("bench.py: a rank could not render the headline configuration (see stderr); nothing was measured")
    dt = timed(args.steps, args.warmup)
    my_kernel_ms = sum(kernel_ms) / max(1, len(kernel_ms))
    headline_stats = list(last_stats)
    # the image of the last timed step, kept on the host for the parity check against the oracle's rows (after the timed region, before anything overwrites it)
    headline_image = image[:W * H * 3].cpu().numpy().reshape(H, W, 3) if rank == 0 and not args.no_cpu else None
    headline_fixture = fixture_check(headline_image, "ref_gpu_detmath_images.json", args, obj, W, H, spp, depth) if headline_image is not None else None
    rehearsal_report = None
    if rehearsal and rank == 0:                                     # (before the rng_mode 1 steps below overwrite `image`)
        whole = torch.zeros(W * H * 3, dtype=torch.uint8, device=dev)
        ctx.render(d.make_desc(W, H, spp, depth, stack_entries=args.stack_entries), whole.data_ptr(), stream=stream, want_stats=True)
        rehearsal_report = {"ranks_on_one_gpu": world, "backend": "gloo", "reassembled_image_equals_whole_frame_render": bool(torch.equal(whole, image))}
    # N > 1: the K steps once more as ONE sharded batch launch per rank (dsrt_render_batch: the rank's tiles of all K frames are one pool, so
    # one frame's serial chains run under the others), one gather and one de-interleave for the lot.  Reported beside the headline, which
    # stays one launch per step, each waited for.
    batched = None
    if shard:
        nb = max(2, args.steps)
        pbytes = lay["rgb8_bytes_padded"]
        bbuf = bimg = None
        try:                                                          # (allocation is the one step that can fail on one rank only: agree on it
            bbuf = torch.zeros(nb * pbytes, dtype=torch.uint8, device=dev)        #  before anybody enters a collective)
            bimg = torch.zeros(nb * W * H * 3, dtype=torch.uint8, device=dev) if rank == 0 else None
            mine_ok = 1.0
        except Exception:  # noqa: BLE001
            mine_ok = 0.0
        agreed = torch.tensor([mine_ok], dtype=torch.float64, device=dev)
        all_reduce(agreed, dist.ReduceOp.MIN)
        if float(agreed.item()) == 1.0:
            sun = tuple(fr.sun_dir_model)

            def batch_step():
                ctx.render_batch(desc, [cam] * nb, [sun] * nb, bbuf.data_ptr(), stream=stream)
                flat = shard_mod.gather_to_root(bbuf, world, rank)
                if rank == 0:
                    ctx.deinterleave_batch(desc, nb, flat.data_ptr(), bimg.data_ptr(), stream=stream)
            batch_step()
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            batch_step()
            torch.cuda.synchronize()
            dist.barrier()
            tb = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
            all_reduce(tb, dist.ReduceOp.MAX)
            per = float(tb.item()) / nb
            batched = {"frames_in_the_launch": nb, "ms_per_frame": per * 1e3, "Msamples/s": W * H * spp / per / 1e6,
                       "every_image_equals_the_step_image": bool(all(torch.equal(bimg[k * W * H * 3:(k + 1) * W * H * 3], image) for k in range(nb))) if rank == 0 else None}
        else:
            batched = {"error": "a rank could not allocate the batch buffers"}
        del bbuf, bimg
    # N > 1: the same sharded step with rng_mode 1 (a Philox sub-sequence per sample: the mode whose work units are samples, not
    # 1000-sample pixel chains, and therefore the one that can scale).  Reported beside the headline, never instead of it.
    mode1 = None
    if shard:
        step_desc[0] = d.make_desc(W, H, spp, depth, shard_rank=rank, shard_count=shard, stack_entries=args.stack_entries, rng_mode=1)
        # (rng_mode 1 makes the library allocate its integer-sum buffer: agree that every rank got it before the first gather of this leg)
        if all_ranks_can(lambda: ctx.render(step_desc[0], part.data_ptr(), stream=stream, want_stats=True), "first render in rng_mode 1"):
            dt1 = timed(args.steps, 1)
            mode1 = {"rng_mode": 1, "value": W * H * spp * args.steps / dt1 / 1e6, "unit": "Msamples/s", "ms_per_step": dt1 / args.steps * 1e3,
                     "rank0_kernel_ms": sum(kernel_ms) / max(1, len(kernel_ms)),
                     "note": "same frame, same tile sharding, gather and de-interleave inside the step; statistically equivalent image (DESIGN.md section 4)"}
        else:
            mode1 = {"error": "a rank could not render in rng_mode 1"}
        step_desc[0] = desc
        last_stats[:] = headline_stats
    # N > 1: what scales, in one object a reader needs no other document for.  The three ways this frame can be rendered on N GPUs (one launch per
    # step in the reference's stream; the same with a Philox stream per sample; K frames per launch), each against ITS OWN one-GPU time, measured
    # in this run: rank 0 renders the whole frame alone while the other ranks wait at the barrier.  Only measured figures go in here.
    scaling_detail = None
    if shard:
        one_gpu = {}
        if rank == 0:
            whole = torch.zeros(W * H * 3, dtype=torch.uint8, device=dev)

            def alone(action, reps=2):
                action()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    action()
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / reps * 1e3
            try:
                one_gpu["single_launch_rng_mode_0"] = alone(lambda: ctx.render(d.make_desc(W, H, spp, depth, stack_entries=args.stack_entries), whole.data_ptr(), stream=stream, want_stats=True))
                one_gpu["single_launch_rng_mode_1"] = alone(lambda: ctx.render(d.make_desc(W, H, spp, depth, stack_entries=args.stack_entries, rng_mode=1), whole.data_ptr(), stream=stream, want_stats=True))
                if batched and "ms_per_frame" in batched:
                    nb1 = batched["frames_in_the_launch"]
                    wholes = torch.zeros(nb1 * W * H * 3, dtype=torch.uint8, device=dev)
                    sun1 = tuple(fr.sun_dir_model)
                    one_gpu["batch_launch_rng_mode_0"] = alone(lambda: ctx.render_batch(d.make_desc(W, H, spp, depth), [cam] * nb1, [sun1] * nb1, wholes.data_ptr(), stream=stream, want_stats=True), reps=1) / nb1
                    del wholes
            except Exception as e:  # noqa: BLE001 -- local work only: the other ranks are at the barrier below either way
                one_gpu["error"] = str(e)[:200]
            del whole
        dist.barrier()
        if rank == 0:
            n_gpu = {"single_launch_rng_mode_0": dt / args.steps * 1e3,
                     "single_launch_rng_mode_1": mode1.get("ms_per_step") if mode1 else None,
                     "batch_launch_rng_mode_0": batched.get("ms_per_frame") if batched else None}
            scaling_detail = {"n_gpus": n_gpus, "unit": "ms per 1920x1080x1000 frame (wall clock, gather and de-interleave included for N GPUs)" if (W, H, spp) == (1920, 1080, 1000) else "ms per frame (wall clock)",
                              "one_gpu_ms_measured_in_this_run_on_rank_0": one_gpu, "n_gpu_ms": n_gpu,
                              "speedup_vs_1gpu": {k: (one_gpu[k] / n_gpu[k] if one_gpu.get(k) and n_gpu.get(k) else None) for k in n_gpu},
                              "frames_in_the_batch_launch": batched.get("frames_in_the_launch") if batched else None,
                              "note": "single_launch_rng_mode_0 is the headline (parity mode: one LCG stream per pixel, so a pixel is a serial chain of spp samples and one frame's speed-up "
                                      "is bound by its longest chain); rng_mode 1 (Philox stream per sample) and K frames per launch are the forms whose work units are small enough to scale"}
    tiles_total, tiles_culled = (last_stats[0].tiles_total, last_stats[0].tiles_culled) if last_stats else (0, 0)

    # ---- work counters of exactly this launch shape (untimed counting build), for Mrays/s and the algorithmic bytes ----
    cdesc = d.make_desc(W, H, spp, depth, shard_rank=desc.shard_rank, shard_count=desc.shard_count, collect_counters=1,
                        stack_entries=args.stack_entries)
    st = ctx.render(cdesc, part.data_ptr(), stream=stream, want_stats=True)
    pixels_mine = lay["tiles_this_shard"] * 64 if shard else W * H
    my_bytes = algorithmic_bytes(st, pixels_mine)
    tot = torch.tensor([float(st.rays), float(st.primary_hits), float(st.samples), float(my_bytes)], dtype=torch.float64, device=dev)
    if world > 1:
        all_reduce(tot)
    rays, primary_hits, samples_counted, _ = [float(v) for v in tot.tolist()]
    lane_slots = {"node_loop_active": st.internal_entered / max(1, st.node_slots), "leaf_loop_active": st.tri_tests / max(1, st.tri_slots),
                  "advance_active": st.adv_active / max(1, st.adv_slots), "node_loop_parked_at_leaf": st.idle_at_leaf / max(1, st.node_slots),
                  "node_loop_waiting_for_advance": st.idle_waiting / max(1, st.node_slots), "node_loop_out_of_work": st.idle_done / max(1, st.node_slots)}

    # ---- extras (single GPU only).  Reported, never the headline. ----
    extras = None
    if not args.no_extras and world == 1:
        def measure(frame_idx, rng_mode):
            frx, camx, _ = frame_scene(frame_idx)
            ctx.set_camera_sun(camx, tuple(frx.sun_dir_model))
            dx = d.make_desc(W, H, spp, depth, stack_entries=args.stack_entries, rng_mode=rng_mode)
            ctx.render(dx, part.data_ptr(), stream=stream, want_stats=True)                       # warm-up
            runs = [ctx.render(dx, part.data_ptr(), stream=stream, want_stats=True) for _ in range(2)]
            ms = min(r.kernel_ms for r in runs)
            cx = d.make_desc(W, H, spp, depth, stack_entries=args.stack_entries, rng_mode=rng_mode, collect_counters=1)
            sx = ctx.render(cx, part.data_ptr(), stream=stream, want_stats=True)
            culled = runs[0].tiles_culled / max(1, runs[0].tiles_total)
            rec = {"frame": frame_idx, "sep_m": round(frx.sep_m, 1), "rng_mode": rng_mode, "kernel_ms": ms, "Mrays/s": sx.rays / ms / 1e3,
                   "coverage": sx.primary_hits / max(1, sx.samples), "tiles_culled_frac": culled}
            # a frame whose tiles are mostly culled is not sampled at this rate: the figure is (samples the image stands for) / time
            rec["Msamples/s" if culled < 0.5 else "Msamples/s_nominal_(culling_rate:_most_tiles_are_proven_empty_and_never_sampled)"] = W * H * spp / ms / 1e3
            return rec
        extras = {"note": "kernel-only times (HIP events), same mesh and size as the headline; rng_mode 1 = rocRAND-compatible Philox stream per "
                          "(pixel, sample): statistically equivalent image, not bit-identical to the reference stream; bvh sah = binned-SAH tree "
                          "instead of the reference's median split (non-parity fast mode, SURVEY.md 8(f) n4)",
                  "runs": [dict(measure(0, 0), bvh=args.bvh), dict(measure(args.frame, 1), bvh=args.bvh), dict(measure(0, 1), bvh=args.bvh)]}
        if args.bvh == "median":
            ctx.set_certified_tree(False)                            # the other trees are timed as what they are: plain walks
            hs_sah = d.HostScene().add_obj(obj)
            t0 = time.perf_counter()
            hs_sah.build_bvh("sah")
            setup["bvh_build_sah_host_s"] = time.perf_counter() - t0
            ctx.upload(hs_sah.view(cam, tuple(fr.sun_dir_model)))
            extras["runs"] += [dict(measure(args.frame, 0), bvh="sah"), dict(measure(args.frame, 1), bvh="sah")]
            del hs_sah
            # the GPU-built linear BVH (csrc/bvh_lbvh.hip): what a per-frame rebuild costs there, and what its frames cost
            hs_l = d.HostScene().add_obj(obj)
            hs_l.lbvh_device = local_rank
            try:
                hs_l.build_bvh("lbvh")
                hs_l.build_bvh("lbvh")                             # (the first call pays hipCUB's one-time module load)
                setup["bvh_build_lbvh_gpu_kernels_ms"] = hs_l.lbvh_build_ms
                setup["bvh_build_lbvh_gpu_total_ms_incl_upload_and_copy_back"] = hs_l.lbvh_total_ms
                ctx.upload(hs_l.view(cam, tuple(fr.sun_dir_model)))
                extras["runs"] += [dict(measure(args.frame, 0), bvh="lbvh (built on the GPU)")]
            except d.DsrtError as e:
                extras["lbvh_error"] = str(e)[:200]
            del hs_l
            ctx.set_certified_tree(certified)
            ctx.upload(scene)                                       # back to the reference tree (and its certified second tree)
        ctx.set_camera_sun(cam, tuple(fr.sun_dir_model))
        # the reference's own three calls, end to end (src/main.cpp:405-428): build_gpu_scene (upload in the reference layouts) ->
        # gpu_render_scene (scene fetched back, re-laid-out, uploaded, rendered, PPM written) -> free_gpu_scene; once cold, once warm
        devs = d.GPUScene()
        sun3 = (C.c_float * 3)(*fr.sun_dir_model)
        cwd = os.getcwd()
        tmpd = tempfile.mkdtemp(prefix="dsrt_dropin_", dir="/tmp")
        os.chdir(tmpd)
        try:
            t0 = time.perf_counter()
            rc = d.lib.dsrt_build_gpu_scene(hs._h, C.byref(cam), sun3, C.byref(devs))
            torch.cuda.synchronize()
            t_build = time.perf_counter() - t0
            if rc == 0:
                times = []
                for _ in range(2):
                    t0 = time.perf_counter()
                    d.lib.gpu_render_scene(C.byref(devs), W, H)
                    times.append((time.perf_counter() - t0) * 1e3)
                extras["drop_in_gpu_render_scene"] = {"dsrt_build_gpu_scene_upload_ms": t_build * 1e3, "gpu_render_scene_first_call_ms": times[0],
                                                      "gpu_render_scene_second_call_same_scene_ms": times[1], "wrote_ppm": os.path.exists("image_gpu.ppm"),
                                                      "note": "whole call: scene check / re-layout, render of this frame, 6 MB copy-back, PPM write"}
                d.lib.dsrt_free_gpu_scene(C.byref(devs))
        finally:
            os.chdir(cwd)
            shutil.rmtree(tmpd, ignore_errors=True)
        # two reference points for reading the numbers above: what this board's HBM does on a plain
        # device-to-device copy, and the headline frame end to end into pinned host memory (render + 6 MB copy)
        try:
            # float4 grid-stride kernels, 2 GiB per buffer (8x the Infinity Cache): plain and non-temporal copy (read + write counted) over three grid sizes, read only, write only
            sweep = {f"{name}, {bpc} workgroups per CU": d.microbench_copy(2 << 30, bpc, 6, device=local_rank, mode=m)["GBps"] for m, name in ((0, "copy"), (3, "non-temporal copy")) for bpc in (8, 16, 32)}
            extras["hbm_copy_GBps_measured"] = max(sweep.values())
            extras["hbm_copy_GBps_sweep"] = sweep
            extras["hbm_GBps_read_only_write_only"] = [d.microbench_copy(2 << 30, 32, 6, device=local_rank, mode=m)["GBps"] for m in (1, 2)]
            extras["hbm_copy_how"] = ("dsrt_microbench_copy: float4 grid-stride kernels (4 independent 16-byte accesses in flight per lane), 256-thread workgroups, 2 GiB per buffer, 6 launches, "
                                      "bytes moved / HIP-event time; the best copy of the sweep is the figure.  The guide quotes 6.29 TB/s for a float4 copy; the boxes of this pool give "
                                      "4.5-5.1 TB/s to a copy in every shape tried (non-temporal a little above plain), 5.2 TB/s to a pure read and 4.0-4.7 to a pure write, and 4.8 TB/s "
                                      "to hipMemcpy device-to-device: the figure is this board's, not the kernel's")
        except d.DsrtError as e:
            extras["hbm_copy_GBps_measured"] = None
            extras["hbm_copy_how"] = str(e)[:160]
        pinned = torch.empty(W * H * 3, dtype=torch.uint8).pin_memory()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.render(desc, part.data_ptr(), stream=stream)
        pinned.copy_(part[:W * H * 3], non_blocking=True)
        torch.cuda.synchronize()
        extras["frame_to_pinned_host_ms"] = (time.perf_counter() - t0) * 1e3
        # The certificate, audited ray by ray (collect_counters = 3): at 32 samples per pixel every answer of the second tree is also walked on the reference tree and compared.
        if certified:
            try:
                aspp = min(spp, 32)
                frx, camx, _ = frame_scene(args.frame)
                ctx.set_camera_sun(d.frame_camera(frx, 40.0, W, H, aspp, depth), tuple(frx.sun_dir_model))
                sa = ctx.render(d.make_desc(W, H, aspp, depth, stack_entries=args.stack_entries, collect_counters=3), part.data_ptr(), stream=stream, want_stats=True)
                extras["certificate_audit"] = {"spp": aspp, "second_tree_answers_audited": int(sa.certificate_audited), "differing_from_the_reference_walk": int(sa.certificate_audit_mismatches),
                                               "certificate_fallbacks": int(sa.certificate_fallbacks), "rays": int(sa.rays),
                                               "what": "counting build: every answer of the second tree (hit or miss) is also walked on the reference tree and compared -- triangle and the bit "
                                                       "patterns of t, u, v; blocked-or-not for any-hit shadow rays"}
                ctx.set_camera_sun(cam, tuple(fr.sun_dir_model))
            except Exception as e:  # noqa: BLE001
                extras["certificate_audit"] = {"error": str(e)[:200]}
        # The same frame with every ray on the reference tree (DSRT_TUNE_REFERENCE_WALK): what the certified second tree buys, and the plain walk's own speed.
        if certified:
            try:
                dr = d.make_desc(W, H, spp, depth, stack_entries=args.stack_entries, tune=(0, 0, 0, 64))
                ctx.render(dr, part.data_ptr(), stream=stream, want_stats=True)
                kr = [ctx.render(dr, part.data_ptr(), stream=stream, want_stats=True).kernel_ms for _ in range(2)]
                same = bool(headline_image is not None and np.array_equal(part[:W * H * 3].cpu().numpy().reshape(H, W, 3), headline_image))
                cr = ctx.render(d.make_desc(W, H, spp, depth, stack_entries=args.stack_entries, tune=(0, 0, 0, 64), collect_counters=1), part.data_ptr(), stream=stream, want_stats=True)
                extras["reference_walk_only"] = {"kernel_ms": min(kr), "Msamples/s": W * H * spp / min(kr) / 1e3, "image_equals_the_headline_image": same if headline_image is not None else None,
                                                 "nodes_entered_per_ray": cr.nodes_entered / max(1, cr.rays), "tri_tests_per_ray": cr.tri_tests / max(1, cr.rays),
                                                 "algorithmic_bytes_per_launch": algorithmic_bytes(cr, W * H),
                                                 "what": "DSRT_TUNE_REFERENCE_WALK: every ray walks the reference's median-split tree, nothing else changed"}
            except Exception as e:  # noqa: BLE001
                extras["reference_walk_only"] = {"error": str(e)[:200]}
        # the headline frame four times over as ONE batch launch (dsrt_render_batch): what is left of the step when a frame's last chains run
        # under the next frame's bulk.  The headline itself stays one launch per step, each waited for.
        try:
            nb = 4
            batch_img = torch.zeros(nb * W * H * 3, dtype=torch.uint8, device=dev)
            bdesc = d.make_desc(W, H, spp, depth, stack_entries=args.stack_entries)
            sun = tuple(fr.sun_dir_model)
            ctx.render_batch(bdesc, [cam] * nb, [sun] * nb, batch_img.data_ptr(), stream=stream, want_stats=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ctx.render_batch(bdesc, [cam] * nb, [sun] * nb, batch_img.data_ptr(), stream=stream)
            torch.cuda.synchronize()
            per_frame = (time.perf_counter() - t0) * 1e3 / nb
            same = all(bool(torch.equal(batch_img[k * W * H * 3:(k + 1) * W * H * 3], part[:W * H * 3])) for k in range(nb))
            extras["headline_frame_x4_as_one_batch_launch"] = {"ms_per_frame": per_frame, "Msamples/s": W * H * spp / per_frame / 1e3,
                                                               "every_image_equals_the_single_launch_image": same}
            del batch_img
        except Exception as e:  # noqa: BLE001 -- an extra never stops the bench
            extras["headline_frame_x4_as_one_batch_launch"] = {"error": str(e)[:200]}
        # The reference-identical mode at the headline size.  math_mode 1 = the same kernels compiled with cosf / sinf / powf from the device math library: the bytes of
        # the reference's own kernel as built for this GPU (oracle/_ref/ref_gpu; tests/golden/ref_gpu_images.json holds its image of exactly this frame).
        try:
            d1 = d.make_desc(W, H, spp, depth, stack_entries=args.stack_entries, math_mode=1)
            ctx.render(d1, part.data_ptr(), stream=stream, want_stats=True)       # warm-up
            torch.cuda.synchronize()
            k1 = []
            t0 = time.perf_counter()
            for _ in range(max(2, args.steps)):
                k1.append(ctx.render(d1, part.data_ptr(), stream=stream, want_stats=True).kernel_ms)
            torch.cuda.synchronize()
            per = (time.perf_counter() - t0) / len(k1)
            extras["math_mode_1"] = {"ms_per_step": per * 1e3, "Msamples/s": W * H * spp / per / 1e6, "kernel_ms": sum(k1) / len(k1), "steps": len(k1),
                                     "what": "same frame, same launch path, the kernels compiled against the device math library: byte-identical to the reference's kernel as built "
                                             "for this GPU with contraction off (hipify-perl + hipcc; nvcc's default contraction and libdevice are not reproducible here)",
                                     "reference_kernel_fixture": fixture_check(part[:W * H * 3].cpu().numpy().reshape(H, W, 3), "ref_gpu_images.json", args, obj, W, H, spp, depth)}
        except Exception as e:  # noqa: BLE001 -- an extra never stops the bench
            extras["math_mode_1"] = {"error": str(e)[:200]}

    # ---- roofline: calibration kernels live, PMC counters of this workload from child passes ----
    roof = None
    if rank == 0:
        secs = my_kernel_ms * 1e-3
        roof = {"bound": "valu_issue", "achieved": None, "peak": None, "unit": "G wave-instructions/s", "frac": None, "traffic": None,
                "kernel": "dsrt_render_kernel", "kernel_ms": my_kernel_ms,
                "achieved_algorithmic_GBps": my_bytes / secs / 1e9 if secs > 0 else None, "algorithmic_bytes_per_launch": my_bytes,
                "algorithmic_over_hbm_peak": my_bytes / secs / 1e9 / HBM_PEAK_GBPS if secs > 0 else None,
                "algorithmic_over_hbm_peak_meaning": "SURVEY.md 8(d)'s own figure: algorithmic bytes / kernel time / 8 TB/s. It may exceed 1: the formula charges every re-read of a node or "
                                                     "triangle record, and the 118 MB scene is served from L2 / Infinity Cache; hbm.frac is the measured fraction",
                "note": "rank 0's launch. The kernel is bound by vector-ALU issue, second by the L1 request rate; HBM is nearly idle (the scene lives in L2 / "
                        "Infinity Cache). frac is against the chip's simple-op issue peak (2 cycles per wave64 instruction per SIMD); mix.valu_busy_estimate prices the "
                        "kernel's own instruction classes. achieved_algorithmic = SURVEY.md 8(d) bytes / kernel time counts every re-read the caches serve and is "
                        "therefore not a fraction of the HBM peak; hbm.frac (PMC, this run) is."}
        if n_gpus == 1:
            cal = {}
            try:
                g0 = d.microbench_gather(0, False, 64, 0, 2 << 20, 2000, device=local_rank)
                g1 = d.microbench_gather(0, False, 64, 0, 19 << 20, 2000, device=local_rank)
                cal = {"l1_requests16_per_s_G_ceiling": g0["Grecords_per_s"] * 4, "random_19MB_table_records_per_s_G": g1["Grecords_per_s"]}
                simds_here = torch.cuda.get_device_properties(dev).multi_processor_count * 4
                cal["issue_costs_cycles_per_instruction_per_simd"] = {
                    k: d.microbench_valu(k, 8, 20000, device=local_rank, simds=simds_here)["cycles_per_instruction_per_simd"]
                    for k in ("v_add_f32", "v_fma_f32", "v_pk_mul_f32", "v_cndmask_b32_e64(sgpr pair)", "v_cmp_lt_f32_e64(sgpr pair)", "v_max_f32", "v_rcp_f32")}
            except d.DsrtError as e:
                cal = {"error": str(e)[:160]}
            pmc = None
            if not args.no_pmc:
                wl = ["--width", str(W), "--height", str(H), "--spp", str(spp), "--depth", str(depth), "--frame", str(args.frame), "--tris", str(args.tris),
                      "--bvh", args.bvh, "--tree", args.tree, "--stack-entries", str(args.stack_entries)] + (["--obj", args.obj] if args.obj else [])
                pmc = pmc_counters(wl)
            if pmc:
                clk = pmc.get("GRBM_GUI_ACTIVE", 0.0) / 8.0 / secs if secs > 0 else 0.0          # Hz (sum over the 8 XCDs)
                props = torch.cuda.get_device_properties(dev)
                simds = props.multi_processor_count * 4
                if pmc.get("SQ_INSTS_VALU") and clk:
                    roof["achieved"] = pmc["SQ_INSTS_VALU"] / secs / 1e9
                    roof["peak"] = simds * clk / VALU_PEAK_CYCLES_PER_WAVE_INSTR / 1e9
                    roof["frac"] = roof["achieved"] / roof["peak"]
                    roof["peak_basis"] = ("1 wave64 VALU instruction per SIMD per 2 cycles (MI355X_MICROARCH.md, SIMD-32) x 1024 SIMDs x the shader clock of this launch "
                                          "(GRBM_GUI_ACTIVE / 8 / kernel time); the best this device was measured to do is issue_costs['v_add_f32'] cycles (simple ops), and "
                                          "most of this kernel's instructions are of the classes that cost twice that -- see mix")
                    roof["instructions_per_sample"] = pmc["SQ_INSTS_VALU"] / float(W * H * spp)       # wave-instructions per sample: the figure optimisation lowers while
                    roof["frac_reading"] = ("frac is an issue RATE: a change that renders the frame with fewer instructions (round 3: leaf dealing 241.7 -> 231.5 wave-"   # frac stands still
                                            "instructions per sample, the trimmed node visit -> 219) raises Msamples/s, not frac")
                    roof["valu_lane_occupancy"] = pmc.get("SQ_THREAD_CYCLES_VALU", 0.0) / (64.0 * pmc.get("SQ_ACTIVE_INST_VALU", 1.0))
                    roof["useful_lane_frac"] = roof["frac"] * roof["valu_lane_occupancy"]
                    roof["shader_clock_GHz"] = clk / 1e9
                    costs = cal.get("issue_costs_cycles_per_instruction_per_simd") or {}
                    if all(k in costs for k in ("v_add_f32", "v_pk_mul_f32", "v_rcp_f32")):
                        half = sum(costs[k] for k in ("v_pk_mul_f32", "v_cndmask_b32_e64(sgpr pair)", "v_max_f32", "v_cmp_lt_f32_e64(sgpr pair)") if k in costs)
                        half /= max(1, sum(1 for k in ("v_pk_mul_f32", "v_cndmask_b32_e64(sgpr pair)", "v_max_f32", "v_cmp_lt_f32_e64(sgpr pair)") if k in costs))
                        avg = (KERNEL_VALU_CLASS_SHARES["simple"] * costs["v_add_f32"] + KERNEL_VALU_CLASS_SHARES["half"] * half +
                               KERNEL_VALU_CLASS_SHARES["quarter"] * costs["v_rcp_f32"])
                        roof["mix"] = {"class_shares_static": KERNEL_VALU_CLASS_SHARES, "average_issue_cycles_per_instruction": avg,
                                       "valu_busy_estimate": pmc["SQ_INSTS_VALU"] * avg / (simds * clk * secs),
                                       "frac_of_measured_simple_op_rate": roof["achieved"] / (simds * clk / costs["v_add_f32"] / 1e9),
                                       "note": "valu_busy_estimate = issued instructions x the measured issue cost of their class / (SIMDs x cycles of the launch)"}
                if pmc.get("TCP_TOTAL_CACHE_ACCESSES_sum") and cal.get("l1_requests16_per_s_G_ceiling"):
                    ach = pmc["TCP_TOTAL_CACHE_ACCESSES_sum"] / secs / 1e9
                    roof["l1"] = {"achieved": ach, "peak": cal["l1_requests16_per_s_G_ceiling"], "unit": "G 16-byte L1 requests/s", "frac": ach / cal["l1_requests16_per_s_G_ceiling"],
                                  "l1_hit_rate": 1.0 - pmc.get("TCP_TCC_READ_REQ_sum", 0.0) / pmc["TCP_TOTAL_CACHE_ACCESSES_sum"],
                                  "l2_hit_rate": pmc.get("TCC_HIT_sum", 0.0) / max(1.0, pmc.get("TCC_HIT_sum", 0.0) + pmc.get("TCC_MISS_sum", 0.0))}
                if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
                    hbm_bytes = (2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0     # KiB counters; gfx950 FETCH_SIZE reads 1/2 (guide, section HBM)
                    roof["traffic"] = hbm_bytes
                    roof["hbm"] = {"achieved": hbm_bytes / secs / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": hbm_bytes / secs / 1e9 / HBM_PEAK_GBPS,
                                   "fetch_size_raw_KiB": pmc["FETCH_SIZE"], "write_size_KiB": pmc["WRITE_SIZE"]}
                roof["pmc"] = {k: pmc[k] for k in sorted(pmc)}
            roof["calibration"] = cal

    if rank == 0:
        total_samples = W * H * spp
        out = {
            "metric": "Msamples/s (ISS-mesh frame, 1920x1080 @1000spp path-traced samples per second, whole job)",
            "value": total_samples * args.steps / dt / 1e6,
            "unit": "Msamples/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "mrays_per_s": rays * args.steps / dt / 1e6,
            "config": {
                "workload": f"{mesh_name}: {n_tris} triangles, pose frame {args.frame} of rendezvous_1s_dt0_01s.txt (camera {fr.sep_m:.1f} m), "
                            f"{W}x{H} @ {spp} spp, max_depth {depth}, seed 1337, rng_mode 0 (reference LCG stream)",
                "mesh_triangles": n_tris, "frame": args.frame, "width": W, "height": H, "spp": spp, "max_depth": depth,
                "coverage": primary_hits / max(1.0, samples_counted), "rays_per_sample": rays / max(1.0, samples_counted),
                "tiles_total_rank0": int(tiles_total), "tiles_culled_rank0": int(tiles_culled),
                "parallelism": f"screen tiles 8x8 interleaved over {n_gpus} GPU(s)" + (", one RCCL gather + de-interleave per step" if shard else ""),
                "bvh": args.bvh, "bvh_stack_need": hs.stack_need, "lds_stack_entries": st.lds_stack_entries,
                "tree": ("certified second tree: rays walk a binned-SAH tree over the reachable triangles; the kernel certifies every answer against the reference's median-split tree "
                         "(no exact tie, inside the reference-leaf box's slab interval, no zero direction component) and re-walks the reference tree otherwise. The image is the "
                         "reference's, byte for byte: reference_kernel_fixture below compares THIS run's image with the reference kernel's own") if certified else
                        "reference tree only (every ray walks the reference's median-split tree)",
                "certified_tree_used_rank0": int(last_stats[0].certified_tree_used) if last_stats else None,
                "certificate_fallbacks_in_the_counting_launch": int(st.certificate_fallbacks), "nodes_entered_per_ray": st.nodes_entered / max(1, st.rays),
                "tri_tests_per_ray": st.tri_tests / max(1, st.rays),
                "math_mode": 0,
                "math_mode_note": "the headline runs in math_mode 0: cosf / sinf / powf from include/dsrt_detmath.h, the one image every machine (the CPU oracle included) reproduces; it "
                                  "equals the reference's kernel built with those three functions (reference_kernel_fixture below). The mode that equals the reference's kernel built "
                                  "with this GPU's own math library is math_mode 1: extras.math_mode_1 has its speed at this size",
                "strong_scaling_note": "rng_mode 0 keeps the reference's ONE LCG stream per pixel, so a pixel is a serial chain of spp samples; the slowest "
                                       "pixel of this frame needs about 0.4 s however many GPUs share the frame, which caps the speed-up near 2.5x "
                                       "(DESIGN.md section 5 has the per-rank times for 1/2/4/8 ranks and the rng_mode 1 column that does scale)",
            },
            "setup": setup,
            "lane_slots": lane_slots,
            "roofline": roof,
        }
        if extras:
            out["extras"] = extras
        if mode1:
            out["rng_mode_1_same_sharding"] = mode1
        if batched:
            out["steps_as_one_sharded_batch_launch"] = batched
        if scaling_detail:
            out["scaling_detail"] = scaling_detail
        if rehearsal_report:
            out["rehearsal"] = rehearsal_report
        if n_gpus == 1 and not args.no_cpu:
            out["cpu_baseline"], out["parity_rows"] = cpu_baseline(d, scene, W, H, spp, args.cpu_budget, gpu_rgb8=headline_image)
            out["reference_kernel_fixture"] = headline_fixture
            if not args.no_extras:
                out["cpu_baseline_book"] = book_baseline(obj, fr, host_cores(), os.path.join(ROOT, "gpurun_out", "timings_threads.tsv"))
                out["reference_kernel_on_this_gpu"] = reference_kernel_on_this_gpu(d, ctx, part, stream, obj, fr, args.frame, W, H, min(spp, 100), depth)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
