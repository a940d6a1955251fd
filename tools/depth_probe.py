import os, sys, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import dsrt_amd as d
from dsrt_amd import meshgen
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
obj = f"/tmp/dsrt_bench_station_v{meshgen.VERSION}_1000000.obj"
if not os.path.exists(obj):
    meshgen.write_obj(meshgen.build_station(1000000), obj)
hs = d.HostScene().add_obj(obj); hs.build_bvh()
poses = d.read_pose_file(os.path.join(ROOT, "tests", "golden", "rendezvous_1s_dt0_01s.txt"))
fr = d.pose_to_frame(poses[98])
W, H, spp = 1920, 1080, 50
cam = d.frame_camera(fr, 40.0, W, H, spp, 50)
ctx = d.Context(0); ctx.upload(hs.view(cam, tuple(fr.sun_dir_model)))
rgb, _, st = ctx.render_to_host(d.make_desc(W, H, spp, 50, collect_counters=1))
n = st.internal_entered
print(json.dumps({"internal_visits": n, "lt6": st.visits_depth_lt6 / n, "lt9": st.visits_depth_lt9 / n, "lt12": st.visits_depth_lt12 / n, "rays": st.rays, "nodes_per_ray": n / st.rays}))
