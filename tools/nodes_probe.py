import os, sys, json
sys.path.insert(0, os.getcwd())
import torch, dsrt_amd as d
from dsrt_amd import meshgen
obj = f"/tmp/dsrt_bench_station_v{meshgen.VERSION}_1000000.obj"
if not os.path.exists(obj): meshgen.write_obj(meshgen.build_station(1000000), obj)
hs = d.HostScene().add_obj(obj); hs.build_bvh()
poses = d.read_pose_file("tests/golden/rendezvous_1s_dt0_01s.txt"); fr = d.pose_to_frame(poses[98])
W,H,spp=1920,1080,16
cam = d.frame_camera(fr, 40.0, W, H, spp, 50)
ctx = d.Context(0).set_certified_tree(True); ctx.upload(hs.view(cam, tuple(fr.sun_dir_model)))
buf = torch.zeros(W*H*3, dtype=torch.uint8, device="cuda"); s = torch.cuda.current_stream().cuda_stream
for walk in (0, 64):
    st = ctx.render(d.make_desc(W,H,spp,50,collect_counters=1,tune=(0,0,0,walk)), buf.data_ptr(), stream=s, want_stats=True)
    print(json.dumps({"lib": os.environ.get("DSRT_LIB","tree"), "reference_walk": walk==64, "nodes_per_ray": st.nodes_entered/st.rays, "internal_per_ray": st.internal_entered/st.rays, "tri_tests_per_ray": st.tri_tests/st.rays, "fallbacks": st.certificate_fallbacks, "node_active": st.internal_entered/max(1,st.node_slots)}))
