import os, sys, json
sys.path.insert(0, "/root/repo")
import torch
import dsrt_amd as d
from dsrt_amd import meshgen
ROOT="/root/repo"
obj = f"/tmp/dsrt_bench_station_v{meshgen.VERSION}_1000000.obj"
if not os.path.exists(obj): meshgen.write_obj(meshgen.build_station(1000000), obj)
hs = d.HostScene().add_obj(obj); hs.build_bvh("median")
poses = d.read_pose_file(os.path.join(ROOT, "tests", "golden", "rendezvous_1s_dt0_01s.txt"))
W,H,spp=1920,1080,250
ctx=d.Context(0)
stream = torch.cuda.current_stream().cuda_stream
buf = torch.zeros(W*H*3, dtype=torch.uint8, device="cuda")
out=[]
first=True
for i in range(len(poses)):
    fr=d.pose_to_frame(poses[i])
    if fr.skipped: continue
    cam=d.frame_camera(fr,40.0,W,H,spp,50)
    if first: ctx.upload(hs.view(cam, tuple(fr.sun_dir_model))); first=False
    else: ctx.set_camera_sun(cam, tuple(fr.sun_dir_model))
    row={"frame":i,"sep_m":fr.sep_m}
    for rng in (1,0):
        desc=d.make_desc(W,H,spp,50,rng_mode=rng)
        ctx.render(desc, buf.data_ptr(), stream=stream, want_stats=True)
        st=ctx.render(desc, buf.data_ptr(), stream=stream, want_stats=True)
        row[f"ms_rng{rng}"]=round(st.kernel_ms,2); row["tiles_culled"]=int(st.tiles_culled)
    out.append(row)
print(json.dumps(out))
