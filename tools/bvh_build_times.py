#!/usr/bin/env python3
"""Development aid: host BVH build times on the 1 M-triangle stand-in mesh, first call in the process and repeated (a host that rebuilds per
frame, as the reference does, sees the repeated figure), with one builder thread and with the machine's."""
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import dsrt_amd as d
    from dsrt_amd import meshgen
    obj = f"/tmp/dsrt_bench_station_v{meshgen.VERSION}_1000000.obj"
    if not os.path.exists(obj):
        meshgen.write_obj(meshgen.build_station(1000000), obj)
    hs = d.HostScene().add_obj(obj)
    out = {"DSRT_BUILD_THREADS": os.environ.get("DSRT_BUILD_THREADS", "machine"), "cores": len(os.sched_getaffinity(0))}
    for kind in ("median", "sah"):
        times = []
        for _ in range(4):
            t = time.perf_counter()
            hs.build_bvh(kind)
            times.append(round(time.perf_counter() - t, 3))
        a = hs.arrays()
        out[kind] = {"first_s": times[0], "repeated_s": min(times[1:]), "nodes_sha1": hashlib.sha1(a["nodes"].tobytes()).hexdigest()[:12],
                     "indices_sha1": hashlib.sha1(a["idx"].tobytes()).hexdigest()[:12]}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        for threads in ("1", None):
            env = dict(os.environ)
            if threads:
                env["DSRT_BUILD_THREADS"] = threads
            else:
                env.pop("DSRT_BUILD_THREADS", None)
            subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, check=True)
