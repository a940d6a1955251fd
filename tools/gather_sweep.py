#!/usr/bin/env python3
"""Sweep of the gather calibration kernel (dsrt_microbench_gather): one JSON line per configuration.  GPU box only.
  (default)  modes x dependent x live lanes x VALU pad on the 19 MB table (the node array's size at 1 M triangles)
  --quick    modes x dependent only
  --tables   table size sweep (L1-resident ... beyond L2) for modes 0 and 1: which level bounds the gather, and the L1 request ceiling
  --valu     a VALU-bound configuration (tiny table, long dependent fma chains): wave-instructions per cycle per SIMD at 4 waves/SIMD"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import dsrt_amd as d
    cfgs = []
    if "--tables" in sys.argv:
        for kb in (16, 256, 2048, 4096, 19 * 1024, 64 * 1024, 512 * 1024):
            for mode in (0, 1):
                cfgs.append((mode, False, 64, 0, kb << 10, 2000))
        for kb in (16, 2048):
            cfgs.append((0, True, 27, 75, kb << 10, 1500))
    elif "--valu" in sys.argv:
        cfgs = [(0, False, 64, 4096, 16 << 10, 100), (0, False, 27, 4096, 16 << 10, 100),
                (0, True, 27, 80, 16 << 10, 1500), (0, True, 27, 80, 2 << 20, 1500), (0, True, 27, 80, 19 << 20, 1500),
                (0, True, 64, 80, 19 << 20, 1500), (1, True, 27, 80, 2 << 20, 1500), (1, True, 27, 80, 19 << 20, 1500)]
    else:
        quick = "--quick" in sys.argv
        only = [int(a.split("=")[1]) for a in sys.argv if a.startswith("--mode=")]
        for mode in (only or (0, 1, 2)):
            for dep in (False, True):
                for live in ((64,) if quick else (64, 27)):
                    for pad in ((0,) if quick else (0, 75)):
                        cfgs.append((mode, dep, live, pad, 19 << 20, 4000 if not dep else 1500))
    for mode, dep, live, pad, tbytes, iters in cfgs:
        r = d.microbench_gather(mode, dep, live, pad, tbytes, iters)
        r["requests16_per_s_G"] = r["Grecords_per_s"] * 4
        r["records_per_CU_ns"] = r["Grecords_per_s"] / 256
        print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
