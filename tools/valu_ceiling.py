#!/usr/bin/env python3
"""The vector-ALU issue ceiling of this device (dsrt_microbench_valu): one JSON line per configuration.  GPU box only.
  (default)      every kind at 4 and 8 waves per SIMD with all lanes live; v_fma_f32 also at 1 and 2; then v_fma_f32, v_pk_fma_f32 and the
                 select at 4 waves per SIMD with partial lane masks (lower half, a quarter, even lanes, 25 scattered lanes, one lane)
  --pmc          what the PMC pass watches (tools/valu_pmc.sh): every kind at 8 waves per SIMD, v_fma_f32 / v_pk_fma_f32 / the mix at 4"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FULL = (1 << 64) - 1


def main():
    import dsrt_amd as d
    iters = 100000
    kinds = range(len(d.VALU_KINDS))
    if "--pmc" in sys.argv:
        cfgs = [(k, 8, FULL) for k in kinds] + [(k, 4, FULL) for k in (0, 1, 14)]
    else:
        cfgs = [(k, w, FULL) for k in kinds for w in (4, 8)] + [(0, 1, FULL), (0, 2, FULL)]
        scattered = sum(1 << ((i * 37 + 11) & 63) for i in range(25))
        for mask in ((1 << 32) - 1, (1 << 16) - 1, 0x5555555555555555, scattered, 1):
            cfgs += [(0, 4, mask), (1, 4, mask), (7, 4, mask)]
    for kind, wps, mask in cfgs:
        print(json.dumps(d.microbench_valu(kind, wps, iters, mask)), flush=True)


if __name__ == "__main__":
    main()
