#!/usr/bin/env python3
"""The vector-ALU issue ceiling of this device (dsrt_microbench_valu): one JSON line per configuration.  GPU box only.
  (default)      every kind x 1 / 2 / 4 / 8 waves per SIMD with all lanes live, then v_fma_f32 and v_pk_fma_f32 at 4 waves per SIMD with
                 partial lane masks (lower half, lower quarter, even lanes, 25 scattered lanes): does a half-empty wave issue faster?
  --pmc          the four configurations the PMC pass watches (tools/valu_pmc.sh): v_fma_f32 and v_pk_fma_f32 at 4 and 8 waves per SIMD
The figure that matters is cycles_per_instruction_per_simd, from the waves' own s_memtime stamps."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FULL = (1 << 64) - 1


def main():
    import dsrt_amd as d
    iters = 40000
    if "--pmc" in sys.argv:
        cfgs = [(k, w, FULL) for k in (0, 1) for w in (4, 8)]
    else:
        cfgs = [(k, w, FULL) for k in range(5) for w in (1, 2, 4, 8)]
        scattered = sum(1 << ((i * 37 + 11) & 63) for i in range(25))
        for mask in ((1 << 32) - 1, (1 << 16) - 1, 0x5555555555555555, scattered, 0xFFFFFFFF00000000, 1):
            cfgs += [(0, 4, mask), (1, 4, mask)]
    for kind, wps, mask in cfgs:
        print(json.dumps(d.microbench_valu(kind, wps, iters, mask)), flush=True)


if __name__ == "__main__":
    main()
