#!/usr/bin/env python3
"""What a vector-ALU instruction costs to issue on this device (dsrt_microbench_valu): one JSON line per configuration.  GPU box only.
  (default)   every kind, 8 workgroups per CU, back to back (x32) and alternating with v_add_f32; the selects and compares also in pairs;
              v_fma_f32 at 1 / 2 / 4 waves per SIMD; v_fma_f32, v_pk_fma_f32 and the select at 4 waves per SIMD with partial lane masks
  --pmc       what the PMC pass watches (tools/valu_pmc.sh): every kind back to back, 8 workgroups per CU"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FULL = (1 << 64) - 1


def main():
    import dsrt_amd as d
    iters = 60000
    kinds = range(len(d.VALU_KINDS))
    if "--order" in sys.argv:            # does the order inside a wave matter?  4 and 8 workgroups per CU
        pick = [k for k in kinds if d.VALU_KINDS[k] in ("v_cndmask_b32_e64(sgpr pair)", "v_max_f32", "v_cmp_lt_f32_e64(sgpr pair)", "v_pk_mul_f32", "v_min3_f32")]
        cfgs = [(k, p, w, FULL) for k in pick for w in (4, 8) for p in (0, 1, 3, 4, 5) if not (d.VALU_KINDS[k].startswith("v_pk") and p >= 4)]
    elif "--pmc" in sys.argv:
        cfgs = [(k, 0, 8, FULL) for k in kinds]
    else:
        cfgs = [(k, p, 8, FULL) for k in kinds for p in (0, 1)]
        cfgs += [(k, 2, 8, FULL) for k in kinds if "cndmask" in d.VALU_KINDS[k] or "cmp" in d.VALU_KINDS[k]]
        cfgs += [(0, 0, w, FULL) for w in (1, 2, 4)]
        scattered = sum(1 << ((i * 37 + 11) & 63) for i in range(25))
        for mask in ((1 << 32) - 1, (1 << 16) - 1, 0x5555555555555555, scattered, 1):
            cfgs += [(0, 0, 4, mask), (1, 0, 4, mask), (7, 0, 4, mask)]
    for kind, pattern, wps, mask in cfgs:
        print(json.dumps(d.microbench_valu(kind, wps, iters, mask, pattern)), flush=True)


if __name__ == "__main__":
    main()
