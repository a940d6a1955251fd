#!/usr/bin/env python3
"""Development aid: the VALU instruction classes of one kernel in a hipcc -S listing, per basic block and in total, priced with the issue
costs measured by dsrt_microbench_valu (tools/valu_ceiling.py, profiles/r03/valu_issue_costs.*): what the kernel's own instruction stream
allows the vector ALU to issue per cycle.

usage: hipcc ... --cuda-device-only -S -o render.s csrc/render_kernel.hip ; tools/isa_mix.py render.s 'ILi8ELb0ELb0ELb1ELi0' [--blocks LBB16_375 ...] [--loops]
--loops groups the blocks by the innermost loop the compiler's comments put them in (`in Loop: Header=...`), which is how the node loop, the plain leaf loop
and the advance loop are told apart without knowing this build's label numbers.
Classes (cycles per wave64 instruction per SIMD, 8 waves per SIMD, PMC):
  simple  2.4   v_add/sub/mul/fma/fmac/mov/and/or/xor/shift/add_u32 ...
  half    4.2   v_pk_*, v_cmp*, v_cndmask*, v_min*/v_max*/v_med3, v_bfi, 64-bit integer
  quarter 8.5   v_rcp/v_sqrt/v_rsq/v_div_* helpers, f64, v_mul_lo/hi_u32, v_mad_u64
(the table below is corrected from the measurements when they are in)"""
import re
import sys

COST = {"simple": 2.4, "half": 4.2, "quarter": 8.5}
HALF = ("v_pk_", "v_cmp", "v_cndmask", "v_min", "v_max", "v_med3", "v_bfi", "v_lshlrev_b64", "v_lshrrev_b64", "v_lshl_add_u64", "v_mov_b64", "v_readlane", "v_writelane",
        "v_readfirstlane", "v_mbcnt")
QUARTER = ("v_rcp", "v_sqrt", "v_rsq", "v_div_", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64", "v_exp", "v_log", "v_sin", "v_cos", "v_cvt_f64", "v_cvt_f32_f64", "v_ldexp", "v_frexp",
           "v_trunc_f64", "v_floor_f64", "v_rndne_f64")


def klass(op):
    if op.endswith("_f64") or "_f64_" in op:
        return "quarter"
    if op.startswith(QUARTER):
        return "quarter"
    if op.startswith(HALF):
        return "half"
    return "simple"


def main():
    path, key = sys.argv[1], sys.argv[2]
    only = sys.argv[sys.argv.index("--blocks") + 1:] if "--blocks" in sys.argv else None
    lines = open(path).read().split("\n")
    start = [i for i, l in enumerate(lines) if l.startswith("_ZN4dsrt") and key in l and l.rstrip().split(";")[0].strip().endswith(":")][0]
    name, tot, per_block, loop_of, per_loop = "entry", {}, {}, {"entry": "(no loop)"}, {}
    for l in lines[start + 1:]:
        t = l.strip()
        if re.match(r"^\.LBB\S+:", t):
            name = t.split(":")[0].lstrip(".")
            m = re.search(r"Header=(\S+) Depth=(\d+)", t)
            if "Loop Header: Depth=" in t or "Inner Loop Header" in t:
                d = re.search(r"Depth=(\d+)", t).group(1)
                loop_of[name] = f"{name[1:]} depth {d}"
            else:
                loop_of[name] = f"{m.group(1)} depth {m.group(2)}" if m else "(no loop)"
            continue
        if not t or t[0] in ";.":
            continue
        op = t.split()[0]
        if op.startswith("s_endpgm"):
            break
        if not op.startswith("v_"):
            continue
        k = klass(op)
        if only is None or name in only:
            tot[k] = tot.get(k, 0) + 1
        per_block.setdefault(name, {}).setdefault(k, 0)
        per_block[name][k] += 1
        lp = per_loop.setdefault(loop_of.get(name, "(no loop)"), {})
        lp[k] = lp.get(k, 0) + 1
    if only:
        for b in only:
            c = per_block.get(b, {})
            n = sum(c.values())
            print(b, c, "VALU", n, "cycles", round(sum(COST[k] * v for k, v in c.items()), 1))
    if "--loops" in sys.argv:
        for lp, c in sorted(per_loop.items(), key=lambda kv: -sum(kv[1].values())):
            n = sum(c.values())
            print(f"{lp:24s} VALU {n:5d}  simple {c.get('simple', 0):4d}  half {c.get('half', 0):4d}  quarter {c.get('quarter', 0):4d}  issue cycles {sum(COST[k] * v for k, v in c.items()):8.1f}")
    n = sum(tot.values())
    cyc = sum(COST[k] * v for k, v in tot.items())
    print("total" if only is None else "selected blocks", tot, "VALU", n, "issue cycles", round(cyc, 1), "average", round(cyc / max(1, n), 3), "cycles per instruction")


if __name__ == "__main__":
    main()
