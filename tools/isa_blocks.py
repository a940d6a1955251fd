#!/usr/bin/env python3
"""Development aid: per-basic-block instruction mix (VALU / SALU / VMEM / LDS) of one kernel in a hipcc -S listing.

usage: hipcc ... --cuda-device-only -S -o render.s csrc/render_kernel.hip ; tools/isa_blocks.py render.s 'ILi8ELb0ELb0ELb1ELi0' [--dump LBB9_15 LBB9_42]
"""
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = [i for i, l in enumerate(lines) if l.startswith("_ZN4dsrt18dsrt_render_kernel") and key in l and l.rstrip().split(";")[0].strip().endswith(":")][0]
    if "--dump" in sys.argv:
        a, b = sys.argv[sys.argv.index("--dump") + 1:][:2]
        on = False
        for l in lines[start:]:
            if l.startswith("." + a + ":"):
                on = True
            if l.startswith("." + b + ":"):
                break
            if on and not l.strip().startswith(";"):
                print(l)
        return
    blocks, cur = [], {"name": "entry", "ins": []}
    blocks.append(cur)
    for l in lines[start + 1:]:
        t = l.strip()
        if re.match(r"^\.LBB\S+:", t):
            cur = {"name": t.split(":")[0], "ins": []}
            blocks.append(cur)
            continue
        if not t or t[0] in ";.":
            continue
        cur["ins"].append(t)
        if t.startswith("s_endpgm"):
            break

    def kind(op):
        return "V" if op.startswith("v_") else "S" if op.startswith("s_") else "L" if op.startswith("ds_") else "M"
    tot = {}
    for b in blocks:
        cnt = {}
        for i in b["ins"]:
            k = kind(i.split()[0])
            cnt[k] = cnt.get(k, 0) + 1
            tot[k] = tot.get(k, 0) + 1
        br = [x.split()[0][2:] + "->" + x.split()[-1] for x in b["ins"] if x.startswith("s_cbranch") or x.startswith("s_branch")]
        print(b["name"], len(b["ins"]), cnt, br)
    print("total", tot)


if __name__ == "__main__":
    main()
