#!/usr/bin/env python3
"""Interleaved A/B of two or more BUILDS of libdsrt_hip.so on one box (development aid): tools/ab_tune.py is run as a child process once per
library per round (DSRT_LIB selects the library, deep-space-ray-tracer_amd/capi.py), round-robin, and the medians of its kernel times are printed.
Devices differ by a few per cent and so do runs, so builds are only ever compared inside one run of this tool.

usage: tools/ab_lib.py --libs deep-space-ray-tracer_amd/libdsrt_hip.so variants_tmp/libdsrt_x.so [--rounds 3] -- 0:0 [more ab_tune arguments]"""
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    argv = sys.argv[1:]
    rest = argv[argv.index("--") + 1:] if "--" in argv else ["0:0"]
    head = argv[:argv.index("--")] if "--" in argv else argv
    libs = []
    rounds = 3
    i = 0
    while i < len(head):
        if head[i] == "--libs":
            i += 1
            while i < len(head) and not head[i].startswith("--"):
                libs.append(head[i]); i += 1
        elif head[i] == "--rounds":
            rounds = int(head[i + 1]); i += 2
        else:
            raise SystemExit(f"unknown argument {head[i]}")
    ms = {l: [] for l in libs}
    for r in range(rounds):
        for l in libs:
            env = dict(os.environ, DSRT_LIB=os.path.abspath(l))
            out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ab_tune.py"), "--reps", "1"] + rest, env=env, capture_output=True, text=True, timeout=600)
            for line in out.stdout.splitlines():
                if line.startswith("{"):
                    d = json.loads(line)
                    if "median_ms" in d:
                        ms[l].append(d["median_ms"])
            if out.returncode != 0:
                print(out.stderr[-2000:], file=sys.stderr)
                raise SystemExit(out.returncode)
            print(json.dumps({"round": r, "lib": l, "kernel_ms": ms[l][-1] if ms[l] else None}), flush=True)
    base = statistics.median(ms[libs[0]])
    for l in libs:
        m = statistics.median(ms[l])
        print(json.dumps({"lib": l, "kernel_ms_median": m, "runs": ms[l], "vs_first": m / base}))


if __name__ == "__main__":
    main()
