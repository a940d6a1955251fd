#!/bin/bash
# usage (GPU box): tools/pmc_sequence.sh <tag> [bench args] -> gpurun_out/pmcseq_<tag>/summary.jsonl
# PMC counters (one pass, no tracing) of the batch launches of `bench.py --sequence`: VALU issue and lanes live per issued instruction of the
# 99-frame pool, to set beside the single frame's (bench.py's own roofline.pmc).
set -o pipefail
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmcseq_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $out/p1 -- python3 $GRAFT_REPO_ROOT/bench.py --sequence "$@" > $out/p1.log 2>&1 || echo "pass failed" >> $out/fail.log
python3 - "$out" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
names = {}
for f in glob.glob(out + '/p1/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'dsrt_render_batch_kernel' in r['Kernel_Name']:
            agg[int(r['Dispatch_Id'])][r['Counter_Name']] += float(r['Counter_Value'])
            names[int(r['Dispatch_Id'])] = r['Kernel_Name'][:60]
with open(out + '/summary.jsonl', 'w') as g:
    for d in sorted(agg):
        c = agg[d]
        rec = {"dispatch": d, "kernel": names[d], "counters": dict(c)}
        if c.get("SQ_ACTIVE_INST_VALU"):
            rec["valu_lane_occupancy"] = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])
        if c.get("GRBM_GUI_ACTIVE"):
            rec["valu_instructions_per_simd_cycle"] = c["SQ_INSTS_VALU"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
            rec["frac_of_2_cycle_peak"] = rec["valu_instructions_per_simd_cycle"] * 2.0
        if c.get("SQ_WAVE_CYCLES"):
            rec["wait_frac"] = c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
        g.write(json.dumps(rec) + "\n")
print(open(out + '/summary.jsonl').read())
PY
