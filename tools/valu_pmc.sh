#!/bin/bash
# usage (GPU box): tools/valu_pmc.sh <tag> -> gpurun_out/valu_<tag>/{sweep.jsonl, pmc/, summary.txt}
# The VALU issue calibration kernel on its own (all kinds, waves per SIMD, lane masks), then one rocprofv3 --pmc pass (counters only, no
# tracing) over v_fma_f32 / v_pk_fma_f32 at 4 and 8 waves per SIMD: SQ_INSTS_VALU / (GRBM_GUI_ACTIVE / 8) / SIMDs = instructions per cycle per SIMD.
set -o pipefail
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/valu_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python3 $GRAFT_REPO_ROOT/tools/valu_ceiling.py > $out/sweep.jsonl 2> $out/sweep.err || echo "sweep failed" >> $out/fail.log
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc -- python3 $GRAFT_REPO_ROOT/tools/valu_ceiling.py --pmc > $out/pmc.log 2>&1 || echo "pmc pass failed" >> $out/fail.log
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(out + '/pmc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'valu_kernel' in r['Kernel_Name']:
            agg[r['Kernel_Name'][:60] + '#' + r['Dispatch_Id']][r['Counter_Name']] += float(r['Counter_Value'])
with open(out + '/summary.txt', 'w') as g:
    for k, v in sorted(agg.items(), key=lambda kv: int(kv[0].split('#')[1])):
        g.write(k + '\n')
        for n, x in sorted(v.items()):
            g.write(f'   {n} = {x:.6g}\n')
        if v.get('GRBM_GUI_ACTIVE') and v.get('SQ_INSTS_VALU'):
            g.write(f"   => cycles per VALU wave-instruction per SIMD (1024 SIMDs) = {v['GRBM_GUI_ACTIVE'] / 8 * 1024 / v['SQ_INSTS_VALU']:.3f}\n")
print(open(out + '/summary.txt').read())
PY
cat $out/sweep.jsonl
