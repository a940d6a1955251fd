#!/bin/bash
# usage (GPU box): tools/gather_pmc2.sh <tag>: table-size sweep + VALU test of the calibration kernel, then one PMC pass over the table sweep
set -o pipefail
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/gather_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python3 $GRAFT_REPO_ROOT/tools/gather_sweep.py --tables > $out/tables.jsonl 2> $out/tables.err || echo "tables failed" >> $out/fail.log
python3 $GRAFT_REPO_ROOT/tools/gather_sweep.py --valu > $out/valu.jsonl 2> $out/valu.err || echo "valu failed" >> $out/fail.log
timeout -k 10 300 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE --output-format csv -d $out/pmc1 -- python3 $GRAFT_REPO_ROOT/tools/gather_sweep.py --tables > $out/pmc1.log 2>&1 || echo "pmc1 failed" >> $out/fail.log
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE SQ_THREAD_CYCLES_VALU --output-format csv -d $out/pmc2 -- python3 $GRAFT_REPO_ROOT/tools/gather_sweep.py --valu > $out/pmc2.log 2>&1 || echo "pmc2 failed" >> $out/fail.log
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(out + '/pmc*/**/*counter_collection.csv', recursive=True):
    tagp = f.split('/pmc')[1][0]
    for r in csv.DictReader(open(f)):
        if 'gather_kernel' in r['Kernel_Name']:
            agg['pmc' + tagp + ' ' + r['Kernel_Name'][40:62] + '#' + r['Dispatch_Id']][r['Counter_Name']] += float(r['Counter_Value'])
with open(out + '/summary.txt', 'w') as g:
    for k, v in sorted(agg.items(), key=lambda kv: (kv[0][:4], int(kv[0].split('#')[1]))):
        g.write(k + '  ' + '  '.join(f'{n}={x:.5g}' for n, x in sorted(v.items())) + '\n')
print(open(out + '/summary.txt').read())
PY
cat $out/tables.jsonl $out/valu.jsonl
