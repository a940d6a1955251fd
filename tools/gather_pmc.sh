#!/bin/bash
# usage (GPU box): tools/gather_pmc.sh <tag> -> gpurun_out/gather_<tag>/{sweep.jsonl, pmc*/, summary.txt}
# The gather calibration kernel on its own, then under separate rocprofv3 --pmc passes (never combined with tracing).
set -o pipefail
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/gather_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python3 $GRAFT_REPO_ROOT/tools/gather_sweep.py > $out/sweep.jsonl 2> $out/sweep.err || echo "sweep failed" >> $out/fail.log
i=0
for set in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $out/pmc$i -- python3 $GRAFT_REPO_ROOT/tools/gather_sweep.py --quick > $out/pmc$i.log 2>&1 || echo "pmc pass $i failed" >> $out/fail.log
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(out + '/pmc*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'gather_kernel' in r['Kernel_Name']:
            agg[r['Kernel_Name'][:60] + '#' + r['Dispatch_Id']][r['Counter_Name']] += float(r['Counter_Value'])
with open(out + '/summary.txt', 'w') as g:
    for k, v in sorted(agg.items(), key=lambda kv: int(kv[0].split('#')[1])):
        g.write(k + '\n')
        for n, x in sorted(v.items()):
            g.write(f'   {n} = {x:.6g}\n')
print(open(out + '/summary.txt').read())
PY
cat $out/sweep.jsonl
