#!/bin/bash
# usage: tools/pmc_run.sh <tag> <probe args...>   -- separate rocprofv3 --pmc passes over tools/perf_probe.py (GPU box only)
set -o pipefail
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout -k 10 280 rocprofv3 --pmc $set --output-format csv -d $out/p$i -- python3 $GRAFT_REPO_ROOT/tools/perf_probe.py "$@" > $out/p$i.log 2>&1 || echo "pass $i failed" >> $out/fail.log
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'dsrt_render' in r['Kernel_Name']:
            agg[r['Kernel_Name'][:70] + '#' + r['Dispatch_Id']][r['Counter_Name']] += float(r['Counter_Value'])
with open(out + '/summary.txt', 'w') as g:
    for k, v in sorted(agg.items()):
        g.write(k + '\n')
        for n, x in sorted(v.items()):
            g.write(f'   {n} = {x:.6g}\n')
print(open(out + '/summary.txt').read())
PY
