#!/bin/bash
# usage (GPU box): tools/profile_bench.sh <tag>  -> gpurun_out/prof_<tag>/{stats,pmc*}/...  +  gpurun_out/prof_<tag>/summary.json
# Profiles the DEFAULT bench workload (python3 bench.py, 1 step, no CPU leg).  Kernel trace/stats and PMC counters are
# collected in separate rocprofv3 runs (never combined), FETCH_SIZE and WRITE_SIZE in separate passes (TCC slot limit).
set -o pipefail
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-extras --no-pmc"     # --no-pmc: bench.py must not start its own nested rocprofv3 passes from under the profiler
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- $B > $out/stats.log 2>&1 || echo "stats pass failed" >> $out/fail.log
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $set --output-format csv -d $out/pmc$i -- $B > $out/pmc$i.log 2>&1 || echo "pmc pass $i failed" >> $out/fail.log
done
python3 - "$out" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
res = {"kernels": {}, "pmc": {}}
for f in glob.glob(out + '/stats/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'dsrt_' in r['Name']:
            res["kernels"][r['Name']] = {"calls": int(r['Calls']), "avg_ns": float(r['AverageNs']), "total_ns": int(r['TotalDurationNs'])}
for f in glob.glob(out + '/pmc*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'dsrt_render_kernel' in r['Kernel_Name']:
            res["pmc"].setdefault(r['Kernel_Name'], collections.defaultdict(float))
            res["pmc"][r['Kernel_Name']][r['Counter_Name']] += float(r['Counter_Value'])
res["pmc"] = {k: dict(v) for k, v in res["pmc"].items()}
json.dump(res, open(out + '/summary.json', 'w'), indent=1)
print(json.dumps(res, indent=1))
PY
