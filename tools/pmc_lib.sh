#!/bin/bash
# Development aid (GPU box): PMC counters of the render kernel for one library build.  usage: tools/pmc_lib.sh <lib.so> <tag> [ab_tune arguments]
# -> gpurun_out/pmc_<tag>.txt   (one line per counter, summed over the dispatches of the render kernel in one tools/ab_tune.py --reps 1 run)
lib=$(readlink -f $1); tag=$2; shift 2
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_INSTS_FLAT SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_WAIT_ANY SQ_BUSY_CYCLES"; do
  i=$((i+1))
  DSRT_LIB=$lib timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $out/p$i -- python3 $GRAFT_REPO_ROOT/tools/ab_tune.py "$@" --reps 1 > $out/p$i.log 2>&1
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob(sys.argv[1] + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'dsrt_render_kernel' in r['Kernel_Name']:
            tot[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
for k in sorted(tot): print(k, tot[k], "dispatch-rows", n[k])
PY
