#!/bin/bash
# rocprofv3 --pmc passes over build/lanepair_ubench (one set of SQ counters per pass): tools/lp_pmc.sh <units> <tag>
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$PWD; OUT=$ROOT/gpurun_out/lp; mkdir -p $OUT; export TMPDIR=/tmp
units=${1:-131072}; tag=${2:-a}
i=0
for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_INSTS_VMEM" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1)); d=/tmp/lp_pmc_$tag_$i; rm -rf $d; mkdir -p $d
  ( cd /tmp && timeout -k 5 200 rocprofv3 --pmc $ctr --output-format csv -d $d -o p -- $ROOT/build/lanepair_ubench --units $units --len 10000 --k 1280 --reps 1 > $d/out.log 2> $d/err.log )
  echo "== pass $i rc=$? ($ctr)"
  f=$(find $d -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 - "$f" <<'PY'
import csv,sys,collections
agg=collections.defaultdict(float); n=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k=(r['Kernel_Name'][:60], r['Counter_Name']); agg[k]+=float(r['Counter_Value']); n[k]+=1
for k in sorted(agg): print(k[0], k[1], 'dispatches', n[k], 'sum %.4g'%agg[k], 'per_dispatch %.4g'%(agg[k]/n[k]))
PY
  else tail -3 $d/err.log; fi
done 2>&1 | tee $OUT/pmc_$tag.log
