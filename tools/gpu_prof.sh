#!/bin/bash
# rocprofv3 kernel trace (csv) + PMC passes on a reduced batch; outputs under gpurun_out/prof_*
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
mkdir -p gpurun_out; export TMPDIR=/tmp
READS=${READS:-262144}
cd /tmp
rm -rf $R/gpurun_out/prof_trace; mkdir -p $R/gpurun_out/prof_trace
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_trace -o kt -- python $R/bench.py --reads $READS --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_trace/bench.json 2> $R/gpurun_out/prof_trace/err.log
tail -2 $R/gpurun_out/prof_trace/err.log
find $R/gpurun_out/prof_trace -name "*.csv" | head
for f in $(find $R/gpurun_out/prof_trace -name "*kernel_stats.csv"); do cat $f | cut -c1-220; done
rocprofv3 -L 2>/dev/null | grep -oE "(SQ_[A-Z_0-9]+|GRBM_[A-Z_]+|FETCH_SIZE|WRITE_SIZE|TCC_[A-Z_0-9]+|VALUBusy|VALUUtilization|MeanOccupancy[A-Za-z]*|OccupancyPercent)" | sort -u > $R/gpurun_out/counters_available.txt
wc -l $R/gpurun_out/counters_available.txt
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_IFETCH SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_COUNT"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/prof_pmc$i; mkdir -p $R/gpurun_out/prof_pmc$i
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/prof_pmc$i -o pmc -- python $R/bench.py --reads ${PMC_READS:-65536} --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/prof_pmc$i/bench.json 2> $R/gpurun_out/prof_pmc$i/err.log
  echo "== pass $i: $set"; tail -1 $R/gpurun_out/prof_pmc$i/err.log | cut -c1-200
  python - <<PY
import csv, glob, collections
for f in glob.glob("$R/gpurun_out/prof_pmc$i/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(float); n = collections.Counter()
    for row in csv.DictReader(open(f)):
        k = (row.get("Kernel_Name","")[:60], row.get("Counter_Name"))
        acc[k] += float(row.get("Counter_Value", 0)); n[k] += 1
    for k in sorted(acc):
        if "scan_reads" in k[0] or "merge" in k[0] or "build_peq" in k[0] or "pack" in k[0]:
            print("  %-62s %-28s sum=%.6g launches=%d" % (k[0], k[1], acc[k], n[k]))
PY
done
