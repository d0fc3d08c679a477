#!/bin/bash
# PMC passes with hard timeouts (rocprofv3 --pmc aborted and hung once on this pool: never run it unbounded)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp; cd /tmp
i=0
for set in ${PMC_SETS:-"FETCH_SIZE" "WRITE_SIZE"}; do
  i=$((i+1)); out=$R/gpurun_out/pmc_$i; rm -rf $out; mkdir -p $out
  timeout -k 5 ${PMC_TIMEOUT:-70} rocprofv3 --pmc $(echo $set | tr ',' ' ') --output-format csv -d $out -o pmc -- \
      python $R/bench.py --reads ${PMC_READS:-16384} --steps 1 --warmup 0 --no-cpu-baseline > $out/bench.json 2> $out/err.log
  echo "== pass $i ($set) rc=$?"; tail -2 $out/err.log | cut -c1-160
  python - <<PY
import csv, glob, collections
for f in glob.glob("$out/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(float); n = collections.Counter()
    for row in csv.DictReader(open(f)):
        k = (row.get("Kernel_Name","")[:50], row.get("Counter_Name"))
        acc[k] += float(row.get("Counter_Value", 0)); n[k] += 1
    for k in sorted(acc):
        print("  %-52s %-22s sum=%.6g launches=%d" % (k[0], k[1], acc[k], n[k]))
PY
done
