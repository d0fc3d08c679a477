#!/bin/bash
# one rocprofv3 --pmc pass of bench.py: tools/pmc_once.sh <tag> "<counters>" <bench args...>  ->  gpurun_out/visit_$R/${R}_pmc_<tag>.csv
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$PWD; R=${R:-r03}; OUT=$ROOT/gpurun_out/visit_$R; mkdir -p $OUT; export TMPDIR=/tmp
tag=$1; ctr=$2; shift; shift
d=$OUT/pmc_$tag; rm -rf $d; mkdir -p $d
( cd /tmp && timeout -k 5 ${PMC_TIMEOUT:-420} rocprofv3 --pmc $ctr --output-format csv -d $d -o p -- python $ROOT/bench.py "$@" --no-cpu-baseline --no-e2e --no-secondary > $d/bench.json 2> $d/err.log )
echo "== pmc $tag rc=$?"
f=$(find $d -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python $ROOT/tools/prof_summaries.py pmc "$f" "$OUT/${R}_pmc_${tag}.csv" || tail -5 $d/err.log
rm -rf $d
