#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/pytest_gpu.log; cat gpurun_out/pytest_gpu.log
out=$PWD/gpurun_out/prof_c4b; rm -rf $out; mkdir -p $out
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o full -- python $OLDPWD/bench.py --config 4 --no-cpu-baseline --no-e2e > $out/bench.json 2> $out/err.log )
cut -c1-260 $out/bench.json; for f in $(find $out -name "*kernel_stats.csv"); do head -7 $f | cut -c1-200; done
find $out -name "*kernel_trace.csv" -delete
timeout 300 python bench.py --config 5 --no-cpu-baseline --no-e2e 2>/dev/null | cut -c1-260
