#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 200 python tools/soak.py 90 2 2> gpurun_out/soak2.err | tee gpurun_out/soak2.json; tail -3 gpurun_out/soak2.err
out=$R/gpurun_out/prof_len; rm -rf $out; mkdir -p $out
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o len -- python $R/tools/bench_read_length_cliff.py > $out/table.json 2> $out/err.log )
for f in $(find $out -name "*kernel_stats.csv"); do head -14 $f | cut -c1-170; done
find $out -name "*kernel_trace.csv" -delete
