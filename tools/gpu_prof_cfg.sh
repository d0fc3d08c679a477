#!/bin/bash
# kernel trace of configs 4 and 5 (tools/bench_configs.py) -> gpurun_out/prof_cfg
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; rm -rf gpurun_out/prof_cfg; mkdir -p gpurun_out/prof_cfg
export TMPDIR=/tmp
R=$PWD
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cfg -o cfg -- python $R/tools/bench_configs.py ${CFG_ARGS:---n4 0 --n5 40000} > $R/gpurun_out/prof_cfg/out.json 2> $R/gpurun_out/prof_cfg/err.log )
tail -2 gpurun_out/prof_cfg/err.log | cut -c1-300; cut -c1-600 gpurun_out/prof_cfg/out.json
for f in $(find gpurun_out/prof_cfg -name "*kernel_stats.csv"); do head -14 $f | cut -c1-220; done
