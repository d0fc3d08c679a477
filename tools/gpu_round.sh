#!/bin/bash
# one GPU visit: tests, smoke, the three bench lines (configs 2, 4, 5); PROFILE_FULL=1 adds the kernel trace of each
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
if [ -z "$SKIP_TESTS" ]; then
  timeout 2400 python -m pytest tests -m gpu -x -q ${PYTEST_ARGS:-} 2>&1 | tail -15 > gpurun_out/pytest_gpu.log; cat gpurun_out/pytest_gpu.log
  timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3 | tee gpurun_out/smoke.log
fi
for cfg in ${CONFIGS:-2 4 5}; do
  timeout 1500 python bench.py --config $cfg ${BENCH_ARGS:-} 2> gpurun_out/bench_c$cfg.err | tee gpurun_out/bench_c$cfg.json
  tail -3 gpurun_out/bench_c$cfg.err | cut -c1-300
done
if [ -n "$PROFILE_FULL" ]; then
  for cfg in ${CONFIGS:-2 4 5}; do
    out=$R/gpurun_out/prof_c$cfg; rm -rf $out; mkdir -p $out
    ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o full -- \
        python $R/bench.py --config $cfg --no-cpu-baseline --no-e2e > $out/bench.json 2> $out/err.log )
    tail -2 $out/err.log | cut -c1-200; cut -c1-600 $out/bench.json
    for f in $(find $out -name "*kernel_stats.csv"); do head -8 $f | cut -c1-220; done
  done
fi
