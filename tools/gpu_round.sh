#!/bin/bash
# one GPU visit: tests, smoke, bench, kernel-trace profile
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/pytest_gpu.log; cat gpurun_out/pytest_gpu.log
python __graft_entry__.py smoke 2>&1 | tail -3 | tee gpurun_out/smoke.log
python bench.py ${BENCH_ARGS:-} 2> gpurun_out/bench.err | tee gpurun_out/bench.json
tail -5 gpurun_out/bench.err
if [ -n "$PROFILE_FULL" ]; then
  # the bench command itself (default flags) under the kernel tracer
  rm -rf gpurun_out/prof_full && mkdir -p gpurun_out/prof_full
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/prof_full -o full -- python $OLDPWD/bench.py > $OLDPWD/gpurun_out/prof_full/bench.json 2> $OLDPWD/gpurun_out/prof_full/err.log )
  tail -2 gpurun_out/prof_full/err.log | cut -c1-200; cut -c1-400 gpurun_out/prof_full/bench.json
  for f in $(find gpurun_out/prof_full -name "*kernel_stats.csv"); do head -6 $f | cut -c1-200; done
fi
if [ -n "$PROFILE" ]; then
  rm -rf gpurun_out/prof && mkdir -p gpurun_out/prof
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/prof -o r01 -- python $OLDPWD/bench.py --reads 262144 --steps 2 --warmup 1 --no-cpu-baseline > $OLDPWD/gpurun_out/prof_bench.json 2> $OLDPWD/gpurun_out/prof.err )
  tail -3 gpurun_out/prof.err; cat gpurun_out/prof_bench.json
  find gpurun_out/prof -name "*stats*" | head; 
  for f in $(find gpurun_out/prof -name "*kernel_stats.csv"); do head -12 $f; done
fi
