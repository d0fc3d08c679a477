#!/bin/bash
# ONE parameterised GPU visit (replaces the gpu_quick* / gpu_final / gpu_last / ... family of rounds 1-2).
#   gpurun --timeout 1500 -- 'R=r03 bash tools/gpu_visit.sh tests bench trace traffic'
# Stages run in the order given; every GPU command runs under `timeout`.  Output: gpurun_out/visit_$R/ (copy what
# should be judged into profiles/).  Stages:
#   tests        pytest -m gpu (PYTEST_ARGS) + smoke
#   bench        the default bench.py line (config 2 + `secondary` configs 4 / 5), BENCH_ARGS appended
#   bench4|bench5  one config on its own (--config N)
#   parity       whole-batch config-2 results for tools/full_parity_c2.py compare (gpurun_out/c2gpu.npz)
#   trace        rocprofv3 --kernel-trace --stats of bench.py for configs 2, 4, 5 (TRACE_CFGS)
#   traffic      rocprofv3 --pmc TCC_EA0_RDREQ_sum (= FETCH_SIZE's source; FETCH_COUNTER overrides) / WRITE_SIZE passes (separate runs) for configs 2, 4, 5 at full size
#   sq           SQ instruction / cycle counters for config 2 (262,144 reads) and config 4 (20,000 units)
#   cliff        tools/bench_read_length_cliff.py (CLIFF_ARGS)
#   latency      build/latency for the engine and the reference
#   short|path|hwlong|wide   tools/bench_short_pairs.py, bench_path.py, bench_hw_long.py, bench_wide.py
#   soak         tools/soak.py for SOAK_SECONDS (default 120)
#   chrom        tools/bench_chromosome.py (CHROM_ARGS; the reference's 1 Mb Chromosome pairs, answers against the fixture)
#   chromtrace   rocprofv3 --kernel-trace --stats of the seven chromosome distance + path calls
#   chromtraffic rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of one chromosome distance call (99 %)
#   ab           headline step of this build and of AB_LIB (default build/ab/libedlib_r02.so), alternating, on one box
#   cmd          run $CMD
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$PWD; R=${R:-r03}; OUT=$ROOT/gpurun_out/visit_$R; mkdir -p $OUT; export TMPDIR=/tmp
say() { echo "==== $* ($(date +%T))"; }
trace() {   # $1 = tag, rest = bench args
  local tag=$1; shift; local d=$OUT/trace_$tag; rm -rf $d; mkdir -p $d
  ( cd /tmp && timeout -k 5 ${TRACE_TIMEOUT:-300} rocprofv3 --kernel-trace --stats --output-format csv -d $d -o t -- python $ROOT/bench.py "$@" > $d/bench.json 2> $d/err.log )
  cp $d/bench.json $OUT/${R}_${tag}_underprofiler.json
  local f=$(find $d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${R}_${tag}_kernel_stats.csv
  f=$(find $d -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python $ROOT/tools/prof_summaries.py trace "$f" "$OUT/${R}_${tag}_kernel_trace_edlib.csv"
  head -8 $OUT/${R}_${tag}_kernel_stats.csv | cut -c1-170
  rm -rf $d
}
pmc() {     # $1 = tag, $2 = counters (space separated), rest = bench args
  local tag=$1 ctr=$2; shift; shift; local d=$OUT/pmc_$tag; rm -rf $d; mkdir -p $d
  ( cd /tmp && timeout -k 5 ${PMC_TIMEOUT:-240} rocprofv3 --pmc $ctr --output-format csv -d $d -o p -- python $ROOT/bench.py "$@" --no-cpu-baseline --no-e2e --no-secondary > $d/bench.json 2> $d/err.log )
  echo "== pmc $tag rc=$?"
  local f=$(find $d -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $ROOT/tools/prof_summaries.py pmc "$f" "$OUT/${R}_pmc_${tag}.csv"
  rm -rf $d
}
for stage in "$@"; do
  say $stage
  case $stage in
    tests)
      timeout 2700 python -m pytest tests -m gpu -x -q ${PYTEST_ARGS:-} 2>&1 | tail -15 | tee $OUT/${R}_pytest_gpu.log
      timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3 | tee $OUT/${R}_smoke.log ;;
    bench)
      timeout 1500 python bench.py ${BENCH_ARGS:-} 2> $OUT/bench.err | tee $OUT/${R}_bench_default.json | cut -c1-3000; tail -3 $OUT/bench.err | cut -c1-300 ;;
    bench4|bench5)
      c=${stage#bench}
      timeout 900 python bench.py --config $c ${BENCH_ARGS:-} 2> $OUT/bench_c$c.err | tee $OUT/${R}_bench_c$c.json | cut -c1-3000; tail -3 $OUT/bench_c$c.err | cut -c1-300 ;;
    parity)
      timeout 600 python tools/full_parity_c2.py gpu --out gpurun_out/c2gpu.npz 2>&1 | tail -3 ;;
    trace)
      for c in ${TRACE_CFGS:-2 4 5}; do trace bench_c$c --config $c --no-cpu-baseline --no-e2e --no-secondary; done ;;
    traffic)
      for c in ${TRAFFIC_CFGS:-2 4 5}; do
        pmc fetch_c$c "${FETCH_COUNTER:-TCC_EA0_RDREQ_sum}" --config $c --steps 1 --warmup 0
        pmc write_c$c "WRITE_SIZE" --config $c --steps 1 --warmup 0
      done
      python tools/make_traffic_json.py $OUT $R "$(cat $ROOT/.visit_commit 2>/dev/null)" | tee $OUT/${R}_traffic_summary.json ;;
    sq)
      pmc sq1_c2_262k "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" --reads 262144 --steps 1 --warmup 0
      pmc sq2_c2_262k "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" --reads 262144 --steps 1 --warmup 0
      pmc sq1_c4_20k "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH" --config 4 --units 20000 --steps 1 --warmup 0
      pmc sq2_c4_20k "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" --config 4 --units 20000 --steps 1 --warmup 0 ;;
    cliff)   timeout 1500 python tools/bench_read_length_cliff.py ${CLIFF_ARGS:-} 2>&1 | tee $OUT/${R}_read_length.json | cut -c1-2500 ;;
    latency) timeout 300 build/latency edlib_amd/libedlib.so | tee $OUT/${R}_latency_engine.json
             timeout 300 build/latency oracle/_ref/libedlib_ref.so | tee $OUT/${R}_latency_reference.json ;;
    short)   timeout 600 python tools/bench_short_pairs.py 2>&1 | tee $OUT/${R}_short_pairs.json | cut -c1-2000 ;;
    path)    timeout 600 python tools/bench_path.py 2>&1 | tee $OUT/${R}_path.json | cut -c1-2000 ;;
    hwlong)  timeout 900 python tools/bench_hw_long.py ${HWLONG_ARGS:-} 2>&1 | tee $OUT/${R}_hw_long.json | cut -c1-2500 ;;
    wide)    timeout 600 python tools/bench_wide.py 2>&1 | tee $OUT/${R}_wide_target.json | cut -c1-1500 ;;
    chrom)   timeout 600 python tools/bench_chromosome.py ${CHROM_ARGS:---ref} 2>&1 | cut -c1-520 | tee $OUT/${R}_chromosome.json ;;
    chromtrace)
      d=$OUT/trace_chrom; rm -rf $d; mkdir -p $d
      ( cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o t -- python $ROOT/tools/bench_chromosome.py --repeat 1 > $d/out.json 2> $d/err.log )
      f=$(find $d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${R}_chromosome_kernel_stats.csv && head -8 $f | cut -c1-170
      f=$(find $d -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python $ROOT/tools/prof_summaries.py trace "$f" "$OUT/${R}_chromosome_kernel_trace_edlib.csv"
      rm -rf $d ;;
    chromtraffic)
      for ctr in FETCH_SIZE WRITE_SIZE; do
        d=$OUT/pmc_chrom_$ctr; rm -rf $d; mkdir -p $d
        ( cd /tmp && timeout -k 5 300 rocprofv3 --pmc $ctr --output-format csv -d $d -o p -- python $ROOT/tools/bench_chromosome.py --percents 99 --no-path --repeat 1 > $d/out.json 2> $d/err.log )
        f=$(find $d -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python $ROOT/tools/prof_summaries.py pmc "$f" "$OUT/${R}_pmc_chromosome_${ctr}.csv"
        rm -rf $d
      done ;;
    soak)    timeout $(( ${SOAK_SECONDS:-120} + 120 )) python tools/soak.py ${SOAK_SECONDS:-120} ${SOAK_SEED:-3} 2>&1 | tail -5 | tee $OUT/${R}_soak.log ;;
    ab)      # the headline step of two builds of the library on THIS box, alternating (AB_LIB: the other build)
             for i in 1 2; do for lib in edlib_amd/libedlib.so ${AB_LIB:-build/ab/libedlib_r02.so}; do
               EDLIB_AMD_LIB=$ROOT/$lib timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['ms_per_step'], d['roofline']['scan_ms_per_step'])"
             done; done | tee $OUT/${R}_ab.log ;;
    cmd)     timeout ${CMD_TIMEOUT:-900} bash -c "$CMD" 2>&1 | tail -${CMD_TAIL:-40} | tee $OUT/${R}_cmd.log ;;
    *) echo "unknown stage $stage" ;;
  esac
done
say done; ls $OUT | head -60
