#!/bin/bash
# The evidence under profiles/ for one round (R, default r02): for each bench config the rocprofv3 kernel trace of
# the bench command itself, then PMC passes (each in its own run, kernel-trace/stats only, under `timeout`:
# an unbounded PMC pass once hung for 25 min on this pool).  Writes gpurun_out/profiles_$R/.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$PWD; R=${R:-r02}; OUT=$ROOT/gpurun_out/profiles_$R; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
trace() {   # $1 = tag, rest = bench args
  tag=$1; shift; d=$OUT/trace_$tag; mkdir -p $d
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o t -- python $ROOT/bench.py "$@" > $d/bench.json 2> $d/err.log )
  cp $d/bench.json $OUT/${R}_${tag}_underprofiler.json
  f=$(find $d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${R}_${tag}_kernel_stats.csv
  f=$(find $d -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python - "$f" "$OUT/${R}_${tag}_kernel_trace_edlib.csv" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "edlib_amd" in r["Kernel_Name"]]
keep = ["Kernel_Name", "Start_Timestamp", "End_Timestamp", "Grid_Size_X", "Grid_Size_Y", "Workgroup_Size_X", "LDS_Block_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size"]
keep = [k for k in keep if rows and k in rows[0]]
w = csv.writer(open(sys.argv[2], "w")); w.writerow(keep + ["Duration_ms"])
for r in rows: w.writerow([r[k] for k in keep] + ["%.4f" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)])
PY
  head -7 $OUT/${R}_${tag}_kernel_stats.csv | cut -c1-160
}
pmc() {     # $1 = tag, $2 = counters (space separated), rest = bench args
  tag=$1; ctr=$2; shift; shift; d=$OUT/pmc_$tag; mkdir -p $d
  ( cd /tmp && timeout -k 5 ${PMC_TIMEOUT:-240} rocprofv3 --pmc $ctr --output-format csv -d $d -o p -- python $ROOT/bench.py "$@" --no-cpu-baseline --no-e2e > $d/bench.json 2> $d/err.log )
  echo "== pmc $tag rc=$?"
  f=$(find $d -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" "$OUT/${R}_pmc_${tag}.csv" <<'PY'
import csv, sys, collections
acc = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Kernel_Name"], r["Counter_Name"])
    a = acc.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
w = csv.writer(open(sys.argv[2], "w")); w.writerow(["Kernel_Name", "Counter_Name", "Dispatches", "Sum"])
for (k, c), (n, v) in acc.items():
    w.writerow([k, c, n, "%.6g" % v])
    if "edlib_amd" in k: print("  %-70s %-22s n=%d sum=%.6g" % (k[:70], c, n, v))
PY
}
if [ -z "$SKIP_TRACE" ]; then
  trace bench_default
  trace bench_c4 --config 4 --no-cpu-baseline
  trace bench_c5 --config 5 --no-cpu-baseline
fi
if [ -z "$SKIP_PMC" ]; then
  pmc fetch_c2_1M "FETCH_SIZE" --steps 1 --warmup 0
  pmc write_c2_1M "WRITE_SIZE" --steps 1 --warmup 0
  pmc sq1_c2_262k "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" --reads 262144 --steps 1 --warmup 0
  pmc sq2_c2_262k "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" --reads 262144 --steps 1 --warmup 0
  # config 4 at 20,000 units: the counter passes serialise the dispatches and the full 100,000 ran into the timeout
  pmc fetch_c4_20k "FETCH_SIZE" --config 4 --units 20000 --steps 1 --warmup 0
  pmc write_c4_20k "WRITE_SIZE" --config 4 --units 20000 --steps 1 --warmup 0
  pmc sq1_c4_20k "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH" --config 4 --units 20000 --steps 1 --warmup 0
  pmc sq2_c4_20k "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" --config 4 --units 20000 --steps 1 --warmup 0
fi
( cd $ROOT && timeout 300 build/latency edlib_amd/libedlib.so > $OUT/${R}_latency_engine.json; timeout 300 build/latency oracle/_ref/libedlib_ref.so > $OUT/${R}_latency_reference.json; cat $OUT/${R}_latency_*.json; timeout 600 python tools/bench_wide.py > $OUT/${R}_wide_target.json; cat $OUT/${R}_wide_target.json )
ls $OUT
