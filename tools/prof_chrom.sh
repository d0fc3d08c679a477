cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/v2
EDLIB_AMD_DEBUG=1 timeout 300 python tools/bench_chromosome.py --percents 99 --no-path --repeat 1 2>&1 | tail -40 | tee gpurun_out/v2/debug99.log
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1 -o t -- python $GRAFT_REPO_ROOT/tools/bench_chromosome.py --percents 99,90 --repeat 1 > $GRAFT_REPO_ROOT/gpurun_out/v2/prof.log 2>&1
f=$(find /tmp/prof1 -name "*kernel_stats.csv" | head -1); cp $f $GRAFT_REPO_ROOT/gpurun_out/v2/kernel_stats.csv; head -20 $f | cut -c1-200
f=$(find /tmp/prof1 -name "*kernel_trace.csv" | head -1); python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n=r['Kernel_Name']
    if 'wide' in n or 'ring' in n:
        print(n[:70], r['Grid_Size_X'] if 'Grid_Size_X' in r else '', (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6,'ms', r.get('Grid_Size',''), r.get('Workgroup_Size',''))
PY
