#!/bin/bash
# kernel trace of bench.py at 262,144 reads (2 timed steps) -> gpurun_out/prof
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; rm -rf gpurun_out/prof && mkdir -p gpurun_out/prof
export TMPDIR=/tmp
R=$PWD
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o r01 -- python $R/bench.py --reads 262144 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof.err )
cut -c1-200 gpurun_out/prof_bench.json
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "scan_reads" in r["Kernel_Name"]]
for r in rows[-7:]:
    print(r["Kernel_Name"][:60], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, "ms", "grid", r.get("Grid_Size_X"), r.get("Grid_Size_Y"))
PY
