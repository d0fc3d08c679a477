#!/bin/bash
# kernel trace of one bench config (CFG=4|5, default 4): per-dispatch durations of this library's kernels
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; CFG=${CFG:-4}; out=$R/gpurun_out/prof_c$CFG; rm -rf $out; mkdir -p $out; export TMPDIR=/tmp
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o cfg -- \
    python $R/bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} > $out/bench.json 2> $out/err.log )
python - <<PY
import csv, glob, json
print({k: v for k, v in json.load(open("$out/bench.json")).items() if k in ("value", "ms_per_step", "roofline", "valu_roofline")})
f = glob.glob("$out/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "edlib_amd" in r["Kernel_Name"]]
n = len(rows) // 3
for r in rows[-n:]:
    print("%-90s %9.3f ms  grid %s  vgpr %s lds %s" % (r["Kernel_Name"][:90], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r.get("Grid_Size_X", r.get("Grid_Size")), r.get("VGPR_Count"), r.get("LDS_Block_Size")))
PY
