#!/bin/bash
# refresh of the config-2 evidence on the final code of a round: kernel trace of the bench command + FETCH / WRITE passes
# (+ the SQ pass of the ring kernel that timed out in the first collection).  Writes gpurun_out/profiles_${R}b/.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export R=${R:-r02}b SKIP_TRACE=1 SKIP_PMC=1
source /dev/null
ROOT=$PWD; OUT=$ROOT/gpurun_out/profiles_$R; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
# reuse the functions of gpu_profiles.sh
eval "$(sed -n '/^trace() {/,/^}/p;/^pmc() {/,/^}/p' tools/gpu_profiles.sh)"
trace bench_default
pmc fetch_c2_1M "FETCH_SIZE" --steps 1 --warmup 0
pmc write_c2_1M "WRITE_SIZE" --steps 1 --warmup 0
PMC_TIMEOUT=200 pmc sq2_c4_20k "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" --config 4 --units 20000 --steps 1 --warmup 0
ls $OUT
