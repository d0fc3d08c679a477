# tall unrelated reads (chained strips, long_reads.hip solveTallFull): run_ms per routing knob
#   LENGTHS="4096 6000" CFGS="minwaves:warms:waves ..." bash tools/tall_matrix.sh
cd ${GRAFT_REPO_ROOT:-.}
for L in ${LENGTHS:-10000}; do
 for cfg in ${CFGS:-"768:1:2048"}; do
  IFS=: read mw wm tw <<< "$cfg"
  echo "== len $L minwaves $mw warms $wm waves ${tw:-2048}"
  EDLIB_AMD_TALL_MIN_WAVES=$mw EDLIB_AMD_TALL_WARMS=$wm EDLIB_AMD_TALL_WAVES=${tw:-2048} timeout -k 5 200 python tools/bench_read_length_cliff.py --lengths $L --no-extra --sample 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:(v['run_ms'],v['path'],v['sample_ok']) for k,v in d.items()})"
 done
done
