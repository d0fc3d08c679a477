mkdir -p gpurun_out/sp
EDLIB_AMD_DEBUG=1 timeout 120 python tools/short_pairs_probe.py hw 2> gpurun_out/sp/hw_debug.txt
EDLIB_AMD_DEBUG=1 timeout 120 python tools/short_pairs_probe.py nw 2> gpurun_out/sp/nw_debug.txt
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/sp/prof_hw -o hw -- python /root/repo/tools/short_pairs_probe.py hw > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/sp/prof_nw -o nw -- python /root/repo/tools/short_pairs_probe.py nw > /dev/null 2>&1
cd /root/repo
sed -n '/==== second run/,$p' gpurun_out/sp/hw_debug.txt | head -60
echo; sed -n '/==== second run/,$p' gpurun_out/sp/nw_debug.txt | head -80
find gpurun_out/sp -name "*kernel_stats.csv" | while read f; do echo $f; head -12 $f | cut -c1-200; done
find gpurun_out/sp -name "*kernel_trace.csv" -size +1M -delete
