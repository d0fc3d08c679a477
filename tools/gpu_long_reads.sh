set -x
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_long_reads.py -x -q > gpurun_out/long_tests.log 2>&1; echo "rc=$?" >> gpurun_out/long_tests.log
tail -15 gpurun_out/long_tests.log
timeout 300 python tools/bench_read_length_cliff.py > gpurun_out/cliff.json 2> gpurun_out/cliff.err; tail -3 gpurun_out/cliff.err; cat gpurun_out/cliff.json
