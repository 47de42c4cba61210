python -m pytest tests -m gpu -x -q > gpurun_out/r3_t20_tests.log 2>&1; tail -8 gpurun_out/r3_t20_tests.log
python tools/bench_conv.py > gpurun_out/r3_t20_conv.log 2>&1; cat gpurun_out/r3_t20_conv.log | tail -12
python bench.py > gpurun_out/r3_t20_bench.json 2> gpurun_out/r3_t20_bench.err; head -c 700 gpurun_out/r3_t20_bench.json; echo
