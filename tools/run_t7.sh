python -m pytest tests -m gpu -x -q > gpurun_out/r3_t7_tests.log 2>&1; tail -6 gpurun_out/r3_t7_tests.log
python bench.py > gpurun_out/r3_t7_bench.json 2> gpurun_out/r3_t7_bench.err; head -c 300 gpurun_out/r3_t7_bench.json; tail -3 gpurun_out/r3_t7_bench.err
