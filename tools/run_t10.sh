python -m pytest tests -m gpu -x -q > gpurun_out/r3_t10_tests.log 2>&1; tail -8 gpurun_out/r3_t10_tests.log
python bench.py --no-extras --no-cpu-baseline > gpurun_out/r3_t10_bench.json 2> gpurun_out/r3_t10_bench.err; head -c 400 gpurun_out/r3_t10_bench.json; tail -3 gpurun_out/r3_t10_bench.err
