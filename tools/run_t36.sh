python -m pytest tests -m gpu -x -q > gpurun_out/r3_t36_tests.log 2>&1; tail -4 gpurun_out/r3_t36_tests.log
python bench.py --no-extras > gpurun_out/r3_t36_bench.json 2>/dev/null; tail -1 gpurun_out/r3_t36_bench.json | head -c 250; echo
