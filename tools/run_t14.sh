python -m pytest tests -m gpu -x -q > gpurun_out/r3_t14_tests.log 2>&1; tail -5 gpurun_out/r3_t14_tests.log
python tools/bench_conv.py hg_s2 hg_c2 hg_s2b hg_c4 dc5 2>&1 | grep -v amdgpu.ids
python bench.py --no-extras --no-cpu-baseline > gpurun_out/r3_t14_bench.json 2> gpurun_out/r3_t14_bench.err; head -c 300 gpurun_out/r3_t14_bench.json; tail -3 gpurun_out/r3_t14_bench.err
