python -m pytest tests -m gpu -x -q > gpurun_out/r3_t9_tests.log 2>&1; tail -6 gpurun_out/r3_t9_tests.log
bash tools/prof_heads.sh
python tools/bench_conv.py hg_s2 conv2 2>&1 | grep -v amdgpu.ids
