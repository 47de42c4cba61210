python -m pytest tests/test_gpu_parity.py tests/test_gpu_f16.py -m gpu -x -q -k "gather or k3_stride2 or hourglass or global or randomised" > gpurun_out/r3_t6_tests.log 2>&1; tail -5 gpurun_out/r3_t6_tests.log
for L in gather_cfg3 gather_f16 gather_proj gather_uniform; do python tools/prof_layers.py $L --reps 20; done 2>&1 | grep -v amdgpu.ids
python tools/bench_conv.py hg_s2 hg_s2b conv2 hg_c2 2>&1 | grep -v amdgpu.ids
