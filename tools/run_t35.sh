python -m pytest tests/test_gpu_neck2d.py tests/test_gpu_parity.py -m gpu -x -q -k "neck or deconv or vernier or hourglass2d or golden" > gpurun_out/r3_t35_tests.log 2>&1; tail -12 gpurun_out/r3_t35_tests.log
for c in 2 8; do python tools/prof_heads.py --crops $c --reps 20 2>&1 | tail -1; done
