python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sheared or training_step" > gpurun_out/r3_t17_tests.log 2>&1; tail -15 gpurun_out/r3_t17_tests.log
python bench.py --mode train --steps 5 --warmup 2 > gpurun_out/r3_t17_train.json 2> gpurun_out/r3_t17_train.err; head -c 600 gpurun_out/r3_t17_train.json; echo
