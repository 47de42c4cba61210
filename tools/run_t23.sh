python -m pytest tests -m gpu -x -q > gpurun_out/r3_t23_tests.log 2>&1; tail -5 gpurun_out/r3_t23_tests.log
python bench.py --mode train --steps 20 --warmup 5 > gpurun_out/r3_t23_train.json 2> gpurun_out/r3_t23_train.err; head -c 400 gpurun_out/r3_t23_train.json; echo
