python -m pytest tests -m gpu -x -q > gpurun_out/r3_t16_tests.log 2>&1; tail -5 gpurun_out/r3_t16_tests.log
python bench.py --config cfg3 --steps 2 --warmup 1 > gpurun_out/r3_t16_cfg3.json 2> gpurun_out/r3_t16_cfg3.err; head -c 600 gpurun_out/r3_t16_cfg3.json; echo
python bench.py --mode train --steps 5 --warmup 2 > gpurun_out/r3_t16_train.json 2> gpurun_out/r3_t16_train.err; head -c 400 gpurun_out/r3_t16_train.json; echo
