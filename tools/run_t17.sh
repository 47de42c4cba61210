python -m pytest tests -m gpu -x -q -k "training or sheared or hourglass or train or statistics" > gpurun_out/r3_t17_tests.log 2>&1; tail -15 gpurun_out/r3_t17_tests.log
python bench.py --mode train --steps 20 --warmup 5 > gpurun_out/r3_t17_train.json 2> gpurun_out/r3_t17_train.err; head -c 600 gpurun_out/r3_t17_train.json; echo
