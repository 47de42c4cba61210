python bench.py > gpurun_out/r3_t27_bench.json 2> gpurun_out/r3_t27_bench.err; head -c 300 gpurun_out/r3_t27_bench.json; echo
bash tools/prof_train_r3.sh > gpurun_out/r3_t27_prof.log 2>&1; tail -2 gpurun_out/r3_t27_prof.log
