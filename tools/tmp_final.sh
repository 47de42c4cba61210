bash tools/prof_train_r3.sh > gpurun_out/final_prof_train.log 2>&1; tail -1 gpurun_out/final_prof_train.log
bash tools/prof_heads.sh > gpurun_out/final_prof_heads.log 2>&1; tail -1 gpurun_out/final_prof_heads.log
bash tools/prof_r3.sh > gpurun_out/final_prof_bench.log 2>&1; tail -1 gpurun_out/final_prof_bench.log
python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; tail -1 gpurun_out/final_bench.json | head -c 250; echo
