bash tools/prof_r3.sh
python bench.py > gpurun_out/r3_t15_bench.json 2> gpurun_out/r3_t15_bench.err; head -c 200 gpurun_out/r3_t15_bench.json; tail -2 gpurun_out/r3_t15_bench.err
LAYERS="conv2_side sheared" SQ_LAYERS="conv2_side" bash tools/pmc_r3_traffic.sh > gpurun_out/pmc_r3_traffic2.log 2>&1; tail -2 gpurun_out/pmc_r3_traffic2.log
