python bench.py --no-extras > gpurun_out/r3_t29_bench.json 2>/dev/null; tail -1 gpurun_out/r3_t29_bench.json | head -c 250; echo
python bench.py --no-extras > gpurun_out/r3_t29_bench2.json 2>/dev/null; tail -1 gpurun_out/r3_t29_bench2.json | head -c 250; echo
