python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r3_t33_smoke.log 2>&1; tail -1 gpurun_out/r3_t33_smoke.log
python bench.py > gpurun_out/r3_t33_bench.json 2> gpurun_out/r3_t33_bench.err; tail -1 gpurun_out/r3_t33_bench.json | head -c 250; echo
bash tools/prof_r3.sh > gpurun_out/r3_t33_prof.log 2>&1; tail -3 gpurun_out/r3_t33_prof.log
