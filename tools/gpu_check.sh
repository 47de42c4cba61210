# The round's standard GPU check (run through gpurun from the repo root):
#   gpurun --timeout 1200 -- bash tools/gpu_check.sh
# full `-m gpu` suite, smoke, the default bench line and the training line; outputs under gpurun_out/.
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/check_tests.log 2>&1; tail -4 gpurun_out/check_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/check_smoke.log 2>&1; tail -1 gpurun_out/check_smoke.log
python bench.py > gpurun_out/check_bench.json 2> gpurun_out/check_bench.err; tail -1 gpurun_out/check_bench.json | head -c 250; echo
python bench.py --mode train --steps 20 --warmup 5 > gpurun_out/check_train.json 2> gpurun_out/check_train.err; tail -1 gpurun_out/check_train.json | head -c 330 | tail -c 130; echo
