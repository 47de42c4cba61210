# Collects the profiles committed under profiles/r1 (run through gpurun from the repo root).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_round; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o bench -- python3 bench.py --steps 20 --warmup 5 > $O/bench.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o p -- python3 tools/prof_layers.py conv1_factored --reps 2 > $O/fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o p -- python3 tools/prof_layers.py conv1_factored --reps 2 > $O/write.log 2>&1
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_WAIT_ANY --output-format csv -d $O/sq -o p -- python3 tools/prof_layers.py conv1_factored --reps 2 > $O/sq.log 2>&1
ls $O/*
