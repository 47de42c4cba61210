# Extra counter passes committed under profiles/r1 (run through gpurun from the repo root).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_extra; mkdir -p $O
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/gather_fetch -o p -- python3 tools/prof_layers.py gather --reps 2 > $O/l1.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/gather_write -o p -- python3 tools/prof_layers.py gather --reps 2 > $O/l2.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_WAIT_ANY --output-format csv -d $O/hg_sq -o p -- python3 tools/prof_layers.py hg --reps 2 > $O/l3.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_WAIT_ANY --output-format csv -d $O/trunk_sq -o p -- python3 tools/prof_layers.py trunk --reps 1 > $O/l4.log 2>&1
ls $O/*
