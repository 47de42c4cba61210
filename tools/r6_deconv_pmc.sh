set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/pmc_dc_a -o p -- python3 $R/tools/r6_deconv_probe.py > $R/gpurun_out/pmc_dc_a.log 2>&1
timeout -k 10 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $R/gpurun_out/pmc_dc_b -o p -- python3 $R/tools/r6_deconv_probe.py > $R/gpurun_out/pmc_dc_b.log 2>&1
