LAYER=wgrad_conv2 MATCH=wgrad_wino_kernel bash tools/pmc_r3.sh > gpurun_out/r3_t18_pmc.log 2>&1
tail -30 gpurun_out/r3_t18_pmc.log
