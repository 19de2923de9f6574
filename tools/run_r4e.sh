mkdir -p gpurun_out/r4e
V=nerfshop_amd/csrc/variants
NRS_PROBE_QUICK=1 NRS_PROBE_N=1,8 python tools/scale_probe_r03.py > gpurun_out/r4e/scale_new.md 2> gpurun_out/r4e/err1.log
NRS_LIB_PATH=$V/libnrs_alloff.so NRS_PROBE_QUICK=1 NRS_PROBE_N=1,8 python tools/scale_probe_r03.py > gpurun_out/r4e/scale_alloff.md 2> gpurun_out/r4e/err2.log
grep "^| [18] " gpurun_out/r4e/scale_new.md; echo; grep "^| [18] " gpurun_out/r4e/scale_alloff.md
