#!/bin/bash
# round 6, GPU session 11: occupancy word of the next sample fetched with the gather (NRS_OCC_PREFETCH): parity of the variant + A/B on the lego scenes
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_s11
mkdir -p $OUT
export NRS_DEV_KNOBS=1
cd $R
V=$R/nerfshop_amd/csrc/variants
( time NRS_LIB_PATH=$V/libnrs_occpre.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_parity.py -x -q -m gpu -k "not garden" ) > $OUT/tests_occpre.log 2>&1
tail -4 $OUT/tests_occpre.log
bash tools/ab_bench.sh $OUT/ab_occpre_lego_cage.txt lego_cage base=default occpre=$V/libnrs_occpre.so
bash tools/ab_bench.sh $OUT/ab_occpre_lego_cage_varied.txt lego_cage_varied base=default occpre=$V/libnrs_occpre.so
bash tools/ab_bench.sh $OUT/ab_occpre_noedit.txt noedit base=default occpre=$V/libnrs_occpre.so
