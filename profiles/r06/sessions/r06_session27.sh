#!/bin/bash
# round 6, GPU session 27: four levels per round trip in the membrane instantiation (main gather / the un-deformed pass's gather / both): parity + A/B
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_s27
mkdir -p $OUT
cd $R
V=$R/nerfshop_amd/csrc/variants
( NRS_LIB_PATH=$V/libnrs_pq3.so timeout 900 python -m pytest tests/test_gpu_bench_parity.py tests/test_gpu_poisson_boundary.py -x -q -m gpu -k "membrane or poisson" ) > $OUT/tests_pq3.log 2>&1
tail -2 $OUT/tests_pq3.log
bash tools/ab_bench.sh $OUT/ab_poisson_quads.txt lego_cage_membrane base=default pq1=$V/libnrs_pq1.so pq2=$V/libnrs_pq2.so pq3=$V/libnrs_pq3.so
