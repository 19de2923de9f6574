#!/bin/bash
# round 6, GPU session 10: four sparse brick levels in two round trips (garden): parity + A/B
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_s10
mkdir -p $OUT
export NRS_DEV_KNOBS=1
cd $R
( time timeout 1500 python -m pytest tests/test_gpu_cell_cache.py tests/test_gpu_numerics.py -x -q -m gpu ) > $OUT/tests_a.log 2>&1
tail -4 $OUT/tests_a.log
( time timeout 900 python -m pytest tests/test_gpu_bench_parity.py -x -q -m gpu -k "garden" ) > $OUT/tests_b.log 2>&1
tail -4 $OUT/tests_b.log
V=$R/nerfshop_amd/csrc/variants
bash tools/ab_bench.sh $OUT/ab_sparse_quads_garden_cage.txt garden_cage base=$V/libnrs_base.so quads=default
bash tools/ab_bench.sh $OUT/ab_sparse_quads_garden.txt garden base=$V/libnrs_base.so quads=default
NRS_L2_GATE=0 bash tools/ab_bench.sh $OUT/ab_sparse_quads_garden_cage_nogate.txt garden_cage base=$V/libnrs_base.so quads=default
