#!/bin/bash
# round 6, GPU session 8: membrane colour with wide loads (parity + A/B), the GATE instantiation (parity on the garden scene + A/B), lazy fine table
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_s8
mkdir -p $OUT
export NRS_DEV_KNOBS=1
cd $R
( time timeout 1500 python -m pytest tests/test_gpu_fine_lut.py tests/test_gpu_parity.py tests/test_gpu_cage_update.py tests/test_gpu_poisson_boundary.py tests/test_gpu_modes.py tests/test_gpu_cell_cache.py -x -q -m gpu ) > $OUT/tests_a.log 2>&1
tail -4 $OUT/tests_a.log
( time timeout 900 python -m pytest tests/test_gpu_bench_parity.py -x -q -m gpu -k "membrane or garden or (views_against_the_oracle and 2)" ) > $OUT/tests_b.log 2>&1
tail -4 $OUT/tests_b.log
V=$R/nerfshop_amd/csrc/variants
bash tools/ab_bench.sh $OUT/ab_membrane_sh.txt lego_cage_membrane wide=default serial=$V/libnrs_shserial.so
for WL in garden_cage garden garden_cage_norecords; do
  bash tools/ab_env.sh $OUT/ab_gate_$WL.txt $WL "nogate=NRS_L2_GATE=0" "gate=NRS_L2_GATE=1"
done
bash tools/ab_env.sh $OUT/ab_gate_lego_cage.txt lego_cage "nogate=NRS_L2_GATE=0" "gate=NRS_L2_GATE=1"
