#!/bin/bash
# round 6, GPU session 9: membrane colour wide loads vs the old coefficient-by-coefficient code (same box), gate condition check
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_s9
mkdir -p $OUT
export NRS_DEV_KNOBS=1
cd $R
V=$R/nerfshop_amd/csrc/variants
bash tools/ab_bench.sh $OUT/ab_membrane_sh.txt lego_cage_membrane wide=default old=$V/libnrs_shold.so
bash tools/ab_env.sh $OUT/ab_gate_garden_cage_norecords.txt garden_cage_norecords "nogate=NRS_L2_GATE=0" "gate=NRS_L2_GATE=1"
