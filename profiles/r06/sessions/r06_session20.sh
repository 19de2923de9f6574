#!/bin/bash
# round 6, GPU session 20: phase length / wait cap of the L2 gate with three and four phases (garden_cage = 3 phases, garden_cage_norecords = 4, garden_cage_records64 = 2)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_s20
mkdir -p $OUT
cd $R
V=$R/nerfshop_amd/csrc/variants
for W in garden_cage garden_cage_norecords garden_cage_records64; do
  bash tools/ab_bench.sh $OUT/ab_gate_tune_$W.txt $W s11c2=default s10c2=$V/libnrs_s10c2.so s11c4=$V/libnrs_s11c4.so s10c4=$V/libnrs_s10c4.so s12c2=$V/libnrs_s12c2.so
done
