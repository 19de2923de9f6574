#!/bin/bash
# round 6, GPU session 22: L2 misses per sample of the knee (three phases) and of the configuration without sparse records (four phases), gate on / off
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_s22
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export NRS_DEV_KNOBS=1
for W in garden_cage garden_cage_norecords; do
  for G in 1 0; do
    D=$OUT/${W}_gate$G
    NRS_L2_GATE=$G rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace -d $D -o bench -- python $R/bench.py --workload $W --steps 4 --warmup 1 --no-cpu-baseline --no-extra > $D.log 2>&1
    python $R/tools/pmc_kernel.py $D > $D.json 2>> $D.log
    echo "$W gate=$G $(cat $D.json)"
    rm -rf $D
  done
done | tee $OUT/garden_gate_phases_pmc.txt
