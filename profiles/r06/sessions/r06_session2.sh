#!/bin/bash
# round 6, GPU session 2: L2 retention under cache policies, the cage scan's counters + phase shares (lego_cage, varied), the partial-refill A/B, the round-latency model, one full bench line
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_s2
mkdir -p $OUT
export NRS_DEV_KNOBS=1
cd $R
# --- phases / walk / cage scan of the profiling instantiation
for WL in lego_cage lego_cage_varied garden_cage; do
  NRS_DEBUG=4 python bench.py --workload $WL --steps 1 --warmup 0 --no-extra --no-cpu-baseline 2> $OUT/prof_$WL.err > $OUT/prof_$WL.out
  grep -E "nrs phases|nrs walk|nrs cage scan|nrs waves|nrs hand-over" $OUT/prof_$WL.err | head -60
done
# --- partial refill A/B
V=$R/nerfshop_amd/csrc/variants
for WL in lego_cage lego_cage_varied; do
  bash tools/ab_bench.sh $OUT/ab_refill_$WL.txt $WL base=default r16=$V/libnrs_refill16.so r32=$V/libnrs_refill32.so r48=$V/libnrs_refill48.so
done
# --- L2 retention probe: hot table (plain 4-byte gathers) + cold 32-byte gathers under a policy / an allocation kind
P=$R/tools/probe/l2_retention_probe
cd /tmp && export TMPDIR=/tmp
run_ret() { # hot_mb policy hot_per cold_per alloc
  $P $1 32768 $2 32 $3 $4 $5 >> $OUT/l2ret_time.jsonl 2>&1
  D=/tmp/l2r; rm -rf $D
  timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace -d $D -o gp -- $P $1 32768 $2 32 $3 $4 $5 > /tmp/gp.log 2>&1
  echo "{\"hot_mb\": $1, \"policy\": $2, \"hot_per\": $3, \"cold_per\": $4, \"alloc\": $5, \"pmc\": $(python $R/tools/pmc_kernel.py $D probe_kernel)}" >> $OUT/l2ret_pmc.jsonl
}
for HOT in 2 4; do
  run_ret $HOT 0 8 0 0          # the hot table alone
  for POL in 0 1 4 7 8 9; do run_ret $HOT $POL 8 2 0; done
  run_ret $HOT 0 8 2 1          # cold table in uncached memory
  run_ret $HOT 0 8 2 2          # cold table in fine-grained memory
done
# request sizes of cold gathers alone from uncached / fine-grained memory
for AL in 0 1 2; do
  D=/tmp/l2r; rm -rf $D
  timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace -d $D -o gp -- $P 2 32768 0 32 0 8 $AL > /tmp/gp.log 2>&1
  echo "{\"cold_only_alloc\": $AL, \"pmc\": $(python $R/tools/pmc_kernel.py $D probe_kernel)}" >> $OUT/l2ret_pmc.jsonl
  $P 2 32768 0 32 0 8 $AL >> $OUT/l2ret_time.jsonl 2>&1
done
cat $OUT/l2ret_time.jsonl $OUT/l2ret_pmc.jsonl
cd $R
# --- round latency model
python tools/round_latency_probe.py > $OUT/round_latency.md 2> $OUT/round_latency.err
cat $OUT/round_latency.md
# --- one full bench line
python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err
tail -c 3000 $OUT/bench_full.json
