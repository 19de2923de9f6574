#!/bin/bash
# round 6, GPU session 1: the widened 1080p parity net, the gather probe's cache policies, the garden frame's misses per level pair, the record-budget curve
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_s1
mkdir -p $OUT
cd $R
( time timeout 900 python -m pytest tests/test_gpu_bench_parity.py -x -q -m gpu -s ) > $OUT/bench_parity.log 2>&1
tail -5 $OUT/bench_parity.log
# --- cache policies of a 32-byte gather that misses L2: request sizes at the fabric side
P=$R/tools/probe/gather_probe
cd /tmp && export TMPDIR=/tmp
for POL in 0 1 2 3; do
  for MB in 65536 128; do
    $P $MB 32 64 524288 $POL >> $OUT/policy_time.jsonl 2>&1
    D=/tmp/gp_pol; rm -rf $D
    timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace -d $D -o gp -- $P $MB 32 64 524288 $POL > /tmp/gp.log 2>&1
    echo "{\"policy\": $POL, \"table_mb\": $MB, \"pmc\": $(python $R/tools/pmc_kernel.py $D gather_kernel)}" >> $OUT/policy_pmc.jsonl
    D=/tmp/gp_pol2; rm -rf $D
    timeout 300 rocprofv3 --pmc TCC_MISS_sum TCC_HIT_sum TCC_REQ_sum FETCH_SIZE --kernel-trace -d $D -o gp -- $P $MB 32 64 524288 $POL > /tmp/gp.log 2>&1
    echo "{\"policy\": $POL, \"table_mb\": $MB, \"pmc\": $(python $R/tools/pmc_kernel.py $D gather_kernel)}" >> $OUT/policy_pmc.jsonl
  done
done
cat $OUT/policy_time.jsonl $OUT/policy_pmc.jsonl
# --- garden: misses per level pair
cd $R
bash profiles/r06/sessions/r06_garden_levels.sh $OUT
# --- garden: the record budget curve (no profiler)
for GB in 0 4 8 16 24 32 48 64; do
  L=$(NRS_SPARSE_GB=$GB python bench.py --workload garden_cage --steps 16 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | tail -1)
  echo "{\"sparse_gb\": $GB, \"bench\": $L}" >> $OUT/garden_budget.jsonl
done
python - <<PY
import json
for l in open("$OUT/garden_budget.jsonl"):
    j = json.loads(l); b = j["bench"]
    print(j["sparse_gb"], b["value"], b["roofline"]["kernel_ms"], b["config"]["cell_records"])
PY
