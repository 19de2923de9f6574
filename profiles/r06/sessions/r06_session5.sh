#!/bin/bash
# round 6, GPU session 5: the fine look-up table's second build (wave per LUT cell, per-cascade subdivision): parity, cage-move cost, A/B
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_s5
mkdir -p $OUT
export NRS_DEV_KNOBS=1
cd $R
( time timeout 1500 python -m pytest tests/test_gpu_fine_lut.py tests/test_gpu_parity.py tests/test_gpu_cage_update.py tests/test_gpu_grid_refresh.py -x -q -m gpu ) > $OUT/tests_fine_lut.log 2>&1
tail -4 $OUT/tests_fine_lut.log
python - > $OUT/next_rows_fine.json 2> $OUT/next_rows_fine.err <<PY
import json, sys, os
sys.path.insert(0, "$R")
import torch, bench
from nerfshop_amd import runtime as rt, synth
ctx = rt.Context(0)
print(json.dumps(bench.next_rows(rt, synth, ctx, torch)))
PY
cat $OUT/next_rows_fine.json | cut -c1-400
for WL in lego_cage garden_cage; do
  bash tools/ab_env.sh $OUT/ab_fine_lut_$WL.txt $WL "nofine=NRS_NO_FINE_LUT=1" "fine=NRS_NOTHING=1"
done
for WL in lego_cage garden_cage; do
NRS_DEBUG=4 python bench.py --workload $WL --steps 1 --warmup 0 --no-extra --no-cpu-baseline 2> $OUT/prof_fine_$WL.err > /dev/null
grep -E "nrs cage scan" $OUT/prof_fine_$WL.err | head -2
done
