#!/bin/bash
# round 6, GPU session 4: the fine look-up table under the cage LUT (parity, A/B, scan counters, cage-move cost); more phase-gate variants on the garden frame
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_s4
mkdir -p $OUT
export NRS_DEV_KNOBS=1
cd $R
( time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cage_update.py tests/test_gpu_grid_refresh.py tests/test_gpu_poisson_boundary.py tests/test_gpu_affine.py tests/test_gpu_cpp_host.py tests/test_gpu_formats.py tests/test_gpu_fine_lut.py -x -q -m gpu ) > $OUT/tests_fine_lut.log 2>&1
tail -4 $OUT/tests_fine_lut.log
( time timeout 900 python -m pytest tests/test_gpu_bench_parity.py -x -q -m gpu -k "views_against_the_oracle and (0 or 5) or membrane or garden" ) > $OUT/tests_fine_lut_bench.log 2>&1
tail -4 $OUT/tests_fine_lut_bench.log
for WL in lego_cage lego_cage_varied lego_cage_membrane garden_cage; do
  bash tools/ab_env.sh $OUT/ab_fine_lut_$WL.txt $WL "nofine=NRS_NO_FINE_LUT=1" "fine=NRS_NOTHING=1"
done
NRS_DEBUG=4 python bench.py --workload lego_cage --steps 1 --warmup 0 --no-extra --no-cpu-baseline 2> $OUT/prof_fine_lego.err > /dev/null
grep -E "nrs phases|nrs cage scan" $OUT/prof_fine_lego.err | head -4
python - > $OUT/next_rows_fine.json 2> $OUT/next_rows_fine.err <<PY
import json, sys, os
sys.path.insert(0, "$R")
import torch, bench
from nerfshop_amd import runtime as rt, synth
ctx = rt.Context(0)
print(json.dumps(bench.next_rows(rt, synth, ctx, torch)))
PY
cat $OUT/next_rows_fine.json | cut -c1-600
NRS_NO_FINE_LUT=1 python - > $OUT/next_rows_nofine.json 2> /dev/null <<PY
import json, sys, os
sys.path.insert(0, "$R")
import torch, bench
from nerfshop_amd import runtime as rt, synth
ctx = rt.Context(0)
print(json.dumps(bench.next_rows(rt, synth, ctx, torch)))
PY
cat $OUT/next_rows_nofine.json | cut -c1-600
# --- garden: more gate variants
V=$R/nerfshop_amd/csrc/variants
bash tools/ab_bench.sh $OUT/ab_garden_gate2.txt garden_cage base=default gate11=$V/libnrs_gate11.so gate12=$V/libnrs_gate12.so gate11c2=$V/libnrs_gate11c2.so gate11c4=$V/libnrs_gate11c4.so gate12c2=$V/libnrs_gate12c2.so
cd /tmp && export TMPDIR=/tmp
for NAME in gate12 gate11c2 gate11c4 gate12c2; do
  export NRS_LIB_PATH=$V/libnrs_$NAME.so
  D=/tmp/gg_$NAME; rm -rf $D
  timeout 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace -d $D -o b -- python $R/bench.py --workload garden_cage --steps 4 --warmup 1 --no-cpu-baseline --no-extra > $D.log 2>&1
  LINE=$(grep '^{"metric"' $D.log | tail -1)
  echo "{\"variant\": \"$NAME\", \"pmc\": $(python $R/tools/pmc_kernel.py $D), \"bench\": ${LINE:-null}}" >> $OUT/garden_gate2_pmc.jsonl
  rm -rf $D
done
python - <<PY
import json
for l in open("$OUT/garden_gate2_pmc.jsonl"):
    j = json.loads(l); b = j["bench"] or {}; p = j["pmc"]
    n = (b.get("config") or {}).get("samples_per_frame", 1)
    print(j["variant"], "miss/sample %.3f" % (p.get("TCC_MISS_sum", 0) / n), "req/sample %.2f" % (p.get("TCC_REQ_sum", 0) / n), "kernel_ms", (b.get("roofline") or {}).get("kernel_ms"), "value", b.get("value"))
PY
cd $R
NRS_LIB_PATH=$V/libnrs_gate11.so NRS_DEBUG=4 python bench.py --workload garden_cage --steps 1 --warmup 0 --no-extra --no-cpu-baseline 2> $OUT/prof_gate11_garden.err > /dev/null
grep -E "nrs phases" $OUT/prof_gate11_garden.err | head -3
