#!/bin/bash
# round 6, GPU session 3: the garden frame under L2 time-multiplexing (phase gate) and streaming record loads (nt): Gsamples/s and L2 misses per sample per variant; the rest of the parity net
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_s3
mkdir -p $OUT
export NRS_DEV_KNOBS=1
cd $R
V=$R/nerfshop_amd/csrc/variants
bash tools/ab_bench.sh $OUT/ab_garden_gate.txt garden_cage base=default gate9=$V/libnrs_gate9.so gate10=$V/libnrs_gate10.so gate11=$V/libnrs_gate11.so ntb=$V/libnrs_ntb.so gate10ntb=$V/libnrs_gate10ntb.so
cd /tmp && export TMPDIR=/tmp
for NAME in base gate9 gate10 gate11 ntb gate10ntb; do
  if [ "$NAME" = "base" ]; then unset NRS_LIB_PATH; else export NRS_LIB_PATH=$V/libnrs_$NAME.so; fi
  D=/tmp/gg_$NAME; rm -rf $D
  timeout 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace -d $D -o b -- python $R/bench.py --workload garden_cage --steps 4 --warmup 1 --no-cpu-baseline --no-extra > $D.log 2>&1
  LINE=$(grep '^{"metric"' $D.log | tail -1)
  echo "{\"variant\": \"$NAME\", \"pmc\": $(python $R/tools/pmc_kernel.py $D), \"bench\": ${LINE:-null}}" >> $OUT/garden_gate_pmc.jsonl
  rm -rf $D
done
unset NRS_LIB_PATH
python - <<PY
import json
for l in open("$OUT/garden_gate_pmc.jsonl"):
    j = json.loads(l); b = j["bench"] or {}; p = j["pmc"]
    n = (b.get("config") or {}).get("samples_per_frame", 1)
    print(j["variant"], "miss/sample %.3f" % (p.get("TCC_MISS_sum", 0) / n), "req/sample %.2f" % (p.get("TCC_REQ_sum", 0) / n), "kernel_ms", (b.get("roofline") or {}).get("kernel_ms"), "value", b.get("value"))
PY
# a lego check of the gate variants (they must not change anything there: levels 12..15 hashed too, the L2 is not the problem)
cd $R
bash tools/ab_bench.sh $OUT/ab_lego_gate.txt lego_cage base=default gate10=$V/libnrs_gate10.so
( time timeout 1200 python -m pytest tests/test_gpu_bench_parity.py -x -q -m gpu -s -k "varied or tcnn or membrane or 3layer or garden" ) > $OUT/bench_parity_rest.log 2>&1
tail -5 $OUT/bench_parity_rest.log
