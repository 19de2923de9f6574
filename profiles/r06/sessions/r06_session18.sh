#!/bin/bash
# round 6, GPU session 18: the L2 phase gate with three / four phases (levels 10..15 / 8..15 hashed without records): parity of the knee configuration + A/B
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_s18
mkdir -p $OUT
export NRS_DEV_KNOBS=1
cd $R
V=$R/nerfshop_amd/csrc/variants
( time NRS_L2_GATE=2 NRS_LIB_PATH=$V/libnrs_gate4.so timeout 900 python -m pytest tests/test_gpu_bench_parity.py -x -q -m gpu -k "garden" ) > $OUT/tests_gate4.log 2>&1
tail -3 $OUT/tests_gate4.log
for W in garden_cage garden_cage_norecords garden_cage_records64; do
  : > $OUT/ab_gate_phases_$W.txt
  for rep in 1 2; do
    for spec in "base|1|default" "gate3|2|$V/libnrs_gate3.so" "gate4|2|$V/libnrs_gate4.so"; do
      name=${spec%%|*}; rest=${spec#*|}; mode=${rest%%|*}; path=${rest#*|}
      if [ "$path" = "default" ]; then unset NRS_LIB_PATH; else export NRS_LIB_PATH=$path; fi
      line=$(NRS_L2_GATE=$mode NRS_KERNEL_LOG=1 python bench.py --workload $W --no-extra --no-cpu-baseline --steps 16 --warmup 3 2> /tmp/ab_err.log | tail -1)
      k=$(grep "nrs kernel" /tmp/ab_err.log | sort | uniq -c | sort -rn | head -1 | sed 's/^ *//')
      echo "$name rep$rep $(echo $line | python -c 'import json,sys; j=json.loads(sys.stdin.read()); print(j["value"], j["roofline"]["kernel_ms"], j["roofline"]["frac"])') | $k" >> $OUT/ab_gate_phases_$W.txt
    done
  done
  cat $OUT/ab_gate_phases_$W.txt
done
