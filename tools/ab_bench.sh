#!/bin/bash
# A/B of libnrs builds on one box: tools/ab_bench.sh <out file> <workload> <name=path-or-"default"> ...   (run through gpurun)
# Each build runs bench.py twice, interleaved (A B C A B C), to see the box's own noise.
export NRS_DEV_KNOBS=1  # the measurement knobs of libnrs are ignored without it (nrs_internal.h: dev_knob)
out=$1; wl=$2; shift 2
: > $out
for rep in 1 2; do
  for spec in "$@"; do
    name=${spec%%=*}; path=${spec#*=}
    if [ "$path" = "default" ]; then unset NRS_LIB_PATH; else export NRS_LIB_PATH=$path; fi
    line=$(NRS_KERNEL_LOG=1 python bench.py --workload $wl --no-extra --no-cpu-baseline --steps 16 --warmup 3 2> /tmp/ab_err.log | tail -1)
    k=$(grep "nrs kernel" /tmp/ab_err.log | sort | uniq -c | sort -rn | head -1 | sed 's/^ *//')
    echo "$name rep$rep $(echo $line | python -c 'import json,sys; j=json.loads(sys.stdin.read()); print(j["value"], j["roofline"]["kernel_ms"], j["roofline"]["frac"])') | $k" >> $out
  done
done
cat $out
