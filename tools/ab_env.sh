#!/bin/bash
# A/B of ONE library under different environments: tools/ab_env.sh <out> <workload> "NAME=ENVVAR=VALUE" ...  (interleaved, two repetitions)
export NRS_DEV_KNOBS=1  # the measurement knobs of libnrs are ignored without it (nrs_internal.h: dev_knob)
out=$1; wl=$2; shift 2
: > $out
for rep in 1 2; do
  for spec in "$@"; do
    name=${spec%%=*}; kv=${spec#*=}
    line=$(env $kv python bench.py --workload $wl --no-extra --no-cpu-baseline --steps 16 --warmup 3 2> /dev/null | tail -1)
    echo "$name rep$rep $(echo $line | python -c 'import json,sys; j=json.loads(sys.stdin.read()); print(j["value"], j["roofline"]["kernel_ms"], j["roofline"]["frac"])')" >> $out
  done
done
cat $out
