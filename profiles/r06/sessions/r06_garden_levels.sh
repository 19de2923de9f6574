#!/bin/bash
# the L2 misses of the garden frame attributed level pair by level pair.  For each mask a rocprofv3 PMC pass (TCC hit / miss / req) over a short
# bench run of garden_cage_records64 (64 GiB of brick records: levels 8..11) with NRS_SKIP_PAIRS=mask (level pairs that are not gathered at all: nrs_mlp.cuh KIND_SKIP).  usage: profiles/r06/sessions/r06_garden_levels.sh <out dir> [masks...]
export NRS_DEV_KNOBS=1
R=$GRAFT_REPO_ROOT
OUT=$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
MASKS="$@"
[ -z "$MASKS" ] && MASKS="0x00 0x01 0x02 0x04 0x08 0x10 0x20 0x40 0x80 0xfe 0xfd 0xfb 0xf7 0xef 0xdf 0xbf 0x7f 0x0f 0xf0 0x3f 0xc0"
for M in $MASKS; do
  D=/tmp/gl_$M; rm -rf $D
  NRS_SKIP_PAIRS=$M timeout 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace -d $D -o b -- \
     python $R/bench.py --workload garden_cage_records64 --steps 4 --warmup 1 --no-cpu-baseline --no-extra > $D.log 2>&1
  LINE=$(grep '^{"metric"' $D.log | tail -1)
  echo "{\"mask\": \"$M\", \"pmc\": $(python $R/tools/pmc_kernel.py $D), \"bench\": ${LINE:-null}}" >> $OUT/levels.jsonl
  rm -rf $D
done
cat $OUT/levels.jsonl | python -c '
import json, sys
for l in sys.stdin:
    j = json.loads(l); b = j["bench"] or {}; p = j["pmc"]
    n = (b.get("config") or {}).get("samples_per_frame", 0)
    print(j["mask"], "samples", n, "miss/sample %.3f" % (p.get("TCC_MISS_sum", 0) / max(n, 1)), "req/sample %.2f" % (p.get("TCC_REQ_sum", 0) / max(n, 1)), "kernel_ms", (b.get("roofline") or {}).get("kernel_ms"), "value", b.get("value"))
'
