# usage: bq3.sh <workload> <steps> -- value only, several env variants given as "K=V K=V" lines on stdin
while read -r ENVS; do
  R=$(env $ENVS python bench.py --workload $1 --steps ${2:-12} --warmup 3 --no-extra --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "BENCH $1 [$ENVS] $R"
done
