# usage: bq2.sh <workload> -- one short bench line: value ms frac
python bench.py --workload $1 --steps 8 --warmup 2 --no-extra --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BENCH', '$1', d['value'], d['ms_per_step'], d['roofline']['frac'], d['config']['cell_records'])"
