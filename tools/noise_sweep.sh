# usage: tools/noise_sweep.sh -> gpurun_out/noise_sweep.txt : the varied-opacity workload at several density-noise amplitudes (Gsamples/s, lanes busy, wave balance)
export NRS_DEV_KNOBS=1  # the measurement knobs of libnrs are ignored without it (nrs_internal.h: dev_knob)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/noise_sweep.txt
: > $OUT
for N in 0 0.5 1.5 3.0; do
  V=$(NRS_BENCH_NOISE=$N python $R/bench.py --workload lego_cage_varied --steps 16 --warmup 3 --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['samples_per_frame'], d['roofline']['frac'])")
  D=$(NRS_BENCH_NOISE=$N NRS_DEBUG=4 python $R/bench.py --workload lego_cage_varied --steps 2 --warmup 1 --no-extra --no-cpu-baseline 2>&1 | grep -E "nrs (phases|walk)" | tail -2 | sed -n 's/.*mean wave lifetime = \([0-9.]*\)%.*/life \1/p; s/.*live lanes per round \([0-9.]*\).*/lanes \1/p' | tr '\n' ' ')
  echo "noise $N : $V : $D" | tee -a $OUT
done
