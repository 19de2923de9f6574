mkdir -p gpurun_out/r4k
for t in 0 -2 -3 -4; do
  echo "== NRS_TEAM=$t" >> gpurun_out/r4k/sweep.txt
  NRS_TEAM=$t NRS_PROBE_N=4,8 timeout 300 python tools/scale_probe_r03.py 2>/dev/null | grep "^| [48] " >> gpurun_out/r4k/sweep.txt
done
cat gpurun_out/r4k/sweep.txt
