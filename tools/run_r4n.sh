bash tools/ab_env.sh gpurun_out/ab_center_out2.txt lego_cage "rowmajor=NRS_CENTER_OUT=0" "outsidein=NRS_CENTER_OUT=2" > /dev/null 2>&1 < /dev/null; cat gpurun_out/ab_center_out2.txt
for co in 0 2; do
  NRS_CENTER_OUT=$co timeout 100 python tools/small_launch_probe.py 8 2>&1 < /dev/null | grep "share:" | sed "s/^/center_out=$co /"
done
