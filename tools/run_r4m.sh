V=$GRAFT_REPO_ROOT/nerfshop_amd/csrc/variants/libnrs_team8.so
for rep in 1 2; do
  python tools/small_launch_probe.py 8 2>&1 | grep "share:" | sed 's/^/default /'
  NRS_LIB_PATH=$V python tools/small_launch_probe.py 8 2>&1 | grep "share:" | sed 's/^/team8 /'
done
for rep in 1 2; do
  python tools/small_launch_probe.py 4 2>&1 | grep "share:" | sed 's/^/default /'
  NRS_LIB_PATH=$V python tools/small_launch_probe.py 4 2>&1 | grep "share:" | sed 's/^/team8 /'
done
bash tools/ab_bench.sh gpurun_out/ab_team8_lego.txt lego_cage t4=default t8=$V > /dev/null 2>&1; cat gpurun_out/ab_team8_lego.txt
bash tools/ab_bench.sh gpurun_out/ab_team8_varied.txt lego_cage_varied t4=default t8=$V > /dev/null 2>&1; cat gpurun_out/ab_team8_varied.txt
NRS_LIB_PATH=$V python -m pytest tests/test_gpu_lane_teams.py -q -x 2>&1 | tail -2
