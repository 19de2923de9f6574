V=$GRAFT_REPO_ROOT/nerfshop_amd/csrc/variants/libnrs_team8end.so
bash tools/ab_bench.sh gpurun_out/ab_team8end_lego.txt lego_cage t4=default t8end=$V > /dev/null 2>&1 < /dev/null; cat gpurun_out/ab_team8end_lego.txt
for rep in 1 2; do
  timeout 100 python tools/small_launch_probe.py 8 2>&1 < /dev/null | grep "share:" | sed 's/^/default /'
  NRS_LIB_PATH=$V timeout 100 python tools/small_launch_probe.py 8 2>&1 < /dev/null | grep "share:" | sed 's/^/team8end /'
done
NRS_LIB_PATH=$V timeout 200 python -m pytest tests/test_gpu_lane_teams.py -q -x 2>&1 < /dev/null | tail -1
