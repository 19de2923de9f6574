V=$GRAFT_REPO_ROOT/nerfshop_amd/csrc/variants
for rep in 1 2; do
  for spec in gen64=default refill32=$V/libnrs_refill32.so refill16=$V/libnrs_refill16.so; do
    name=${spec%%=*}; path=${spec#*=}
    if [ "$path" = "default" ]; then unset NRS_LIB_PATH; else export NRS_LIB_PATH=$path; fi
    line=$(NRS_TEAM=1 timeout 100 python bench.py --workload lego_cage --no-extra --no-cpu-baseline --steps 16 --warmup 3 2> /dev/null < /dev/null | tail -1)
    echo "team1 $name rep$rep $(echo $line | python -c 'import json,sys; j=json.loads(sys.stdin.read()); print(j["value"], j["roofline"]["kernel_ms"])')"
  done
done
unset NRS_LIB_PATH
line=$(timeout 100 python bench.py --workload lego_cage --no-extra --no-cpu-baseline --steps 16 --warmup 3 2> /dev/null < /dev/null | tail -1)
echo "automatic $(echo $line | python -c 'import json,sys; j=json.loads(sys.stdin.read()); print(j["value"], j["roofline"]["kernel_ms"])')"
