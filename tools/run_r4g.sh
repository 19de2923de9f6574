mkdir -p gpurun_out/r4g
V=$GRAFT_REPO_ROOT/nerfshop_amd/csrc/variants
NRS_LIB_PATH=$V/libnrs_brickmorton.so python bench.py --workload garden_cage --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | tail -1 > gpurun_out/r4g/garden_morton_3.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4g/garden_*3.json')):
    j=json.loads(open(f).read())
    r=j['roofline']; s=j['config']['samples_per_frame']
    print(f, j['value'], r['kernel_ms'], r['traffic'], round(r['traffic']/128/s,2) if r['traffic'] else None, r['traffic_source'][:60])
PY
