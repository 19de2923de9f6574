mkdir -p gpurun_out/r4j
timeout 900 python -m pytest tests/test_gpu_verify_snapshot.py tests/test_gpu_modes.py tests/test_gpu_parity.py -m gpu -q -x --timeout=900 > gpurun_out/r4j/tests.log 2>&1; echo "tests rc $?"
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r4j/tests.log | tail -15
timeout 600 python bench.py --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['cpu_baseline']['value'], j['cpu_baseline']['sample'], j['cpu_baseline']['config1']['value'])"
