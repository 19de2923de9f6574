mkdir -p gpurun_out/r4h
timeout 300 python tools/gather_probe.py > gpurun_out/r4h/gather_probe.md 2> gpurun_out/r4h/gather_err.log; echo "rc $?"; cat gpurun_out/r4h/gather_probe.md; tail -3 gpurun_out/r4h/gather_err.log
NRS_PROBE_N=1,2,4,8 timeout 600 python tools/scale_probe_r03.py > gpurun_out/r4h/scaling.md 2> gpurun_out/r4h/scale_err.log; grep "^| [1248] " gpurun_out/r4h/scaling.md
timeout 600 python -m pytest tests -m gpu -q -x --timeout=900 -k "comm or lane_teams or dist or bench_multiproc or modes or introspection or parity" > gpurun_out/r4h/tests.log 2>&1; echo "tests rc $?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r4h/tests.log | tail -5
