#!/bin/bash
# Register / scratch / occupancy of ONE instantiation of the render kernel, compiled alone (seconds instead of a minute): for register-pressure work.
# usage: tools/one_kernel.sh "render_kernel<8, 4, false, true, false, 0, 0, false>" [extra hipcc flags, e.g. -DNRS_EXP_P=3]   (or render_kernel_c128<...>)
#        add -S as an extra flag to keep the ISA listing in /tmp/one_kernel.s
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
INST="$1"; shift
TMP=$(mktemp -d /tmp/one_kernel.XXXXXX)
cat > "$TMP/k.hip" <<SRC
#define NRS_BODY_ONLY 1
#include "$ROOT/nerfshop_amd/csrc/nrs_kernels.hip"
template __global__ void nrs::$INST(const nrs::DeviceModel, const nrs::RenderArgs);
SRC
OUT=/dev/null
EXTRA=()
for f in "$@"; do if [ "$f" == "-S" ]; then OUT=/tmp/one_kernel.s; EXTRA+=(-S --cuda-device-only); else EXTRA+=("$f"); fi; done
if [ "$OUT" == "/dev/null" ]; then EXTRA+=(-c); fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -I"$ROOT/nerfshop_amd/csrc" -Rpass-analysis=kernel-resource-usage "${EXTRA[@]}" "$TMP/k.hip" -o "$OUT" 2>&1 | python3 "$ROOT/tools/resource_usage.py" render_kernel
rm -rf "$TMP"
