#!/usr/bin/env bash
# Build libformation_hip.so for MI355X (gfx950).  hipcc cross-compiles without a GPU.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
OUT="$HERE/../lib"
mkdir -p "$OUT"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
# -fapprox-func: sqrtf / __expf / __logf / __sinf / __cosf lower to the hardware instructions (v_sqrt_f32, v_exp_f32,
#   v_log_f32, ~1 ulp) without the math library's fix-up code for denormal-range arguments and results.  On this path a
#   denormal can only be exp(-|x|) of a pair far outside contact (a penetration below 1e-38).  NaN / inf behaviour is
#   untouched (no -ffinite-math-only): coincident agents still give NaN as in core.py:312.  9 x 4096 rollout: -11 %.
# -amdgpu-kernarg-preload-count=16: the first 16 dwords of a kernel's arguments arrive in SGPRs at wave start
#   (fg::step_kernel lists its state pointers first for this; see fg_step_kernel.hpp).
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on -fapprox-func -mllvm -amdgpu-kernarg-preload-count=16 -fPIC -shared \
    -I"$ROOT/include" ${FG_EXTRA_FLAGS:-} \
    -o "$OUT/libformation_hip.so" "$HERE/formation_hip.hip"
echo "built $OUT/libformation_hip.so"
# Parity-mode build of the SAME step-kernel source in fp64 (tests only, never loaded by the product): see
# formation_hip_f64.hip.  No -fapprox-func: full-precision exp / log / sqrt.
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on -DFG_F64=1 -fPIC -shared \
    -I"$ROOT/include" -o "$OUT/libformation_hip_f64.so" "$HERE/formation_hip_f64.hip"
echo "built $OUT/libformation_hip_f64.so"
