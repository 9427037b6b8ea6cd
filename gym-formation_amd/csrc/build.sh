#!/usr/bin/env bash
# Build libformation_hip.so for MI355X (gfx950).  hipcc cross-compiles without a GPU.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
OUT="$HERE/../lib"
mkdir -p "$OUT"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on -fPIC -shared \
    -I"$ROOT/include" ${FG_EXTRA_FLAGS:-} \
    -o "$OUT/libformation_hip.so" "$HERE/formation_hip.hip"
echo "built $OUT/libformation_hip.so"
