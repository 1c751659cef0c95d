#!/bin/bash
# Builds libsgpr_hip.so for gfx950 (cross-compiles without a GPU).
set -e
cd "$(dirname "$0")"
OUT=../libsgpr_hip.so
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function -Wno-unused-variable"
mkdir -p build
pids=()
for f in api descriptor neighbor gemm linalg tsqr; do
  stale=0
  [ -f build/$f.o ] || stale=1
  for dep in $f.hip *.inc *.h ../../include/sgpr_hip.h; do [ $dep -nt build/$f.o ] && stale=1; done   # (every include: peer.inc, bandqr.inc, ...)
  [ "${EXTRA_FLAGS}" != "$(cat build/$f.flags 2>/dev/null)" ] && stale=1
  if [ $stale = 1 ]; then
    echo "${EXTRA_FLAGS}" > build/$f.flags
    # gemm, tsqr: MFMA accumulators stay in VGPRs (the AGPR form copies them in and out around every trip of a loop)
    X=""; [ $f = gemm -o $f = tsqr ] && X="-mllvm -amdgpu-mfma-vgpr-form=1"
    $HIPCC $FLAGS $X ${EXTRA_FLAGS} -c $f.hip -o build/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o $OUT build/api.o build/descriptor.o build/neighbor.o build/gemm.o build/linalg.o build/tsqr.o -ldl
echo "built $(realpath $OUT)"
