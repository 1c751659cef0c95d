#!/bin/bash
# The -DSGPR_PHASE_STAMPS variant of the library (per-wave s_memtime stamps in nl_fwd / desc_rev): autoforce_amd/libsgpr_hip_stamps.so,
# objects in autoforce_amd/csrc/build_st/.  Use: SGPR_HIP_LIB=$PWD/autoforce_amd/libsgpr_hip_stamps.so SGPR_STAMPS=1 python tools/stamps_share.py
set -e
cd "$(dirname "$0")/../autoforce_amd/csrc"
mkdir -p build_st
for f in api descriptor neighbor gemm linalg tsqr; do
  X=""; [ $f = gemm -o $f = tsqr ] && X="-mllvm -amdgpu-mfma-vgpr-form=1"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -DSGPR_PHASE_STAMPS $X -c $f.hip -o build_st/$f.o 2>/dev/null &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libsgpr_hip_stamps.so build_st/*.o -ldl
echo "built $(realpath ../libsgpr_hip_stamps.so)"
