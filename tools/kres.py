#!/usr/bin/env python3
"""VGPR / SGPR / scratch / LDS of the kernels in a build object (AMDGPU metadata): python3 tools/kres.py gemm [pattern]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
obj = os.path.join(ROOT, "autoforce_amd", "csrc", "build", sys.argv[1] + ".o") if not sys.argv[1].endswith(".o") else sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
with tempfile.TemporaryDirectory() as tmp:
    fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", obj])
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}",
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
    notes = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co], text=True)
cur, rows = None, {}
for line in notes.splitlines():
    m = re.match(r"\s*\.name:\s+(\S+)", line)
    if m:
        cur = m.group(1); rows.setdefault(cur, {})
    m = re.match(r"\s*\.(private_segment_fixed_size|vgpr_count|sgpr_count|group_segment_fixed_size|agpr_count):\s+(\d+)", line)
    if m and cur:
        rows[cur][m.group(1)] = int(m.group(2))
for k, v in sorted(rows.items()):
    if pat in k and v:
        print(f"{k[:70]:70s} vgpr={v.get('vgpr_count')} agpr={v.get('agpr_count')} sgpr={v.get('sgpr_count')} scratch={v.get('private_segment_fixed_size')} lds={v.get('group_segment_fixed_size')}")
