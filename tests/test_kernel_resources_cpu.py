"""The hot kernels must keep their register budget without spilling: a change that is neutral in the source can push a
kernel at its VGPR limit into scratch (the compile-time species count took rows16_kernel<3,3,4> from 36 B to 1172 B of
scratch per lane and 8 ms per update step before anything measured it).  Reads the AMDGPU metadata of the objects that
autoforce_amd/csrc/build.sh leaves in csrc/build/ (no GPU needed); skipped when the objects or the LLVM tools are absent."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
OBJ = os.path.join(ROOT, "autoforce_amd", "csrc", "build")

# kernel-name pattern -> (largest scratch in bytes per lane, largest VGPR count)
LIMITS = {
    r"nl_fwd_kernelILi3ELi3ELi[1-4]E": (0, 128),
    r"desc_rev_kernelILi3ELi3ELi[1-4]ELb1ELb0ELb0E": (24, 128),      # (the predict path's gather form: five dword spills)
    r"rows16_kernelILi3ELi3ELi[1-4]E": (40, 256),
    r"finalize_next_kernelILi[12]E": (0, 96),
    r"finalize_gather_kernel": (0, 64),
    r"gemm_nt_kernel8ILi1E": (0, 128),
    r"gemm_nt_kernelILi4ELi1ELi16E": (0, 128),
    r"gemm_nt_kernelILi5ELi1ELi16E": (0, 128),      # (the fused K_nm + W + covloss launch)
    r"tsqr_leaf_wave_kernel": (0, 128),
    r"gemm_nt_kernel8r64x3ILi[0-4]E": (0, 80),      # (three workgroups of 512 per CU: six waves per SIMD)
    r"gemm_nt_kernel4hILi[0-4]E": (0, 128),
}


def _metadata(obj, tmp):
    fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", obj])
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}",
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
    notes = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co], text=True)
    out, name = {}, None
    for line in notes.splitlines():
        m = re.match(r"\s*\.name:\s+(\S+)", line)
        if m:
            name = m.group(1)
            out.setdefault(name, {})
        m = re.match(r"\s*\.(private_segment_fixed_size|vgpr_count):\s+(\d+)", line)
        if m and name:
            out[name][m.group(1)] = int(m.group(2))
    return out


@pytest.mark.skipif(not (os.path.isdir(OBJ) and os.path.isfile(os.path.join(LLVM, "llvm-readelf"))), reason="no build objects / LLVM tools")
def test_hot_kernels_do_not_spill(tmp_path):
    meta = {}
    for f in ("descriptor.o", "api.o", "gemm.o", "tsqr.o"):
        p = os.path.join(OBJ, f)
        if os.path.isfile(p):
            d = tmp_path / f
            d.mkdir()
            meta.update(_metadata(p, str(d)))
    if not meta:
        pytest.skip("no kernel metadata found")
    seen = 0
    for pat, (scratch, vgpr) in LIMITS.items():
        for name, md in meta.items():
            if re.search(pat, name) and "private_segment_fixed_size" in md:
                seen += 1
                assert md["private_segment_fixed_size"] <= scratch, (name, md)
                assert md.get("vgpr_count", 0) <= vgpr, (name, md)
    assert seen >= 8, sorted(meta)[:20]
