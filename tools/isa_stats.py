#!/usr/bin/env python3
"""Per-kernel instruction mix from a gfx950 assembly listing (hipcc --offload-device-only -S).
usage: isa_stats.py file.s [name-filter]"""
import re, sys, collections
src = open(sys.argv[1]).read().split("\n")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cur = None
stats = {}
meta = {}
for ln in src:
    m = re.match(r"^(_Z\w+):", ln)
    if m:
        cur = m.group(1); stats[cur] = collections.Counter(); continue
    if cur is None: continue
    s = ln.strip()
    if s.startswith(".end_amdhsa_kernel") or s.startswith(".Lfunc_end"):
        pass
    m = re.match(r"^\s+([a-z_0-9]+)\s", ln)
    if m and not s.startswith("."):
        op = m.group(1)
        c = stats[cur]
        c["total"] += 1
        if op.startswith("v_mfma"): c["mfma"] += 1
        elif op.startswith("v_") and "f64" in op: c["valu_f64"] += 1
        elif op.startswith("v_"): c["valu_other"] += 1
        elif op.startswith("ds_"): c["lds"] += 1
        elif op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_"): c["vmem"] += 1
        elif op.startswith("scratch_"): c["scratch"] += 1
        elif op.startswith("s_waitcnt"): c["waitcnt"] += 1
        elif op.startswith("s_"): c["salu"] += 1
    m = re.match(r"^;\s*(NumVgprs|NumAgprs|ScratchSize|Occupancy|LDSByteSize|NumSgprs|TotalNumVgprs):\s*(\d+)", s)
    if m: meta.setdefault(cur, {})[m.group(1)] = int(m.group(2))
import subprocess
def dem(n):
    try: return subprocess.run(["/usr/bin/c++filt", n], capture_output=True, text=True).stdout.strip()
    except Exception: return n
for k, c in stats.items():
    d = dem(k)
    if flt and flt not in d: continue
    if c["total"] < 20: continue
    print(d[:110])
    print("   ", dict(c), meta.get(k, {}))
