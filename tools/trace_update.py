"""From a rocprofv3 kernel trace of examples/md_nvt_config5.py: the kernels of ONE model-update step at the size limit
(from the rows16_kernel of a pushed frame back to the previous prediction step and on to the next one).
usage: python3 tools/trace_update.py <dir with *kernel_trace.csv>"""
import collections, csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"]
# an update = a stretch between two finalize_gather kernels that contains a rows16_kernel with many chunks (long one)
fg = [i for i, r in enumerate(rows) if "finalize_gather" in name(r) or "finalize_next" in name(r)]
best = None
for a, b in zip(fg, fg[1:]):
    seg = rows[a + 1:b + 1]
    if any("rows16_kernel" in name(r) and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 3e6 for r in seg):
        best = seg  # keep the last such stretch
seg = best
t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
agg = collections.defaultdict(list)
for r in seg:
    agg[name(r)[:58]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in agg.values())
print(f"update step: {len(seg)} launches, GPU busy {tot / 1e3:.1f} ms of a {(t1 - t0) / 1e6:.1f} ms span")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:22]:
    print(f"  {k:60s} {len(v):5d} x {sum(v) / len(v):8.1f} us = {sum(v) / 1e3:7.2f} ms")
# idle gaps > 200 us
prev = t0
gaps = []
for r in seg:
    s = int(r["Start_Timestamp"])
    if s - prev > 200e3:
        gaps.append(((prev - t0) / 1e6, (s - prev) / 1e6, name(r)[:40]))
    prev = max(prev, int(r["End_Timestamp"]))
print("idle gaps > 0.2 ms (at ms, length ms, next kernel):")
for g in gaps:
    print("   %.1f  %.2f  %s" % g)
