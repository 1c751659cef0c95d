"""From a rocprofv3 kernel trace of examples/md_nvt_config5.py: the TSQR launches of the LAST update step, by kernel and grid
size (how many launches, mean duration), in launch order for one panel.  usage: python3 tools/trace_tsqr.py <dir>"""
import collections, csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
# the last long rows16_kernel marks the last data trial
idx = [i for i, r in enumerate(rows) if "rows16_kernel" in r["Kernel_Name"] and dur(r) > 3000]
i0 = idx[-1]
seg = rows[i0:i0 + 2500]
# cut at the next finalize_next / finalize_gather (end of the update)
end = next((k for k, r in enumerate(seg) if k > 50 and ("finalize_next" in r["Kernel_Name"] or "finalize_gather" in r["Kernel_Name"])), len(seg))
seg = seg[:end]
agg = collections.defaultdict(list)
gk = [k for k in rows[0].keys() if "Grid" in k and k.endswith("X")] or [k for k in rows[0].keys() if "Grid" in k]
wk = [k for k in rows[0].keys() if "Workgroup" in k and k.endswith("X")] or [k for k in rows[0].keys() if "Workgroup" in k]
for r in seg:
    g = int(r[gk[0]]) // max(int(r[wk[0]]), 1) if gk and wk else -1
    agg[(name(r), g)].append(dur(r))
print(f"{len(seg)} launches after the trial's rows16_kernel, {sum(sum(v) for v in agg.values()) / 1e3:.1f} ms busy, span {(int(seg[-1]['End_Timestamp']) - int(seg[0]['Start_Timestamp'])) / 1e6:.1f} ms")
for (n, g), v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:40]:
    print(f"  {n:42s} grid {g:6d}  {len(v):4d} x {sum(v) / len(v):8.1f} us = {sum(v) / 1e3:7.2f} ms")
print("first 60 launches in order:")
t0 = int(seg[0]["Start_Timestamp"])
for r in seg[:60]:
    g = int(r[gk[0]]) // max(int(r[wk[0]]), 1) if gk and wk else -1
    print(f"   +{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us  {dur(r):8.1f} us  grid {g:6d}  {name(r)}")
