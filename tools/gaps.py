#!/usr/bin/env python3
"""Timeline of one steady-state step from a rocprofv3 --kernel-trace run: per kernel the busy
time and the idle gap before it (previous kernel's end -> this kernel's start)."""
import csv, glob, sys
from collections import defaultdict
d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].replace("void ", "")[:40] for r in rows]
busy, gap, cnt = defaultdict(float), defaultdict(float), defaultdict(int)
skip = len(rows) // 3  # warm-up
for k in range(max(skip, 1), len(rows)):
    s, e = int(rows[k]["Start_Timestamp"]), int(rows[k]["End_Timestamp"])
    pe = int(rows[k - 1]["End_Timestamp"])
    busy[names[k]] += e - s
    g = s - pe
    if g < 50000:  # ignore host-side pauses between phases
        gap[names[k]] += g
    cnt[names[k]] += 1
tb = tg = 0
for n in busy:
    if cnt[n] < 20:
        continue
    print(f"{n:42s} n={cnt[n]:5d} busy={busy[n]/cnt[n]/1e3:7.2f}us gap_before={gap[n]/cnt[n]/1e3:6.2f}us")
    tb += busy[n] / cnt[n]; tg += gap[n] / cnt[n]
print(f"busy {tb/1e3:.1f} us + gaps {tg/1e3:.1f} us")
