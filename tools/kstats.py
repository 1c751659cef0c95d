#!/usr/bin/env python3
"""Compact per-kernel table from a rocprofv3 --kernel-trace --stats run directory."""
import csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
mincalls = int(sys.argv[2]) if len(sys.argv) > 2 else 0
keep = [r for r in rows if int(r["Calls"]) >= mincalls]
tot = 0.0
for r in keep:
    name = r["Name"].split("(")[0].replace("void ", "")[:44]
    print(f"{name:46s} calls={int(r['Calls']):5d} avg={float(r['AverageNs'])/1e3:8.2f}us min={int(r['MinNs'])/1e3:8.2f} max={int(r['MaxNs'])/1e3:8.2f}")
    tot += float(r["AverageNs"]) / 1e3
print(f"sum of averages: {tot:.1f} us")
