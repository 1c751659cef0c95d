#!/usr/bin/env python3
import csv, glob, sys, collections
import statistics as st
d = sys.argv[1]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:36]
    acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    if max(len(x) for x in v.values()) < 10:
        continue
    print(f"{k:38s} " + " ".join(f"{c}={st.median(x):.4g}" for c, x in sorted(v.items())))
