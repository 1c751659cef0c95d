"""From a rocprofv3 kernel trace of tools/qr_prof.py: the LAST stretch of tsqr kernels (one factorisation) — its span, the
busy time per stream/queue, and how much of the leaves' time ran beside an update.  usage: qr_overlap.py <trace dir>"""
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"))[-1]
rows = [r for r in csv.DictReader(open(f)) if "tsqr_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# split into stretches separated by > 2 ms without a tsqr kernel
st, cur, prev = [], [], None
for r in rows:
    s = int(r["Start_Timestamp"])
    if prev is not None and s - prev > 2e6:
        st.append(cur); cur = []
    cur.append(r); prev = int(r["End_Timestamp"])
st.append(cur)
big = [x for x in st if len(x) > 100]
for seg in big[-3:]:
    t0, t1 = int(seg[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in seg)
    per = collections.defaultdict(float)
    for r in seg:
        per[(r["Kernel_Name"].split("(")[0][:24], r.get("Queue_Id", "?"))] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    print(f"{len(seg)} launches, span {(t1 - t0) / 1e6:.2f} ms:", {k: round(v, 2) for k, v in per.items()})
    leaf = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in seg if "leaf" in r["Kernel_Name"]]
    app = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in seg if "apply" in r["Kernel_Name"]]
    ov = 0
    for a, b in leaf:
        for c, d in app:
            if c < b and d > a:
                ov += min(b, d) - max(a, c)
    print(f"   leaf time that overlaps an update: {ov / 1e6:.2f} ms of {sum(b - a for a, b in leaf) / 1e6:.2f} ms")
