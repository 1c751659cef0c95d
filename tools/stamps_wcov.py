"""Per-tile s_memtime stamps of the two GEMM launches of a predict step (SGPR_STAMPS=1, SGPR_STAMPS_FILE=path):
where the time of a tile goes — tile entry, first loads (prologue), main loop, epilogue — and how the CUs fill.
s_memtime counts from a different origin on every CU: times are taken relative to the first tile start on the same CU."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = "gpurun_out/stamps.txt"
if len(sys.argv) < 2:
    os.environ["SGPR_STAMPS"] = "1"
    os.environ["SGPR_STAMPS_FILE"] = out
    import numpy as np, bench
    from autoforce_amd.workloads import lips
    numbers, pos, cell, pbc = lips(16, seed=0)
    mdl = bench.build_model(0, numbers, pos, cell, pbc, 512)
    for _ in range(5):
        mdl.predict(numbers, pos, cell, pbc)
    mdl.close()
else:
    out = sys.argv[1]
rec = collections.defaultdict(list)
for line in open(out):
    f = line.split()
    k, b = f[0], int(f[1])
    t0, t1, t2, w, te, tp = (int(x) for x in f[2:8])
    if (w & 0xfffff) <= 0:
        continue
    hw = w >> 32
    rec[k].append(dict(b=b, t0=t0, t1=t1, t2=t2, te=te, tp=tp, nst=(w & 0xfffff) // 32, second=(w >> 20) & 1,
                       cu=((w >> 24) & 15, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15)))
for k, rs in rec.items():
    bycu = collections.defaultdict(list)
    for r in rs:
        bycu[r["cu"]].append(r)
    for v in bycu.values():
        o = min(r["t0"] for r in v)
        for r in v:
            for q in ("t0", "t1", "t2", "te", "tp"):
                r[q] -= o
    span = max(r["t2"] for r in rs)
    print(f"== {k}: {len(rs)} tiles on {len(bycu)} CUs, {sum(r['nst'] for r in rs)} stages of 32; longest CU {span} ticks")
    ends = sorted(max(r["t2"] for r in v) for v in bycu.values())
    print("   CU finish (ticks): p10 %d p50 %d p90 %d max %d" % tuple(ends[int(q * (len(ends) - 1))] for q in (0.1, 0.5, 0.9, 1.0)))
    print("   stages per CU min/mean/max = %d/%.1f/%d" % (min(sum(r['nst'] for r in v) for v in bycu.values()),
          sum(r['nst'] for r in rs) / len(bycu), max(sum(r['nst'] for r in v) for v in bycu.values())))
    for rnd, name in ((0, "tiles that start with the launch"), (1, "tiles that start later")):
        sel = [r for r in rs if (r["t0"] > 2000) == bool(rnd)]
        bynst = collections.defaultdict(list)
        for r in sel:
            bynst[(r["second"], r["nst"])].append(r)
        print(f"   {name}: {len(sel)}")
        for key in sorted(bynst):
            v = bynst[key]; n = len(v)
            m = lambda f: sum(f(r) for r in v) / n
            print(f"      problem {key[0]} nst {key[1]:2d}: {n:4d} tiles: start {m(lambda r: r['t0']):6.0f} | entry {m(lambda r: r['te'] - r['t0']):5.0f} | "
                  f"first loads {m(lambda r: r['tp'] - r['te']):5.0f} | loop {m(lambda r: r['t1'] - r['tp']):6.0f} ({m(lambda r: (r['t1'] - r['tp']) / r['nst']):5.0f}/stage) | "
                  f"epilogue {m(lambda r: r['t2'] - r['t1']):5.0f} | total {m(lambda r: r['t2'] - r['t0']):6.0f}")
