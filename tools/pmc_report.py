"""Summarise rocprofv3 --pmc sqlite outputs: python tools/pmc_report.py <dir-with-*_results.db> [kernel-substring] [min_us]"""
import sqlite3, sys, glob, collections
root = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else "conv_"; min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 20
max_us = float(sys.argv[4]) if len(sys.argv) > 4 else 1e9
for f in sorted(glob.glob(root + "/**/*_results.db", recursive=True)):
    con = sqlite3.connect(f); cur = con.cursor()
    try:
        rows = list(cur.execute("select dispatch_id, kernel_name, counter_name, value, duration from counters_collection where kernel_name like ?", ("%" + pat + "%",)))
    except Exception as e:
        continue
    by = collections.defaultdict(dict)
    for d, k, c, v, du in rows:
        by[d][c] = by[d].get(c, 0) + v; by[d]["_dur"] = du; by[d]["_name"] = k
    act = [d for d in by if min_us * 1000 < by[d]["_dur"] < max_us * 1000]
    if not act: continue
    agg = collections.defaultdict(list)
    for d in act:
        for c, v in by[d].items():
            if not c.startswith("_"): agg[c].append(v)
    n = len(act); dur = sum(by[d]["_dur"] for d in act) / n / 1000
    print("%s: %d dispatches of %s..., avg %.1f us" % (f.split("/")[-2], n, by[act[0]]["_name"][:60], dur))
    for c, v in sorted(agg.items()): print("   %-24s %14.0f" % (c, sum(v) / len(v)))
