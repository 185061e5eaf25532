"""Condense the rocprofv3 CSV outputs of tools/profile_bench.sh into small files fit for profiles/:
   <tag>_kernel_stats.csv   per-kernel calls / total / average duration (from --kernel-trace --stats)
   <tag>_traffic.json       per-launch HBM-side bytes of the stencil kernels from the PMC passes,
                            corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950:
                            FETCH_SIZE and WRITE_SIZE are in KiB, and FETCH_SIZE counts 128-B requests
                            as 64 B for wide coalesced reads (x2).
"""
import csv, glob, json, os, sys, collections

root, tag = sys.argv[1], sys.argv[2]


def find(pattern):
    hits = glob.glob(os.path.join(root, "**", pattern), recursive=True)
    return hits[0] if hits else None


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return n[:n.index("(")] if "(" in n else n


stats = find("trace/**/*kernel_stats.csv") or find("*kernel_stats.csv")
rows = []
if stats:
    with open(stats) as f:
        for r in csv.DictReader(f):
            rows.append(r)
    with open(os.path.join(root, tag + "_kernel_stats.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ns", "avg_ns", "pct", "min_ns", "max_ns"])
        for r in rows:
            w.writerow([short(r.get("Name", "")), r.get("Calls"), r.get("TotalDurationNs"), r.get("AverageNs"),
                        r.get("Percentage"), r.get("MinNs"), r.get("MaxNs")])


def counter_avg(sub, counter):
    f = find(os.path.join(sub, "**", "*counter_collection.csv"))
    if not f:
        return {}
    acc = collections.defaultdict(lambda: [0.0, 0])
    seen = collections.defaultdict(float)
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if r.get("Counter_Name") != counter:
                continue
            key = (short(r["Kernel_Name"]), r["Dispatch_Id"])
            seen[key] += float(r["Counter_Value"])
    for (k, _), v in seen.items():
        acc[k][0] += v
        acc[k][1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items()}


fetch = counter_avg("pmc_fetch", "FETCH_SIZE")
write = counter_avg("pmc_write", "WRITE_SIZE")
traffic = {}
for k in sorted(set(fetch) | set(write)):
    fk, n = fetch.get(k, (0.0, 0))
    wk, _ = write.get(k, (0.0, 0))
    traffic[k] = dict(launches=n, fetch_size_kib=round(fk, 1), write_size_kib=round(wk, 1),
                      hbm_bytes_per_launch=int((2 * fk + wk) * 1024))
sq = {}
for c in ("SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE",
          "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE"):
    for k, (v, n) in counter_avg("pmc_sq", c).items():
        sq.setdefault(k, {})[c] = round(v)
# per-kernel time of the traced run beside the counters (bench.py's roofline.issue divides one by the other), and the clock
# the long kernels ran at: GRBM_GUI_ACTIVE is summed over the 8 XCDs
times = {short(r.get("Name", "")): dict(calls=int(r.get("Calls") or 0), total_ns=int(r.get("TotalDurationNs") or 0)) for r in rows}
clock = None
try:
    num = den = 0.0
    for k, v in sq.items():
        t = times.get(k)
        if t and t["calls"] and v.get("GRBM_GUI_ACTIVE") and t["total_ns"] / t["calls"] > 20000:
            num += v["GRBM_GUI_ACTIVE"] / 8.0 * t["calls"]; den += t["total_ns"]
    clock = round(num / den, 3) if den else None
except Exception:
    clock = None
import hashlib, subprocess
def conv_sources_hash():
    """sha256 over the reblurring pass's sources: bench.py refuses HBM-traffic numbers taken from other code"""
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(here, "polyblur_amd", "csrc", "conv*"))) + glob.glob(os.path.join(here, "polyblur_amd", "csrc", "khat.h")):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]
try:
    sha = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or os.environ.get("PB_GIT_SHA", "?")
except Exception:
    sha = os.environ.get("PB_GIT_SHA", "?")
json.dump(dict(tag=tag, git=os.environ.get("PB_GIT_SHA", sha), conv_sources_sha256_16=conv_sources_hash(), note="per-launch averages over every launch of the profiled bench.py run; "
                             "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 correction)",
               traffic=traffic, sq=sq, time=times, clock_ghz=clock), open(os.path.join(root, tag + "_traffic.json"), "w"), indent=1)
print(open(os.path.join(root, tag + "_kernel_stats.csv")).read() if stats else "no stats csv found")
print(json.dumps({k: v for k, v in traffic.items() if "conv" in k}, indent=1))
