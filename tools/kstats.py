"""Print the rocprofv3 --kernel-trace --stats table (csv) found under a directory: python tools/kstats.py <dir> [rows]"""
import csv, glob, os, sys
root = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
hits = glob.glob(os.path.join(root, "**", "*kernel_stats.csv"), recursive=True)
for r in list(csv.DictReader(open(hits[0])))[:n]:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print("%-60s calls %5s  avg %9.1f us  %5s%%" % (name[:60], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
