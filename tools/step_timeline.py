"""The kernels of the last timed step of a traced bench run, with the gaps between them: rocprofv3 --kernel-trace --output-format csv -d DIR
-- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-context --settle-ms 0; python tools/step_timeline.py DIR"""
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ours = [r for r in rows if any(s in r["Kernel_Name"] for s in ("gray_rows", "cols_fixed", "blur_params", "conv_w", "grad_cols", "gray_minmax"))]
# a step = 3 iterations x (gray_rows, cols, params, w128, wfft) = 15 launches: take the last 15
last = ours[-15:]
t0 = int(last[0]["Start_Timestamp"]); prev_end = None
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:40]
    print("%-40s start %7.1f us  dur %6.1f us  gap before %5.1f us" % (name, (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3 if prev_end else 0.0))
    prev_end = e
print("step span %.1f us; kernels %.1f us; gaps %.1f us" % ((prev_end - t0) / 1e3, sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last) / 1e3,
      (prev_end - t0) / 1e3 - sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last) / 1e3))
