#!/usr/bin/env python3
"""Per-launch durations of one kernel family from a rocprofv3 --kernel-trace CSV (tools/prof.sh stats <tag>):
   tools/trace_stage.py gpurun_out/prof_<tag> resize   -> launch order, grid, duration, gap to the previous launch
Only the last bench step's launches are listed."""
import csv, glob, sys
d, pat = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last step = launches after the last knn2 ... simpler: take the last occurrence block of the pattern
idx = [i for i, r in enumerate(rows) if pat in r["Kernel_Name"]]
# split into steps by gaps > 1 ms between consecutive matching launches
steps, cur = [], [idx[0]]
for a, b in zip(idx, idx[1:]):
    if int(rows[b]["Start_Timestamp"]) - int(rows[a]["End_Timestamp"]) > 1000000:
        steps.append(cur)
        cur = []
    cur.append(b)
steps.append(cur)
last = steps[-1]
t0 = int(rows[last[0]]["Start_Timestamp"])
tot = 0
for i in last:
    r = rows[i]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    tot += e - s
    print("%8.1f us  dur %7.1f us  grid %s x %s x %s  q %s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "?")),
          r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""), r.get("Queue_Id", "?"), r["Kernel_Name"][:50]))
print("launches %d, sum of durations %.1f us, span %.1f us" % (len(last), tot / 1e3,
      (int(rows[last[-1]]["End_Timestamp"]) - t0) / 1e3))
