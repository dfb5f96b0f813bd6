"""Timeline of the kernels of a few consecutive pipelined frames from a rocprofv3 --kernel-trace csv (debug aid)."""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:48], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
n = len(rows)
lo = int(sys.argv[2]) if len(sys.argv) > 2 else n - 160
t0 = rows[lo][0]
busy = 0
prev_end = t0
for s, e, name, q, st in rows[lo:lo + 150]:
    print("%9.1f %9.1f  %7.1f us  q%-3s s%-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, st, name))
