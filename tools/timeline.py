"""Prints a per-step kernel timeline out of a rocprofv3 kernel trace (csv or the results .db)."""
import csv, sqlite3, sys, os

def load(path):
    if path.endswith(".csv"):
        rows = list(csv.DictReader(open(path)))
        return [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Stream_Id"], r["Queue_Id"]) for r in rows]
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    q = "select s.display_name, d.start, d.end, d.stream_id, d.queue_id from %s d join %s s on d.kernel_id = s.id" % (kd, ks)
    return [(r[0], int(r[1]), int(r[2]), str(r[3]), str(r[4])) for r in cur.execute(q)]

rows = load(sys.argv[1])
rows.sort(key=lambda r: r[1])
first = int(sys.argv[2]) if len(sys.argv) > 2 else 3
nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "")[:44]
idx = [i for i, r in enumerate(rows) if "fast_march_kernel<false" in r[0] or "fast_march_resident_kernel<false" in r[0]]
s, e = idx[first], idx[min(first + nsteps, len(idx) - 1)]
t0 = rows[s][1]
agg = None
for r in rows[max(s - 40, 0):e]:
    if "resize_strip" in r[0]:
        if agg is None:
            agg = [r[1], r[2], 1]
        else:
            agg[1] = max(agg[1], r[2]); agg[2] += 1
        continue
    if agg:
        print("%9.1f %8.1f  %-44s x%d" % ((agg[0] - t0) / 1e3, (agg[1] - agg[0]) / 1e3, "resize_strip (span)", agg[2])); agg = None
    print("%9.1f %8.1f  %-44s s%s q%s" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, short(r[0]), r[3], r[4]))
