import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); c = db.cursor()
pat = sys.argv[2] if len(sys.argv) > 2 else '%'
q = "select s.kernel_name, d.start, d.end-d.start from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id=s.id where s.kernel_name like ? order by d.start"
rows = list(c.execute(q, (pat,)))
import collections
by = collections.defaultdict(list)
for n, st, du in rows: by[n].append(du / 1e3)
for n, v in by.items():
    v2 = sorted(v)
    print("%-60s n=%3d min %8.1f med %8.1f us  first8: %s" % (n[17:77], len(v), v2[0], v2[len(v2)//2], " ".join("%.0f" % x for x in v[:8])))
