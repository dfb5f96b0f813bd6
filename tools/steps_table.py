"""Per-step start/end of each stage out of a rocprofv3 kernel trace (results .db): which dependency paces the pipeline."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = [(r[0], int(r[1]), int(r[2]), r[3]) for r in cur.execute(
    "select s.display_name, d.start, d.end, d.stream_id from %s d join %s s on d.kernel_id = s.id order by d.start" % (kd, ks))]
per_step_resize = int(sys.argv[2]) if len(sys.argv) > 2 else 46
groups = {"P": [], "Ffull": [], "Fhalf": [], "blur": [], "sel256": [], "sel64": [], "desc": [], "knn_main": [], "tail_first": [], "tail_last": []}
rs = []
main_stream = None
for n, a, b, st in rows:
    if "resize_strip" in n or "pyramid_image" in n: rs.append((a, b))
    elif "fast_march" in n and "<false" in n: groups["Ffull"].append((a, b))
    elif "fast_march" in n and "<true" in n: groups["Fhalf"].append((a, b))
    elif "blur_mma" in n: groups["blur"].append((a, b))
    elif "orb_select_kernel<256" in n: groups["sel256"].append((a, b)); main_stream = st
    elif "orb_select_kernel<64" in n: groups["sel64"].append((a, b))
    elif "describe" in n: groups["desc"].append((a, b))
    elif "knn2" in n and st == main_stream: groups["knn_main"].append((a, b))
    elif "stereo_residual" in n: groups["tail_first"].append((a, b))
    elif "pack_copy" in n: groups["tail_last"].append((a, b))
for i in range(0, len(rs) - per_step_resize + 1, per_step_resize):
    g = rs[i:i + per_step_resize]
    groups["P"].append((g[0][0], max(x[1] for x in g)))
t0 = groups["Ffull"][0][0]
n = min(len(v) for v in groups.values() if v)
print("step " + " ".join("%-17s" % k for k in groups))
for i in range(n):
    print("%4d " % i + " ".join(("%7.2f-%-7.2f  " % ((v[i][0] - t0) / 1e6, (v[i][1] - t0) / 1e6)) if i < len(v) else " " * 18 for v in groups.values()))
