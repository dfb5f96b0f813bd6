#!/usr/bin/env python3
"""Copies what a reader of profiles/ needs out of the scratch output of tools/prof_round.sh (gpurun_out/prof_<tag>/):
    python tools/collect_profiles.py <tag> <name> <round dir>      e.g.  collect_profiles.py r04_vga vga profiles/r04
-> <name>_bench.json, <name>_bench_under_rocprof.json, <name>_kernel_stats.csv, <name>_kernel_stats_in_line.csv,
   <name>_pmc_{fetch,write,valu}.txt (per-kernel means of each counter pass)."""
import collections
import csv
import glob
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
tag, name, dst = sys.argv[1], sys.argv[2], ROOT / sys.argv[3]
src = ROOT / "gpurun_out" / ("prof_" + tag)
dst.mkdir(parents=True, exist_ok=True)


def last_json_line(p):
    lines = [l for l in Path(p).read_text().splitlines() if l.startswith('{"metric"')]
    return lines[-1] + "\n"


(dst / (name + "_bench.json")).write_text(last_json_line(src / "bench.json"))
(dst / (name + "_bench_under_rocprof.json")).write_text(last_json_line(src / "bench_under_rocprof.json"))
for sub, out in (("stats", "_kernel_stats.csv"), ("stats_inline", "_kernel_stats_in_line.csv")):
    f = glob.glob(str(src / sub / "**" / "*kernel_stats.csv"), recursive=True)
    if f:
        shutil.copy(f[0], dst / (name + out))
for pas in ("fetch", "write", "valu"):
    f = glob.glob(str(src / pas / "**" / "*counter_collection.csv"), recursive=True)
    if not f:
        continue
    acc, seen = collections.OrderedDict(), collections.defaultdict(set)
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"][:70]
        acc.setdefault(k, collections.defaultdict(float))[r["Counter_Name"]] += float(r["Counter_Value"])
        seen[k].add(r["Dispatch_Id"])
    with open(dst / ("%s_pmc_%s.txt" % (name, pas)), "w") as o:
        for k, d in acc.items():
            n = len(seen[k])
            o.write("%s launches %d %s\n" % (k, n, {c: int(v / n) for c, v in sorted(d.items())}))
print("collected", tag, "->", dst)
