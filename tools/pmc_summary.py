#!/usr/bin/env python3
"""Per-kernel means of a rocprofv3 --pmc pass (pmc_counter_collection.csv) -> one text line per kernel.
    python tools/pmc_summary.py <pass dir>  > profiles/rNN/<config>_pmc_<pass>.txt"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.OrderedDict()
seen = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:70]
    d = acc.setdefault(k, collections.defaultdict(float))
    d[r["Counter_Name"]] += float(r["Counter_Value"])
    seen[k].add(r["Dispatch_Id"])
for k, d in acc.items():
    n = len(seen[k])
    print(k, "launches", n, {c: int(v / n) for c, v in sorted(d.items())})
