import csv, glob, sys, collections
# sums counters per kernel name (per-launch mean) from a rocprofv3 --pmc csv directory
d = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else ''
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if pat not in k: continue
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[k].add(r['Dispatch_Id'])
for k, c in acc.items():
    print(k[:70], 'launches', len(n[k]), {a: round(b / len(n[k])) for a, b in c.items()})
