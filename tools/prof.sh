#!/bin/bash
# Profiling helper for the GPU box (run through gpurun from the repo root):
#   tools/prof.sh stats <tag> [bench args]   -> kernel-trace + stats CSVs under gpurun_out/prof_<tag>/
#   tools/prof.sh pmc <tag> "<counters>" [bench args] -> one --pmc pass (counters only; no trace domains)
set -e
mode=$1; tag=$2; shift 2
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_$tag
mkdir -p "$out"
if [ "$mode" = stats ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o bench -- python3 bench.py --no-cpu-baseline "$@" > "$out/bench.json" 2> "$out/bench.err"
  python3 - "$out" <<'PY'
import sys, glob
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    print(open(f).read()[:3000])
PY
else
  counters=$1; shift
  rocprofv3 --pmc $counters --output-format csv -d "$out" -o pmc -- python3 bench.py --no-cpu-baseline "$@" > "$out/bench.json" 2> "$out/bench.err"
  python3 - "$out" <<'PY'
import sys, glob, csv, collections
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"][:60], r["Counter_Name"])
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    for (k, c), (n, v) in sorted(agg.items()):
        print("%-62s %-28s n=%-5d mean=%.4g" % (k, c, n, v / n))
PY
fi
