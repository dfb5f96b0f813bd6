#!/bin/bash
# All profiles of a round in one gpurun call (run from the repo root on the GPU box):
#   tools/prof_round.sh <tag>     -> gpurun_out/prof_<tag>/{stats,fetch,write,valu}/..., traffic.json rebuilt,
#                                    bench line of the same build under gpurun_out/prof_<tag>/bench.json
# Counter passes carry --pmc only (no trace domains); the kernel trace is its own run.
set -e
tag=$1
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_$tag
mkdir -p "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o bench -- python3 bench.py --no-cpu-baseline > "$out/bench_under_rocprof.json" 2> "$out/stats.err"
for pass in fetch:FETCH_SIZE write:WRITE_SIZE "valu:SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES"; do
  name=${pass%%:*}; counters=${pass#*:}
  rocprofv3 --pmc $counters --output-format csv -d "$out/$name" -o pmc -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 > "$out/$name.json" 2> "$out/$name.err"
  echo "pmc pass $name done"
done
python3 tools/make_traffic.py "$out/fetch" "$out/write" "$out/valu" 256
mkdir -p gpurun_out/prof_$tag && cp profiles/traffic.json gpurun_out/prof_$tag/traffic.json

python3 bench.py > "$out/bench.json" 2> "$out/bench.err"
python3 - "$out" <<'PY'
import sys, glob
for f in glob.glob(sys.argv[1] + "/stats/**/*kernel_stats.csv", recursive=True):
    print(open(f).read()[:2500])
print(open(sys.argv[1] + "/bench.json").read())
PY
