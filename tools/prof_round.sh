#!/bin/bash
# All profiles of a round in one gpurun call (run from the repo root on the GPU box):
#   tools/prof_round.sh <tag> [name width height nfeatures batch [bench args ...]]
#     -> gpurun_out/prof_<tag>/{stats,fetch,write,valu}/..., profiles/traffic[_<name>].json rebuilt,
#        bench line of the same build under gpurun_out/prof_<tag>/bench.json
#   e.g.  tools/prof_round.sh r03_vga
#         tools/prof_round.sh r03_1080p 1080p 1920 1080 8000 32 --config 1080p --batch 32
# Counter passes carry --pmc only (no trace domains); the kernel trace is its own run.  The counter passes run the blur in
# line (--blur-inline): per-kernel counters do not depend on what runs beside the kernel, and the run then has exactly
# warm-up + steps steps.
set -e
tag=$1; name=${2:-}; w=${3:-640}; h=${4:-480}; nf=${5:-2000}; batch=${6:-256}
shift; shift || true; shift || true; shift || true; shift || true; shift || true
extra="$@"
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_$tag
mkdir -p "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o bench -- python3 bench.py --no-cpu-baseline --no-observe --no-sustained --no-other-configs $extra > "$out/bench_under_rocprof.json" 2> "$out/stats.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_inline" -o bench -- python3 bench.py --no-cpu-baseline --no-observe --no-sustained --no-other-configs --blur-inline --no-pipeline $extra > "$out/bench_inline_under_rocprof.json" 2> "$out/stats_inline.err"
for pass in fetch:FETCH_SIZE write:WRITE_SIZE "valu:SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES"; do
  pname=${pass%%:*}; counters=${pass#*:}
  rocprofv3 --pmc $counters --output-format csv -d "$out/$pname" -o pmc -- python3 bench.py --no-cpu-baseline --no-observe --no-sustained --no-other-configs --no-inline-pass --blur-inline --no-pipeline --steps 3 --warmup 1 $extra > "$out/$pname.json" 2> "$out/$pname.err"
  echo "pmc pass $pname done"
done
python3 tools/make_traffic.py "$out/fetch" "$out/write" "$out/valu" $batch $w $h $nf $name
cp profiles/traffic${name:+_$name}.json "$out/"
python3 bench.py $extra > "$out/bench.json" 2> "$out/bench.err"
python3 - "$out" <<'PY'
import sys, glob
for f in glob.glob(sys.argv[1] + "/stats/**/*kernel_stats.csv", recursive=True):
    print(open(f).read()[:3000])
print(open(sys.argv[1] + "/bench.json").read())
PY
