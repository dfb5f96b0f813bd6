#!/bin/bash
# Shader clock, memory clock, socket power and utilisation (rocm-smi, one sample per ~0.3 s) while bench.py runs its timed
# region AND its sustained leg: is the sustained figure clock- or power-limited?  Run from the repo root on the GPU box:
#   tools/clocks.sh <out dir under gpurun_out> [bench args ...]   ->  <out>/sustained_clocks.txt, <out>/sustained_bench.json
# Each line: unix time, sclk MHz, mclk MHz, socket power W, GPU use %; the bench line's sustained.unix_time brackets the leg.
out=gpurun_out/${1:-clocks}; shift
mkdir -p "$out"
( for i in $(seq 1 200); do
    t=$(date +%s.%N | cut -c1-14)
    rocm-smi --showclocks --showpower --showuse 2>/dev/null | python3 -c "
import re, sys
txt = sys.stdin.read()
g = lambda pat: (re.search(pat, txt) or [None, '?'])[1]
print('$t sclk', g(r'sclk clock level: \\d+: \\((\\d+)Mhz'), 'mclk', g(r'mclk clock level: \\d+: \\((\\d+)Mhz'), 'power_w', g(r'Power \\(W\\): ([0-9.]+)'), 'use', g(r'GPU use \\(%\\): (\\d+)'))"
  done ) > "$out/sustained_clocks.txt" &
SM=$!
sleep 1
python3 bench.py --no-cpu-baseline --no-observe "$@" > "$out/sustained_bench.json" 2> "$out/sustained_bench.err"
kill $SM 2>/dev/null
wait $SM 2>/dev/null
python3 - "$out" <<'PY'
import json, sys
out = sys.argv[1]
b = json.load(open(out + "/sustained_bench.json"))
s = b["sustained"]
print("burst %.0f frames/s (%.3f ms/step), sustained %.0f frames/s over %d steps (%.3f ms/step; first 100: %.3f, last 100: %.3f)"
      % (b["value"], b["ms_per_step"], s["value"], s["steps"], s["ms_per_step"], s["first_100_ms"], s["last_100_ms"]))
t0, t1 = s["unix_time"]
inside = []
for line in open(out + "/sustained_clocks.txt"):
    f = line.split()
    if len(f) >= 9 and t0 <= float(f[0]) <= t1:
        inside.append(line.strip())
print("samples inside the sustained leg (%d):" % len(inside))
print("\n".join(inside))
PY
