#!/bin/bash
out=gpurun_out/r5; mkdir -p $out
J="python3 bench.py --ingest jpeg --no-cpu-baseline --no-observe --no-sustained --steps 40 --warmup 5"
run() { name=$1; shift; "$@" > $out/jm_$name.json 2> $out/jm_$name.err; python3 -c "
import json,sys; d=json.load(open('$out/jm_$name.json')); print('%-34s %8.0f fps %6.3f ms  %s %s' % ('$name', d['value'], d['ms_per_step'], d['config']['fast_resident'][:60], d['config']['fast_tune']))" ; }
run jpeg_tuned $J
run jpeg_high_tuned $J --ingest-priority high
run jpeg_grid_high $J --fast-resident 0 --ingest-priority high
run jpeg_grid_normal $J --fast-resident 0
run jpeg_grid_blur_inline $J --fast-resident 0 --blur-inline
