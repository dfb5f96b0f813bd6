#!/bin/bash
# The round's bench records that need no counters, in one gpurun call (run from the repo root on the GPU box):
#   tools/records_round.sh <dir under gpurun_out>
# -> jpeg_ingest_bench.json, png_ingest_bench.json, png_time.txt, window8_bench.json, sparse_scene_bench.json, 1080p_batch32_bench.json,
#    rehearsal_2ranks_bare_form_bench.json, capi_collectives_world_of_one_bench.json, cpp_sharded_driver_1gpu.json,
#    time_match.txt, observe_depth.txt, sustained_clocks.txt / sustained_bench.json
out=gpurun_out/$1; mkdir -p $out
B="python3 bench.py --no-cpu-baseline --no-observe"
$B --ingest jpeg > $out/jpeg_ingest_bench.json 2> $out/jpeg_ingest_bench.err
$B --ingest png --no-sustained > $out/png_ingest_bench.json 2> $out/png_ingest_bench.err
$B --window 8 > $out/window8_bench.json 2> $out/window8_bench.err
$B --scene sparse > $out/sparse_scene_bench.json 2> $out/sparse_scene_bench.err
$B --config 1080p > $out/1080p_batch32_bench.json 2> $out/1080p_batch32_bench.err
$B --collectives capi --no-sustained > $out/capi_collectives_world_of_one_bench.json 2> $out/capi.err
VSF_BENCH_ONE_GPU=1 python3 bench.py --gpus 2 --batch 16 --steps 10 --sustained-steps 600 > $out/rehearsal_2ranks_bare_form_bench.json 2> $out/rehearsal.err
echo "rehearsal rc=$?" >> $out/rehearsal.err
python3 tools/time_match.py > $out/time_match.txt 2>&1
python3 tools/time_png.py 512 2>&1 | grep -E "^(synthetic|photographs)" > $out/png_time.txt
python3 tools/time_observe_depth.py 10000 332 3 4 > $out/observe_depth.txt 2>&1
python3 tools/time_observe_depth.py 2000 332 3 4 >> $out/observe_depth.txt 2>&1
(cd tools && make -s time_sharded time_frontend)
python3 - <<'PY'
import sys
import numpy as np
sys.path.insert(0, ".")
from vision_slam_frontend_amd import synth
f = synth.bench_batch(256, 640, 480, seed=synth.BASE_SEED)  # the bench's three rotating batches, as one file of 768 frames
np.concatenate([f, np.roll(f, (3, 11), (2, 3)), np.roll(f, (8, 29), (2, 3))]).tofile("/tmp/frames_sharded.raw")
PY
./tools/time_sharded /tmp/frames_sharded.raw 640 480 768 2000 256 1 20 > $out/cpp_sharded_driver_1gpu.json 2> $out/cpp_sharded.err
python3 tools/time_frontend.py --dump /tmp/frames.raw 14 > /dev/null 2>&1 && ./tools/time_frontend /tmp/frames.raw 640 480 14 2000 10000 > $out/observe_image_cpp.json 2> $out/observe_image_cpp.err
tools/clocks.sh $1 > $out/clocks.log 2>&1
for f in jpeg_ingest png_ingest window8 sparse_scene 1080p_batch32 capi_collectives_world_of_one rehearsal_2ranks_bare_form; do python3 -c "
import json; d=json.load(open('$out/${f}_bench.json')); print('%-36s %8.0f frames/s %7.3f ms/step' % ('$f', d['value'], d['ms_per_step']), (d.get('sustained') or {}).get('value'))"; done
cat $out/cpp_sharded_driver_1gpu.json | cut -c1-300; tail -4 $out/observe_depth.txt; head -3 $out/clocks.log
