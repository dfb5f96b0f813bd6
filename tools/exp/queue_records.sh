#!/bin/bash
# The ObserveImage queue's two records (run from the repo root on the GPU box): tools/exp/queue_records.sh <dir under gpurun_out>
out=gpurun_out/$1; mkdir -p $out
(cd tools && make -s time_frontend)
python3 tools/time_frontend.py --dump /tmp/frames.raw 14 > /dev/null 2>&1
./tools/time_frontend /tmp/frames.raw 640 480 14 2000 10000 > $out/observe_image_cpp.json 2> $out/observe_image_cpp.err
python3 tools/time_frontend.py --json > $out/observe_image_queue.json 2> $out/observe_image_queue.err
python3 - $out <<'PY'
import json, sys
d = json.load(open(sys.argv[1] + "/observe_image_cpp.json"))
for k, v in d["results"].items():
    print("%-32s %8.0f frames/s  batches %3d largest %3d" % (k, v["frames_per_s"], v["batches"], v["largest_batch"]))
PY
tail -c 1500 $out/observe_image_queue.json
