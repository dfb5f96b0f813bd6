#!/bin/bash
# A/B of an environment switch on the ObserveImage timings, alternating in one call: tf_ab.sh VAR
out=gpurun_out/r3
mkdir -p $out
(cd tools && make -s time_frontend) || exit 1
python tools/time_frontend.py --dump /tmp/frames.raw 14 > /dev/null 2>&1 || exit 1
for v in 1 0 1 0 1 0; do
  env $1=$v ./tools/time_frontend /tmp/frames.raw 640 480 14 2000 > $out/tf_ab.json 2>/dev/null
  python - <<P
import json
r=json.load(open("$out/tf_ab.json"))["results"]
print("$1=$v", {k: round(v["observe_image_ms_mean"],4) for k,v in r.items()}, round(r["pipelined_2000"]["frames_per_s"]))
P
done
