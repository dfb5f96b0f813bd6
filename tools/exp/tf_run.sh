#!/bin/bash
# ObserveImage through the C++ driver at 2000 and 10000 features -> gpurun_out/r3/observe_image.json
out=gpurun_out/r3
mkdir -p $out
(cd tools && make -s time_frontend) || exit 1
python tools/time_frontend.py --dump /tmp/frames.raw 14 > /dev/null 2>&1 || exit 1
./tools/time_frontend /tmp/frames.raw 640 480 14 2000 10000 > $out/observe_image.json 2> $out/observe_image.err
cat $out/observe_image.json
