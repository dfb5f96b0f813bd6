#!/bin/bash
# ObserveImage through the C++ driver: timings, then a kernel trace of the synchronous mode
out=gpurun_out/r3
mkdir -p $out
(cd tools && make -s time_frontend) || exit 1
python tools/time_frontend.py --dump /tmp/frames.raw 14 > /dev/null 2>&1 || exit 1
./tools/time_frontend /tmp/frames.raw 640 480 14 2000 > $out/tf_$1.txt 2>&1
cat $out/tf_$1.txt | tail -8
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$out/tf_trace_$1 -o tf -- $GRAFT_REPO_ROOT/tools/time_frontend /tmp/frames.raw 640 480 14 2000 > /dev/null 2>&1
ls $GRAFT_REPO_ROOT/$out/tf_trace_$1
