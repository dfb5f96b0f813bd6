#!/bin/bash
out=gpurun_out/r3
mkdir -p $out
(cd tools && make -s time_frontend) || exit 1
python tools/time_frontend.py --dump /tmp/frames.raw 14 > /dev/null 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 150 rocprofv3 --hip-trace --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/tf_hip -o tf -- $GRAFT_REPO_ROOT/tools/time_frontend /tmp/frames.raw 640 480 14 2000 > /dev/null 2>&1
ls $GRAFT_REPO_ROOT/$out/tf_hip
