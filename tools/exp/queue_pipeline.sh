#!/bin/bash
# RECORD OF AN EXPERIMENT (the option it drives, VSF_OPT_OBSERVE_PIPELINE = 11, existed only for it and is gone again: NOTES.md):
# the queue's batches with their pyramid beside the previous batch (0 / 1 one chain / 2 two chains) and the
# chain's start (VSF_OPT_PIPE_AFTER_FAST = option 6: 1 behind the previous FAST / 0 as soon as the upload is there)
python3 tools/time_frontend.py --dump /tmp/frames.raw 32 > /dev/null
for rep in 1 2; do for v in "11=0" "11=1" "11=2" "11=1 @6=0" "11=2 @6=0"; do
  args=""; for a in $v; do args="$args @${a#@}"; done
  tools/time_frontend /tmp/frames.raw 640 480 32 2000 10000 +queued_d128 +queued_no_copy $args 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$v:', '  '.join('%s %.0f (%d batches, wait %.1f)' % (k.replace('queued_',''), v['frames_per_s'], v['batches'], v['wait_us_per_frame']) for k,v in d['results'].items()))"
done; done
