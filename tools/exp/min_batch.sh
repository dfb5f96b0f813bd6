#!/bin/bash
# RECORD OF AN EXPERIMENT (the modes qx_* of tools/time_frontend.cc existed only for it: NOTES.md): the queue's min_batch
# at several depths, same box, alternating runs
python3 tools/time_frontend.py --dump /tmp/frames.raw 32 > /dev/null
for rep in 1 2 3; do
  tools/time_frontend /tmp/frames.raw 640 480 32 2000 10000 +qx_ 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
for k,v in d['results'].items():
    if 'qx_' in k or k.startswith('queued_2') or k.startswith('queued_1000'):
        print('%-26s %7.0f fps  batches %3d largest %3d  copy %.1f launch %.1f wait %.1f' % (k, v['frames_per_s'], v['batches'], v['largest_batch'], v['copy_us_per_frame'], v['launch_us_per_frame'], v['wait_us_per_frame']))
print('--')"
done
