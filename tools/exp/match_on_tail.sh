#!/bin/bash
# RECORD OF AN EXPERIMENT (the switch VSF_X_MATCH_ON_EX and the modes qx_* existed only for it: NOTES.md): the queue's stereo
# matcher as the tail stream's first kernel (as in the batched step) against the extraction's stream, where it stayed
python3 tools/time_frontend.py --dump /tmp/frames.raw 32 > /dev/null
for rep in 1 2 3; do for v in tail ex; do
  if [ $v = ex ]; then export VSF_X_MATCH_ON_EX=1; else unset VSF_X_MATCH_ON_EX; fi
  tools/time_frontend /tmp/frames.raw 640 480 32 2000 10000 +qx_ +fused 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$v:', '  '.join('%s %.0f' % (k.replace('qx_',''), v['frames_per_s']) for k,v in d['results'].items()))"
done; done
