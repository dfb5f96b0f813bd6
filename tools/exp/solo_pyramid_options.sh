#!/bin/bash
# synchronous ObserveImage against the slab pyramid's launch choices: VSF_OPT_PYRAMID_CHAIN (3), VSF_OPT_PYRAMID_ROWS (4)
python3 tools/time_frontend.py --dump /tmp/frames.raw 32 > /dev/null
for chain in 8 10 12 16 24; do for rows in 4 6 8 12; do
  r=$(tools/time_frontend /tmp/frames.raw 640 480 32 2000 +fused @3=$chain @4=$rows 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['results']['fused_2000']['observe_image_ms_mean'])")
  echo "chain $chain rows $rows: $r ms"
done; done
