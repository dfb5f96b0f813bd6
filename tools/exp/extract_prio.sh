# the extraction's kernels at issue priority n (-DVSF_EXTRACT_PRIO=n) over whatever shares the chip with them (the JPEG
# decode of the next step): bench.py --ingest jpeg, and the HBM-resident bench for comparison
set -e
cd vision_slam_frontend_amd/csrc
for v in none 1 3; do
  if [ $v = none ]; then D=""; else D="-DVSF_EXTRACT_PRIO=$v"; fi
  for f in k_pyramid k_fast k_blur k_select k_describe; do
    /opt/rocm/bin/hipcc -O3 -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt $D -c $f.hip -o $f.o &
  done
  /opt/rocm/bin/hipcc -O3 -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -mllvm -amdgpu-mfma-vgpr-form $D -c k_match.hip -o k_match.o &
  wait
  make ../libvsf_hip.so > /dev/null 2>&1
  cd ../..
  for args in "--ingest jpeg" ""; do
    for rep in 1 2; do
      python3 bench.py --no-cpu-baseline --no-observe --no-sustained $args > /tmp/b.json 2>/tmp/b.err
      python3 -c "
import json,sys; d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]); print('extract prio %-5s %-14s %7.0f frames/s %7.3f ms/step' % (sys.argv[1], sys.argv[2] or 'hbm', d['value'], d['ms_per_step']))" $v "$args"
    done
  done
  cd vision_slam_frontend_amd/csrc
done
