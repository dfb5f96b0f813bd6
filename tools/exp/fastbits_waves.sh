set -e
cd vision_slam_frontend_amd/csrc
for w in 1 2 4; do
  /opt/rocm/bin/hipcc -O3 -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -DVSF_FB_WAVES=$w -c k_fastbits.hip -o k_fastbits.o
  make ../libvsf_hip.so > /dev/null 2>&1
  cd ../..
  echo "== waves per workgroup: $w"
  timeout -k 10 200 python3 tools/time_fastbits.py 640 480 2000 512 bench 2>&1 | grep -E "form 2|differing|equal"
  cd vision_slam_frontend_amd/csrc
done
