# positions a lane tries per span (-DVSF_PNG_SPAN_REGS=n): inflate kernel time per 512 files, tools/time_png.py
set -e
cd vision_slam_frontend_amd/csrc
for v in 2 3 6 8; do
  /opt/rocm/bin/hipcc -O3 -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -DVSF_PNG_SPAN_REGS=$v -c k_png.hip -o k_png.o
  make ../libvsf_hip.so > /dev/null 2>&1
  cd ../..
  echo "== span registers $v"
  bash tools/exp/png_kernels.sh 512 | grep -E "level (6|cv|1) |png_inflate" | cut -c1-150
  cd vision_slam_frontend_amd/csrc
done
