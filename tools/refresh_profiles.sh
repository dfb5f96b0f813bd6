#!/bin/bash
# After tools/prof_round.sh r05_vga / r05_nf10000 / r05_1080p have run on the GPU box (gpurun_out/prof_<tag>/ merged back):
# puts the three traffic files, the per-config records and the issue ceilings in place and checks that all of them carry
# the digest of the sources they were taken from.    tools/refresh_profiles.sh [round dir]   (default profiles/r05)
set -e
cd "$(dirname "$0")/.."
dst=${1:-profiles/r05}
cp gpurun_out/prof_r05_vga/traffic.json profiles/traffic.json
cp gpurun_out/prof_r05_nf10000/traffic_nf10000.json profiles/traffic_nf10000.json
cp gpurun_out/prof_r05_1080p/traffic_1080p.json profiles/traffic_1080p.json
python3 tools/collect_profiles.py r05_vga vga $dst
python3 tools/collect_profiles.py r05_nf10000 nf10000 $dst
python3 tools/collect_profiles.py r05_1080p 1080p $dst
python3 tools/valu_ceiling.py > /dev/null
python3 - <<'PY'
import json
from vision_slam_frontend_amd.buildinfo import kernel_source_hash as h
ok = all(json.load(open(p))["source_hash"] == h() for p in ("profiles/traffic.json", "profiles/traffic_nf10000.json", "profiles/traffic_1080p.json", "profiles/valu_ceiling.json"))
print("source digests fresh:", ok)
for f in ("vga", "nf10000", "1080p"):
    d = json.load(open("profiles/r05/%s_bench.json" % f)); r = d["roofline"]
    print("%-8s %6.0f frames/s %7.3f ms  roofline %.4f (in line %.4f)  sustained %.0f" % (f, d["value"], d["ms_per_step"], r["frac"], r["in_line"]["frac"], (d.get("sustained") or {}).get("value", 0)))
raise SystemExit(0 if ok else 1)
PY
