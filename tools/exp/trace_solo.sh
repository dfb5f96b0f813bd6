#!/bin/bash
# kernel trace of the synchronous ObserveImage (a batch of one frame): where its 0.36 ms go on the GPU
set -e
export TMPDIR=/tmp
out=$PWD/gpurun_out/trace_solo
mkdir -p $out
python3 tools/time_frontend.py --dump /tmp/frames.raw 32 > /dev/null
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out -o t -- tools/time_frontend /tmp/frames.raw 640 480 32 2000 +fused > $out/run.json 2> $out/run.err
python3 - $out <<'PY'
import sys, glob, csv
out = sys.argv[1]
ev = []
for f in glob.glob(out + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
for f in glob.glob(out + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")))
ev.sort()
# frames are separated by the H2D copy of 614 KB: take the 100th frame
starts = [i for i, e in enumerate(ev) if e[2].startswith("COPY") and "HOST_TO_DEVICE" in e[2].upper()]
print("events", len(ev), "h2d copies", len(starts))
i0 = starts[100]; i1 = starts[101]
t0 = ev[i0][0]
prev_end = t0
for s, e, n in ev[i0:i1]:
    print("%8.1f  +%6.1f us  gap %5.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, n))
    prev_end = e
print("frame span on the GPU: %.1f us" % ((ev[i1 - 1][1] - t0) / 1e3))
PY
