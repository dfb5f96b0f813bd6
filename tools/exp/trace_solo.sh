#!/bin/bash
# kernel traces of ObserveImage: (1) the synchronous call (a batch of one frame): where its 0.36 ms go on the GPU;
# (2) the default queue (depth 256, <= 128 frames per batch): the GPU's timeline over two consecutive batches
set -e
export TMPDIR=/tmp
out=$PWD/gpurun_out/trace_solo
rm -rf $out; mkdir -p $out
python3 tools/time_frontend.py --dump /tmp/frames.raw 32 > /dev/null
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out/solo -o t -- tools/time_frontend /tmp/frames.raw 640 480 32 2000 +fused > $out/solo.json 2> $out/solo.err
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out/queue -o t -- tools/time_frontend /tmp/frames.raw 640 480 32 2000 +queued_d256 > $out/queue.json 2> $out/queue.err || true
python3 - $out <<'PY'
import sys, glob, csv, collections
out = sys.argv[1]
def load(d):
    ev = []
    for f in glob.glob(out + "/" + d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:44], r.get("Queue_Id", "?")))
    for f in glob.glob(out + "/" + d + "/**/*memory_copy_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", ""), "copy"))
    ev.sort()
    return ev
ev = load("solo")
starts = [i for i, e in enumerate(ev) if e[2].startswith("COPY") and "HOST_TO_DEVICE" in e[2].upper()]
print("== synchronous ObserveImage, frame 100 of the run (us from the upload's start; duration; gap to the previous end)")
i0, i1 = starts[100], starts[101]
t0 = ev[i0][0]; prev = t0
for s, e, n, q in ev[i0:i1]:
    print("%8.1f  +%6.1f  gap %5.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, n)); prev = e
print("frame span on the GPU: %.1f us" % ((ev[i1 - 1][1] - t0) / 1e3))
ev = load("queue")
if ev:
    big = [i for i, e in enumerate(ev) if e[2].startswith("COPY") and "HOST_TO_DEVICE" in e[2].upper() and e[1] - e[0] > 200000]
    print("== the default queue: uploads of > 0.2 ms:", len(big))
    if len(big) > 12:
        a, b = big[8], big[10]
        t0 = ev[a][0]
        per = collections.OrderedDict()
        busy = []
        for s, e, n, q in ev[a:b]:
            k = (n, q)
            d = per.setdefault(k, [0, 0.0, 1e18, 0])
            d[0] += 1; d[1] += (e - s) / 1e3; d[2] = min(d[2], (s - t0) / 1e3); d[3] = max(d[3], (e - t0) / 1e3)
            busy.append((s, e))
        print("two consecutive batches: %.0f us from the first upload's start to the third's; per kernel and queue: launches, summed us, first start, last end" % ((ev[b][0] - t0) / 1e3))
        for (n, q), d in per.items():
            print("  %-46s q%-4s x%-4d %8.1f us   %8.1f .. %8.1f" % (n, q, d[0], d[1], d[2], d[3]))
        # union of busy intervals of kernels (any queue)
        busy.sort(); tot = 0; cs, ce = busy[0]
        for s, e in busy[1:]:
            if s > ce: tot += ce - cs; cs, ce = s, e
            else: ce = max(ce, e)
        tot += ce - cs
        print("  some kernel or copy was running during %.0f us of them" % (tot / 1e3))
PY
