#!/usr/bin/env python3
"""Headline benchmark: stereo frames/s through the per-frame hot path of Frontend::ObserveImage
(slam_frontend.cc:400-443) on synthetic 640x480 stereo pairs, 2000 keypoints per frame (BASELINE.json configs[1]),
inputs resident in HBM, on N GPUs of one node: one process per GPU, frames sharded in blocks, the RemoveAmbigStereo
means all-gathered, compact VisionFeature / FeatureMatch payloads gathered to rank 0 over RCCL (configs[3]).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
    python bench.py --config 1080p            # BASELINE configs[2]: 1920x1080, 8000 keypoints

A "step" is one pass of the hot path over one batch of `--batch` stereo frames per GPU: extract(L) + extract(R) +
GetMatches(L, R) -- the part BASELINE's metric names and >= 97 % of the step -- followed by the reference's own steps up
to the output records (RemoveAmbigStereo, GetFeatureMatches against the previous frame, Calculate3DPoints,
UndistortFeaturePoints) and the packing of the payload; the SAME per-GPU work at every N, plus the collectives for
N > 1 (vision_slam_frontend_amd/distributed.py).  Rank 0 prints ONE JSON line (task contract + `roofline`,
`roofline_valu`, `matcher` and `cpu_baseline` objects).
"""
from __future__ import annotations

import argparse
import json
import subprocess
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
VALU_PEAK_GINST = 545.0      # measured: a stream of 4-cycle wave64 VALU instructions chip-wide (profiles/r04/valu_issue_table.json;
                             # 1024 SIMDs x ~2.2 GHz under that load / 4.1 cycles); per kernel: valu_ceilings()
MFMA_I8_PEAK_TOPS = 5000.0   # dense int8 = the fp8 rate (MI355X_MICROARCH.md: ~5 PFLOP/s fp8 dense)
MFMA_FP4_PEAK_TOPS = 10000.0  # dense FP4 / FP6 (same guide: ~10 PF dense): the form the matcher runs in since round 5

CONFIGS = {
    "vga": dict(width=640, height=480, nfeatures=2000, batch=256, label="BASELINE configs[1]"),
    "1080p": dict(width=1920, height=1080, nfeatures=8000, batch=32, label="BASELINE configs[2]"),
}


def stage_algorithmic_bytes(ctx, n_images: int, n_pairs: int, nfeatures: int) -> dict:
    """Algorithmic (compulsory) HBM bytes of each stage for ONE step; SURVEY.md section 8(d)."""
    L = ctx.nlevels
    px = [ctx.level_info(l)[0] * ctx.level_info(l)[1] for l in range(L)]
    P = sum(px)
    rec = 28 + 32
    return {
        "pyramid_resize": n_images * (sum(px[:-1]) + sum(px[1:])),  # every level read once, written once
        "fast_score_nms": n_images * P,                            # each pyramid pixel read once
        "select_harris_angle": n_images * nfeatures * 12,         # final level-keypoint records (gathers excluded)
        "gauss_blur7": n_images * 2 * P,                           # read + write
        "orb_describe": n_images * nfeatures * rec,                # cv::KeyPoint + descriptor out
        "hamming_knn2": n_pairs * (32 * 2 * nfeatures + 16 * nfeatures),
        "ratio_compact": n_pairs * (16 * nfeatures + 16 * nfeatures),
        "frontend_tail": n_pairs * nfeatures * (16 + 60 + 28),     # matches in, filtered frames + VisionFeature out
    }


def committed_counters(width: int, height: int, nfeatures: int, batch: int):
    """(per-stage counters, stale) of the committed rocprofv3 --pmc passes (profiles/traffic*.json: FETCH_SIZE, WRITE_SIZE
    and SQ_INSTS_VALU collected in their own runs) taken on this configuration; ({}, False) if there are none.  PMC counters
    cannot be collected inside the timed run, so the file carries a digest of the kernel sources it was taken from
    (tools/make_traffic.py): `stale` says the sources have changed since -- the numbers are then NOT published."""
    from vision_slam_frontend_amd.buildinfo import kernel_source_hash
    for path in sorted((ROOT / "profiles").glob("traffic*.json")):  # one file per profiled configuration
        try:
            t = json.loads(path.read_text())
        except (OSError, ValueError):
            continue
        c = t.get("config", {})
        if (c.get("width"), c.get("height"), c.get("nfeatures"), c.get("batch")) == (width, height, nfeatures, batch):
            return t.get("stages", {}), t.get("source_hash") != kernel_source_hash()
    return {}, False


def valu_ceilings():
    """Per-stage VALU issue ceilings in G wave-inst/s (profiles/valu_ceiling.json: the kernel's own opcode mix priced with
    the measured issue table; tools/valu_ceiling.py), and whether the file is stale against the sources.  The fallback is the
    measured rate of a stream of 4-cycle instructions."""
    from vision_slam_frontend_amd.buildinfo import kernel_source_hash
    try:
        t = json.loads((ROOT / "profiles" / "valu_ceiling.json").read_text())
    except (OSError, ValueError):
        return {}, VALU_PEAK_GINST, True
    return ({k: v["ceiling_mix_g_wave_inst_per_s"] for k, v in t["kernels"].items()}, t["rate_all4_g_wave_inst_per_s"],
            t.get("source_hash") != kernel_source_hash())


def host_cores():
    """(threads to use, how they were counted): the cores this process may actually run on -- its affinity mask, cut
    down to the cgroup's CPU quota when there is one (a GPU box shows every core of the host but grants a share)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    how = "all %d cores of the affinity mask" % n
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = Path(path).read_text().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
            if quota not in ("max", "-1") and float(quota) > 0:
                q = max(1, int(float(quota) / period + 0.5))
                if q < n:
                    n, how = q, "the cgroup CPU quota of %d cores (host shows %d)" % (q, len(os.sched_getaffinity(0)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n), how


def cpu_baseline(width, height, nfeatures, seed, seconds=15.0, kind="port"):
    """The CPU oracle (a scalar C++ restatement of the OpenCV routines, kind "port": real OpenCV cannot be built here)
    timed on ALL of this box's host cores, one stereo frame per thread at a time.  `digest`: sha256 over the keypoints,
    descriptors and matches of the first frames -- what lets two builds of the oracle be held against each other."""
    import hashlib
    from concurrent.futures import ThreadPoolExecutor

    from oracle import binding as ob
    from vision_slam_frontend_amd import synth

    if not os.environ.get("VSF_ORACLE_LIB"):
        ob.build()
    cores, how = host_cores()
    n_render = min(4 * cores, 64)
    frames = synth.bench_batch(n_render, width, height, seed=seed, n_scenes=min(4, n_render))

    def one(i, digest=None):
        a, b = ob.Orb(nfeatures=nfeatures), ob.Orb(nfeatures=nfeatures)
        a.run(frames[i % n_render, 0])
        b.run(frames[i % n_render, 1])
        ka, da = a.result()
        kb, db = b.result()
        m = ob.get_matches(da, db)
        if digest is not None:
            for arr in (ka, da, kb, db, m):
                digest.update(arr.tobytes())
        return len(m)

    h = hashlib.sha256()
    for i in range(min(4, n_render)):  # warm (page in, build tables) + the digest of four frames
        one(i, h)
    # A box may show more cores than it grants (a CPU share without a readable cgroup quota): one calibration round on
    # the visible cores measures how many actually ran (process CPU time / wall time); the timed rounds use that many.
    if cores > 1:
        c0, w0 = sum(os.times()[:2]), time.perf_counter()
        with ThreadPoolExecutor(min(cores, 64)) as ex:
            list(ex.map(one, range(min(cores, 64))))
        granted = (sum(os.times()[:2]) - c0) / max(time.perf_counter() - w0, 1e-9)
        if granted < 0.75 * min(cores, 64):
            cores = max(1, int(granted + 0.5))
            how = "the %d cores this process was granted (measured: CPU time / wall time; the host shows more)" % cores
    # rounds of one frame per thread until `seconds` of wall time have passed (at least two rounds): bounded whatever the box
    n, t0 = 0, time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        while True:
            list(ex.map(one, range(n, n + cores)))
            n += cores
            dt = time.perf_counter() - t0
            if dt >= seconds and n >= 2 * cores:
                break
    return {"value": n / dt, "unit": "stereo frames/s", "cores": cores, "kind": kind, "digest": h.hexdigest(),
            "sample": "%d synthetic %dx%d stereo frames (%d kp) through the oracle port's extract(L)+extract(R)+GetMatches "
                      "(scalar C++ restatement of OpenCV 3.2, not OpenCV itself), %d threads = %s, %.1f s wall"
                      % (n, width, height, nfeatures, cores, how, dt)}


def cpu_baseline_native(width, height, nfeatures, seed, parity_digest):
    """The same sources built with -O3 -march=native ON THIS HOST (oracle/Makefile `native`: hardware popcount,
    auto-vectorised loops) and timed the same way in a child process (the oracle's binding holds one library per process);
    refused unless its outputs equal the parity build's byte for byte."""
    try:
        r = subprocess.run(["make", "-s", "-C", str(ROOT / "oracle"), "native"], capture_output=True, text=True, timeout=300)
        if r.returncode != 0:
            return {"error": "build failed: " + (r.stderr or r.stdout)[-300:]}
        env = dict(os.environ, VSF_ORACLE_LIB=str(ROOT / "oracle" / "_native" / "libvsf_oracle_native.so"))
        r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--cpu-baseline-child", "--width", str(width), "--height",
                            str(height), "--nfeatures", str(nfeatures)], capture_output=True, text=True, timeout=300, env=env)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not line:
            return {"error": "child rc %d: %s" % (r.returncode, (r.stderr or "")[-300:])}
        j = json.loads(line[-1])
    except (subprocess.TimeoutExpired, OSError, ValueError) as e:
        return {"error": repr(e)}
    if j.get("digest") != parity_digest:
        return {"error": "outputs differ from the parity build's: not timed as a baseline", "digest": j.get("digest")}
    j["flags"] = "-O3 -DNDEBUG -march=native -ffp-contract=off (oracle/Makefile native), outputs byte-equal to the parity build"
    j["kind"] = "port"
    return j


class JpegIngest:
    """DecodeImage (slam_frontend_main.cc:98-109) on the device for `bench.py --ingest jpeg`: the frames of a step arrive
    as baseline-JPEG files in HOST memory (the CompressedImage payloads; encoded here once with Pillow, quality 80, and
    treated as bayer_rggb8 mosaics like the reference's camera topics) and are uploaded, decoded
    (vsf_jpeg_decode_gray_batch) and demosaiced (vsf_bayer_bg_to_gray_batch_dev) on a context and stream of their own,
    one step AHEAD of the extraction: two image buffers alternate, two events per buffer order the streams."""

    def __init__(self, frames, width, height, nfeatures, device_index, dev, consumer_stream, priority="normal", fmt="jpeg", depth=1):
        import ctypes as C
        import io

        import numpy as np
        import torch
        from PIL import Image

        from vision_slam_frontend_amd import capi
        self.capi, self.C, self.torch = capi, C, torch
        self.W, self.H, self.consumer = width, height, consumer_stream
        self.files = []
        self.fmt = fmt  # "jpeg": baseline JPEG, quality 80; "png": grayscale PNG, libpng's level 6 (vsf_png_decode_gray_batch)
        for f in frames.reshape(-1, height, width):
            bio = io.BytesIO()
            if fmt == "png":
                Image.fromarray(f, "L").save(bio, "PNG", compress_level=6)
            else:
                Image.fromarray(f, "L").save(bio, "JPEG", quality=80)
            self.files.append(np.frombuffer(bio.getvalue(), np.uint8))
        self.n_files = len(self.files)
        self.avg_kb = sum(len(f) for f in self.files) / self.n_files / 1024
        self.ptrs = (C.c_void_p * self.n_files)(*[f.ctypes.data for f in self.files])
        self.sizes = (C.c_size_t * self.n_files)(*[len(f) for f in self.files])
        # The decode should only fill what the extraction leaves free: the LOWEST stream priority.  torch offers two
        # levels (and the step's tail already has the high one); HIP has a third, so the stream is made with HIP itself.
        self.stream = None
        if priority in ("low", "high"):
            try:
                hip = C.CDLL("libamdhip64.so")
                least, greatest, h = C.c_int(0), C.c_int(0), C.c_void_p()
                if hip.hipDeviceGetStreamPriorityRange(C.byref(least), C.byref(greatest)) == 0 and \
                        hip.hipStreamCreateWithPriority(C.byref(h), C.c_uint(1), least if priority == "low" else greatest) == 0 \
                        and h.value:
                    self.stream = torch.cuda.ExternalStream(h.value, device=dev)
                    self.stream_priority = (least if priority == "low" else greatest).value
            except OSError:
                pass
        if self.stream is None:
            self.stream = torch.cuda.Stream(device=dev)
            self.stream_priority = 0
        # `depth` steps' files per decode call (a PNG decode is one wave per file and far longer than a step: the files of
        # two steps in one launch put four waves on every CU instead of two); two group buffers take turns
        self.depth = depth = max(1, int(depth))
        self.ctx = capi.Context(capi.default_params(width, height, max_images=2, nfeatures=nfeatures), device=device_index)
        self.ctx.set_stream(self.stream.cuda_stream)
        if depth > 1:
            self.files = self.files * depth
            self.ptrs = (C.c_void_p * len(self.files))(*[f.ctypes.data for f in self.files])
            self.sizes = (C.c_size_t * len(self.files))(*[len(f) for f in self.files])
        self.n_call = len(self.files)  # files per decode call
        self.d_mosaic = torch.empty((self.n_call, height, width), dtype=torch.uint8, device=dev)
        self.d_in = [torch.empty((depth, self.n_files // 2, 2, height, width), dtype=torch.uint8, device=dev) for _ in range(2)]
        self.ready = [torch.cuda.Event() for _ in range(2)]
        self.consumed = [torch.cuda.Event() for _ in range(2)]
        self.issued = 0   # decode calls started
        self.taken = 0    # steps handed out
        self.current = 0
        self._issue()

    def _issue(self):
        C, W, H = self.C, self.W, self.H
        slot = self.issued & 1
        if self.issued >= 2:
            self.stream.wait_event(self.consumed[slot])  # (the last step that read this buffer has been queued)
        decode = self.capi.lib().vsf_png_decode_gray_batch if self.fmt == "png" else self.capi.lib().vsf_jpeg_decode_gray_batch
        st = decode(self.ctx._h, C.cast(self.ptrs, C.c_void_p), C.cast(self.sizes, C.c_void_p), self.n_call, W, H,
                    C.c_void_p(self.d_mosaic.data_ptr()), W * H, W)
        if st != self.capi.VSF_OK:
            raise self.capi.VsfError(st, "vsf_%s_decode_gray_batch" % self.fmt)
        self.ctx.bayer_bg_to_gray_batch_dev(self.d_mosaic.data_ptr(), self.n_call, W, H, W * H, W,
                                            self.d_in[slot].data_ptr(), W * H, W)
        self.ready[slot].record(self.stream)
        self.issued += 1

    def next_batch(self):
        """(this step's frames, the event behind their producer): the extraction is handed the event (vsf_set_input_event)
        -- its pipelined pyramid, which is ordered after nothing else, waits for it on the GPU; with the first step of a
        group the decode of the next group is started."""
        group, within = divmod(self.taken, self.depth)
        self.taken += 1
        slot = group & 1
        if within == 0:
            self.consumer.wait_event(self.ready[slot])
            self._issue()
        self.current = slot
        self.last_of_group = within == self.depth - 1
        return self.d_in[slot][within], self.ready[slot]

    def release(self):
        if self.last_of_group:  # the buffer may be overwritten once the group's last step has read it
            self.consumed[self.current].record(self.consumer)


# ---------------------------------------------------------------------------------------------------------------------
# `python3 bench.py --gpus N` WITHOUT a launcher: the parent starts the N ranks itself
# ---------------------------------------------------------------------------------------------------------------------

STAGE_ENV = "VSF_BENCH_STAGE_FILE"  # a rank writes the stage it has reached into this file (the parent's watchdog reads it)


def report_stage(name: str) -> None:
    """One line `<unix time> <stage>` into the rank's stage file, when a launching parent asked for one."""
    path = os.environ.get(STAGE_ENV)
    if path:
        try:
            with open(path, "w") as f:
                f.write("%.3f %s\n" % (time.time(), name))
        except OSError:
            pass


def _free_port() -> int:
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n: int, argv, *, child_cmd=None, deadline_s: float = 900.0, stall_s: float = 420.0,
                 grace_s: float = 20.0, extra_env=None, out=sys.stdout, err=sys.stderr, result=None) -> int:
    """Starts `n` ranks of this script as fresh child processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 /
    MASTER_PORT in their environment: what torch.distributed.run would set), relays rank 0's standard output (the ONE JSON
    line) and every rank's standard error, and returns the worst exit code.  This process never imports torch and never
    touches a GPU, and nothing is exec'ed over a process that has.

    Watchdog: a rank that exits non-zero takes the others with it after `grace_s` (they would wait for it in a rendezvous
    forever); no stage change on any rank for `stall_s`, or no end after `deadline_s`, kills every child (each in its own
    session: the exact process groups started here) and returns 124 after printing the stage each rank last reported --
    a hang inside ncclCommInitRank must not eat the caller's timeout silently."""
    import shutil
    import signal
    import subprocess
    import tempfile
    import threading

    cmd = list(child_cmd) if child_cmd else [sys.executable, str(Path(__file__).resolve())]
    port = _free_port()
    stage_dir = tempfile.mkdtemp(prefix="vsf_bench_")
    procs, pumps = [], []
    json_lines = []

    def pump(stream, sink, keep=None):
        # rank 0's standard output carries the result; anything else a library prints there (gloo's "[Gloo] Rank 0 is
        # connected to ..." goes to stdout) is passed on as diagnostics, so the parent's stdout is the JSON line alone
        for line in iter(stream.readline, ""):
            to = sink
            if keep is not None:
                if line.lstrip().startswith("{"):
                    keep.append(line)
                else:
                    to = err
            to.write(line)
            to.flush()
        stream.close()

    try:
        for r in range(n):
            env = dict(os.environ)
            env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                        "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "VSF_BENCH_CHILD": "1",
                        STAGE_ENV: os.path.join(stage_dir, "rank%d" % r)})
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            env.setdefault("OMP_NUM_THREADS", "4")
            env.update(extra_env or {})
            p = subprocess.Popen(cmd + list(argv), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                                 start_new_session=True)
            procs.append(p)
            # rank 0's stdout is the result; the other ranks print nothing there (anything they do goes to stderr)
            t1 = threading.Thread(target=pump, args=(p.stdout, out if r == 0 else err, json_lines if r == 0 else None), daemon=True)
            t2 = threading.Thread(target=pump, args=(p.stderr, err), daemon=True)
            t1.start(), t2.start()
            pumps += [t1, t2]

        def stages():
            out_ = []
            for r in range(n):
                try:
                    ts, name = open(os.path.join(stage_dir, "rank%d" % r)).read().split(None, 1)
                    out_.append((float(ts), name.strip()))
                except (OSError, ValueError):
                    out_.append((0.0, "(no stage reported)"))
            return out_

        def kill_all():
            for p in procs:
                if p.poll() is None:
                    try:
                        os.killpg(p.pid, signal.SIGTERM)
                    except (ProcessLookupError, PermissionError):
                        pass
            t_end = time.time() + 10.0
            for p in procs:
                try:
                    p.wait(max(0.1, t_end - time.time()))
                except subprocess.TimeoutExpired:
                    try:
                        os.killpg(p.pid, signal.SIGKILL)
                    except (ProcessLookupError, PermissionError):
                        pass
                    p.wait()

        t_start = time.time()
        last_change, last_seen = t_start, None
        first_failure = None
        verdict = None
        while True:
            codes = [p.poll() for p in procs]
            if all(c is not None for c in codes):
                break
            now = time.time()
            seen = stages()
            if seen != last_seen:
                last_seen, last_change = seen, now
            if first_failure is None and any(c not in (None, 0) for c in codes):
                first_failure = now
            if first_failure is not None and now - first_failure > grace_s:
                verdict = "rank(s) %s exited non-zero; the others were still running %.0f s later" % (
                    [r for r, c in enumerate(codes) if c not in (None, 0)], grace_s)
            elif now - t_start > deadline_s:
                verdict = "no end after %.0f s" % deadline_s
            elif now - last_change > stall_s:
                verdict = "no rank reported a new stage for %.0f s" % stall_s
            if verdict:
                err.write("bench.py launcher: %s -- killing the ranks.  Last stage per rank:\n" % verdict)
                for r, (ts, name) in enumerate(seen):
                    err.write("  rank %d: %s%s (exit code %s)\n" % (r, name, " at +%.0f s" % (ts - t_start) if ts else "", codes[r]))
                err.flush()
                kill_all()
                break
            time.sleep(0.2)
        for t in pumps:
            t.join(5.0)
        codes = [p.returncode for p in procs]
        if result is not None:
            result.update(codes=codes, verdict=verdict, printed_json=bool(json_lines))
        if verdict:
            bad = [c for c in codes if c not in (None, 0) and c > 0]
            return max(bad) if (bad and first_failure is not None) else 124
        if any(c != 0 for c in codes):
            err.write("bench.py launcher: exit codes per rank %s\n" % codes)
            return max((c if c > 0 else 128 - c) for c in codes if c != 0)
        if not json_lines:
            err.write("bench.py launcher: every rank returned 0 but rank 0 printed no JSON line\n")
            return 5
        return 0
    finally:
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGKILL)
                except (ProcessLookupError, PermissionError):
                    pass
        shutil.rmtree(stage_dir, ignore_errors=True)


def start_rank_watchdog(seconds: float):
    """Inside a rank: if the set-up (process group, handshake, communicator) has not finished after `seconds`, say which
    stage it hangs in and leave with code 124 instead of waiting for the caller's timeout.  Returns the cancel function."""
    import threading
    state = {"stage": "start"}

    def fire():
        sys.stderr.write("bench.py rank %s: set-up did not finish within %.0f s (stage: %s) -- giving up\n"
                         % (os.environ.get("RANK", "0"), seconds, state["stage"]))
        sys.stderr.flush()
        os._exit(124)

    t = threading.Timer(seconds, fire)
    t.daemon = True
    t.start()

    def mark(stage=None):
        if stage is None:
            t.cancel()
        else:
            state["stage"] = stage
    return mark


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="vga")
    ap.add_argument("--batch", type=int, default=None, help="stereo frames per step per GPU")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--nfeatures", type=int, default=None)
    ap.add_argument("--window", type=int, default=1, help="temporal GetFeatureMatches per frame (previous frames)")
    ap.add_argument("--lanes", type=int, default=1, help="concurrent half-batches per step (vsf_set_lanes)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="do not overlap a step's pyramid with the previous step's latency-bound stages (vsf_set_pipeline); "
                         "default: on (frames resident in HBM are complete before the call; the JPEG ingest hands the call an "
                         "event behind its decode: vsf_set_input_event)")
    ap.add_argument("--pipeline", action="store_true", help="(accepted for compatibility: now the default)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="run a step's tail (RemoveAmbigStereo ... payload, collectives) on the extraction's stream instead of "
                         "beside the next step's extraction")
    ap.add_argument("--scene", choices=["bench", "sparse"], default="bench",
                    help="sparse: few objects on a smooth background (~2 %% corner pixels) instead of SURVEY 8(d)'s stream")
    ap.add_argument("--ingest", choices=["hbm", "jpeg", "png"], default="hbm",
                    help="hbm (the metric's definition): frames resident in HBM when the timed region starts.  jpeg: every "
                         "step starts from 2*B baseline-JPEG files in HOST memory (the CompressedImage payloads of "
                         "slam_frontend_main.cc:98-109): upload, vsf_jpeg_decode_gray_batch, Bayer->gray, then the step; an "
                         "extra data point, reported under another metric name (needs Pillow to make the files)")
    ap.add_argument("--blur-inline", action="store_true",
                    help="keep the Gaussian blur on the extraction's stream (vsf_set_blur_overlap(0)); default: it runs on its "
                         "own stream beside FAST and the keypoint selection")
    ap.add_argument("--collectives", choices=["torch", "capi"], default="torch",
                    help="what carries the step's exchanges: torch.distributed (nccl = RCCL; the default, what the driver's "
                         "scaling run uses) or the C ABI's own vsf_allgather_dev / vsf_gather_payload_dev on librccl (the route "
                         "of a C++ host; in a world of one every exchange still runs, through RCCL)")
    ap.add_argument("--fast-resident", type=int, default=None,
                    help="force the FAST launch form (0 = grid, 2..4 = resident with that many waves per SIMD) instead of "
                         "measuring it (experiments)")
    ap.add_argument("--pipe-after-fast", type=int, default=None,
                    help="VSF_OPT_PIPE_AFTER_FAST (experiments): 1 = a step's pipelined pyramid waits for the previous step's "
                         "FAST, 0 = it starts as soon as its inputs are ready; default 1 with frames in HBM, 0 with --ingest jpeg")
    ap.add_argument("--pipe-priority", type=int, default=None,
                    help="VSF_OPT_PIPE_PRIORITY (experiments): stream priority of the pipelined pyramid chain, 0 / 1 low / -1 high")
    ap.add_argument("--ingest-depth", type=int, default=None,
                    help="--ingest jpeg / png: steps whose files one decode call takes (default 1; 2 with PNG puts four decoder waves on every CU, whose LDS the extraction then waits for: no gain, NOTES.md)")
    ap.add_argument("--ingest-priority", choices=["low", "normal", "high"], default=None,
                    help="--ingest jpeg: HIP stream priority of the decode stream")
    ap.add_argument("--match-on-tail", choices=["auto", "on", "off"], default="auto",
                    help="the stereo GetMatches as the first kernel of the step's tail (its matrix-core work beside the next "
                         "step's pyramid and FAST) instead of the last of the extraction; auto: from 5000 features per frame on "
                         "(10 000 features: 21.3 -> 21.9 k frames/s; 2000: 40.8 -> 40.5 k)")
    ap.add_argument("--tune-steps", type=int, default=16,
                    help="whole steps timed per FAST launch form by the set-up's tune call (after three untimed ones)")
    ap.add_argument("--tune-warm", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-observe", action="store_true",
                    help="skip the drop-in API figures (slam::Frontend::ObserveImage one frame at a time, ~3 s)")
    ap.add_argument("--no-inline-pass", action="store_true",
                    help="skip the few untimed steps that give every stage's own duration (counter passes: the run then has "
                         "exactly warm-up + steps steps, which tools/make_traffic.py divides by)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the two short legs on BASELINE's other single-GPU configurations (configs[2] 1920x1080 / 8000 "
                         "features, configs[4] a temporal window of 8; ~10 s each, child processes of this script)")
    ap.add_argument("--cpu-baseline-child", action="store_true",
                    help="(internal) time the oracle library named by VSF_ORACLE_LIB and print the record; no GPU")
    ap.add_argument("--leg", action="store_true",
                    help="(internal) this run IS such a leg: the timed steps and the roofline only")
    ap.add_argument("--no-sustained", action="store_true",
                    help="skip the sustained-throughput leg (>= 600 further steps, ~5 s, after the timed region)")
    ap.add_argument("--sustained-steps", type=int, default=700)
    ap.add_argument("--launch-deadline", type=float, default=900.0,
                    help="bare `--gpus N` form: seconds after which the launching parent kills its ranks")
    ap.add_argument("--launch-stall", type=float, default=420.0,
                    help="bare `--gpus N` form: seconds without any rank reporting a new stage before the parent kills them")
    ap.add_argument("--setup-timeout", type=float, default=360.0,
                    help="N > 1: a rank whose process group / handshake has not come up after this many seconds exits 124")
    ap.add_argument("--no-gloo-retry", action="store_true",
                    help="bare `--gpus N` form: do not repeat a run whose RCCL set-up failed with the collectives on gloo")
    args = ap.parse_args()
    if args.leg:
        args.no_sustained = args.no_observe = args.no_cpu_baseline = args.no_other_configs = True
    cfg = CONFIGS[args.config]
    if args.cpu_baseline_child:
        from vision_slam_frontend_amd import synth as _synth
        print(json.dumps(cpu_baseline(args.width or cfg["width"], args.height or cfg["height"], args.nfeatures or cfg["nfeatures"],
                                      _synth.BASE_SEED, seconds=8.0)))
        return 0
    W = args.width or cfg["width"]
    H = args.height or cfg["height"]
    NF = args.nfeatures or cfg["nfeatures"]
    B = args.batch or cfg["batch"]

    # `python3 bench.py --gpus N` with no launcher around it (no WORLD_SIZE / RANK in the environment): this process becomes
    # the launcher -- N fresh children, one per GPU, BEFORE torch is imported or any GPU call is made here.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        child = json.loads(os.environ["VSF_BENCH_CHILD_CMD"]) if os.environ.get("VSF_BENCH_CHILD_CMD") else None
        res = {}
        rc = launch_ranks(args.gpus, sys.argv[1:], child_cmd=child, deadline_s=args.launch_deadline,
                          stall_s=args.launch_stall, result=res)
        # (exit code 2 is this script's own "cannot run here": fewer GPUs than ranks, no GPU, a world-size mismatch -- no
        # other backend changes that)
        if (rc not in (0, 2) and not res.get("printed_json") and not args.no_gloo_retry and not os.environ.get("VSF_BENCH_ONE_GPU")
                and os.environ.get("VSF_BENCH_BACKEND", "nccl") == "nccl"):
            # The RCCL run did not produce a line.  One labelled second attempt with the step's exchanges on gloo (tensors
            # take a detour through the host; each rank still on its own GPU): a measured curve with its backend named
            # beats no record -- the line says `rccl.backend: gloo` and why.
            sys.stderr.write("bench.py launcher: the RCCL run failed (rc %d, %s); ONE retry with the collectives on gloo\n"
                             % (rc, res.get("verdict") or "exit codes %s" % res.get("codes")))
            rc2 = launch_ranks(args.gpus, sys.argv[1:], child_cmd=child, deadline_s=args.launch_deadline,
                               stall_s=args.launch_stall,
                               extra_env={"VSF_BENCH_BACKEND": "gloo",
                                          "VSF_BENCH_FALLBACK_REASON": "RCCL attempt: rc %d, %s" % (
                                              rc, res.get("verdict") or "exit codes %s" % res.get("codes"))})
            return rc2
        return rc

    # Standard output carries the ONE JSON line and nothing else: whatever a library prints there from here on (RCCL's
    # version banner, gloo's connection notes) goes to standard error instead -- file descriptor 1 is pointed at 2 and the
    # line is written to the saved descriptor at the end.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    mark = start_rank_watchdog(args.setup_timeout) if world > 1 else (lambda stage=None: None)

    def stage(name):
        report_stage(name)
        mark(name)

    stage("import torch")
    import numpy as np  # noqa: F401
    import torch
    import torch.distributed as dist

    from vision_slam_frontend_amd import capi, frontend, synth
    from vision_slam_frontend_amd import distributed as vd

    if args.gpus != world:
        # a run that was asked for N GPUs and got another world size must not pass for an N-GPU measurement
        print("bench.py: --gpus %d but WORLD_SIZE=%d in the environment: either start it bare (`python3 bench.py --gpus %d`, "
              "which launches its own ranks) or under `python -m torch.distributed.run --nnodes=1 --nproc-per-node %d "
              "--master-addr 127.0.0.1 bench.py --gpus %d ...`" % (args.gpus, world, args.gpus, args.gpus, args.gpus),
              file=sys.stderr)
        return 2
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible (the HIP path has no CPU fallback)", file=sys.stderr)
        return 2
    rehearsal = bool(os.environ.get("VSF_BENCH_ONE_GPU"))  # N > 1 control flow on a one-GPU box: ranks share device 0
    if rehearsal:                                          # and talk over gloo (RCCL refuses two ranks on one GPU)
        local_rank = 0
    # VSF_BENCH_BACKEND=gloo: every rank on its own GPU, the exchanges on gloo (the launcher's labelled fallback)
    on_gloo = rehearsal or os.environ.get("VSF_BENCH_BACKEND", "nccl") == "gloo"
    if not rehearsal and local_rank >= torch.cuda.device_count():
        print("bench.py: rank %d wants GPU %d, the node shows %d" % (rank, local_rank, torch.cuda.device_count()), file=sys.stderr)
        return 2
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        stage("init_process_group")
        if on_gloo:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    # the ranks that really take part, seen through the backend that carries the step's collectives (part of the setup)
    stage("handshake")
    handshake = vd.collective_handshake(dev)
    if os.environ.get("VSF_BENCH_FALLBACK_REASON"):
        handshake["fallback_reason"] = os.environ["VSF_BENCH_FALLBACK_REASON"]
    if sorted(handshake["ranks_seen"]) != list(range(world)):
        print("bench.py: the all-gather of rank ids returned %s in a world of %d" % (handshake["ranks_seen"], world),
              file=sys.stderr)
        return 3

    # One explicit (non-default) stream carries everything: the HIP kernels (vsf_set_stream), torch's copies, the
    # per-stage hipEvents and, for N > 1, the RCCL collectives' dependencies.
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    p = capi.default_params(W, H, max_images=2 * B, nfeatures=NF)
    ctx = capi.Context(p, device=local_rank)
    if args.scene == "sparse":
        frames = synth.bench_batch(B, W, H, seed=synth.BASE_SEED + 100003 * rank, n_objects=max(8, W * H // 20000))
    else:
        frames = synth.bench_batch(B, W, H, seed=synth.BASE_SEED + 100003 * rank)
    d_img = torch.from_numpy(frames).to(dev)  # [B, 2, H, W] uint8, resident in HBM before timing
    # Three distinct batches take turns (the second and third are the first one rolled by a few pixels: different images
    # with the same statistics): 3 x 157 MB of input do not stay in the 256-MB Infinity Cache from step to step, so the
    # level-0 reads of every step come out of HBM as they would on a live stream.
    d_imgs = [d_img, torch.roll(d_img, shifts=(3, 11), dims=(2, 3)), torch.roll(d_img, shifts=(8, 29), dims=(2, 3))]
    step_no = [0]
    calib = frontend.default_calibration()
    # the synthetic pairs are rectified (pure horizontal disparity): l^T F r = y_r - y_l
    calib.set("fundamental", [0, 0, 0, 0, 0, -1, 0, 1, 0])
    comm = None
    stage("communicator")
    if args.collectives == "capi":
        if on_gloo:
            print("bench.py: --collectives capi needs one GPU per rank (RCCL refuses two ranks on a device)", file=sys.stderr)
            return 2
        cid = [vd.CapiComm.unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(cid, src=0)  # (the 128 bytes of ncclGetUniqueId travel over the process group)
        comm = vd.CapiComm(ctx, cid[0], rank, world)
        seen = comm.ranks_seen(dev)
        if sorted(seen) != list(range(world)):
            print("bench.py: vsf_allgather_dev of the rank ids returned %s in a world of %d" % (seen, world), file=sys.stderr)
            return 3
        handshake = dict(handshake, backend=comm.name, ranks_seen=seen, nccl_version=str(comm.rccl_version))
    mark()  # the process group and the communicator are up: the set-up watchdog is off
    report_stage("frontend set-up")
    sf = vd.ShardedStereoFrontend(ctx, B, W, H, calib, window=args.window, device=dev, stream=stream,
                                  overlap=not args.no_overlap, comm=comm, match_on_tail=args.match_on_tail == "on" or (args.match_on_tail == "auto" and NF >= 5000))
    sf.keep_outputs = False  # rank 0 receives every payload; the bench does not retain them
    ctx.set_lanes(args.lanes)
    ctx.set_blur_overlap(not args.blur_inline)
    # cross-call pipelining: every step's input is complete in HBM before the call (the rotating synthetic batches), or
    # the call is handed the event behind its producer (the JPEG ingest: vsf_set_input_event)
    pipeline = not args.no_pipeline
    if args.pipe_priority is not None:
        ctx.set_option(capi.OPT_PIPE_PRIORITY, args.pipe_priority)
    ctx.set_pipeline(pipeline)
    if args.pipe_after_fast is not None:
        ctx.set_option(capi.OPT_PIPE_AFTER_FAST, args.pipe_after_fast)
    torch.cuda.synchronize()

    ingest = None
    if args.ingest in ("jpeg", "png"):
        ingest = JpegIngest(frames, W, H, NF, local_rank, dev, stream, priority=args.ingest_priority or "normal", fmt=args.ingest,
                            depth=args.ingest_depth or 1)

    host_s = [0.0, 0.0]  # host wall time inside the ingest call / inside the step's launches (is the host the limit?)

    def run_step():
        h0 = time.perf_counter()
        if ingest is None:
            sf.step(d_imgs[step_no[0] % len(d_imgs)])
            step_no[0] += 1
        else:
            batch, ready = ingest.next_batch()  # starts the next step's decode
            h1 = time.perf_counter()
            host_s[0] += h1 - h0
            h0 = h1
            sf.step(batch, input_event=ready)   # waits (on the GPU) for this step's decode
            ingest.release()                    # the buffer may be overwritten once this step's extraction has read it
        host_s[1] += time.perf_counter() - h0

    # Set-up, not warm-up: ONE explicit, blocking measurement of the two FAST launch forms on this rank's batch
    # (vsf_tune_fast_resident, then six whole steps per form on the rotating batches: the form's worth shows in the composed step), made common
    # over the ranks by one all-reduce -- every rank issues the same collectives whatever it measured, and no library call
    # inside the timed steps measures or waits for anything.
    tune = None
    report_stage("tune")
    if args.fast_resident is not None:
        ctx.set_fast_resident(args.fast_resident)
    elif not args.blur_inline:
        # (with --ingest jpeg the two forms are timed on the real steps, decode beside them: what FAST shares the chip with
        # decides which form wins)
        tune = sf.tune(d_imgs, steps=args.tune_steps if ingest is None else 20, step_fn=None if ingest is None else run_step, warm=args.tune_warm)
    report_stage("warm-up")
    for _ in range(args.warmup):
        run_step()
    sf.drain()
    report_stage("timed steps")
    for c in sf.contexts():
        c.sync(allow_capacity=True)
        c.profile_enable(True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    host_s[0] = host_s[1] = 0.0
    for _ in range(args.steps):
        run_step()
    host_ms = {"ingest_call": 1e3 * host_s[0] / args.steps, "step_launches": 1e3 * host_s[1] / args.steps}
    sf.drain()  # every gather of the timed steps has completed inside the timed region
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    report_stage("after the timed steps")
    # `stages`: the stream that carries the extraction and the stereo matcher (the step's critical path);
    # `tail_stages`: the tail's own stream when it overlaps the next step (wall intervals under contention with the
    # extraction kernels, so they add up to more than the step -- reported separately, never summed into it)
    per_ctx, status = [], capi.VSF_OK
    for c in sf.contexts():
        per_ctx.append(c.profile_read(reset=True))
        c.profile_enable(False)
        status = max(status, c.sync(allow_capacity=True))
    stages = per_ctx[0]
    tail_stages = {k: v for k, v in per_ctx[1].items() if v[1] > 0} if len(per_ctx) > 1 else None
    # With the blur on its own stream its stage timer is the wall span of a kernel that shares the chip with FAST and the
    # selection, not its duration.  A few untimed steps with the blur back in line give every stage's own duration; the
    # roofline figures of the streaming stages use those, the headline value and ms_per_step do not.
    blur_beside = not args.blur_inline and 2 * B >= 32
    inline_stages = None
    if (blur_beside or pipeline or sf.overlap) and not args.no_inline_pass:
        # Every stage by itself: the blur back on the extraction's stream, no cross-step pyramid, and the host waits for the
        # GPU after every step, so that neither the previous step's tail nor -- when it rides the tail stream -- the stereo
        # matcher runs beside the stage being timed (round 5 left the tail beside it: at 10 000 features the pyramid read
        # 3.25 ms "in line" next to the 6-ms matcher, 1.25 ms at 2000 features for the same work).  Stages the tail
        # context times (the tail, the matcher when it is the tail's first kernel) are folded in from that context.
        ctx.set_blur_overlap(False)
        ctx.set_pipeline(False)
        inline_steps = 3
        run_step()
        sf.drain()
        torch.cuda.synchronize()
        for c in sf.contexts():
            c.sync(allow_capacity=True)
            c.profile_enable(True)
        for _ in range(inline_steps):
            run_step()
            sf.drain()
            torch.cuda.synchronize()
        inline_stages = {}
        for c in sf.contexts():
            for k, v in c.profile_read(reset=True).items():
                inline_stages[k] = inline_stages.get(k, 0.0) + v[0] / inline_steps
            c.profile_enable(False)
        ctx.set_blur_overlap(not args.blur_inline)
        ctx.set_pipeline(pipeline)
    rank_ms = [1e3 * elapsed / args.steps]
    blocked_ms, overflow_ranks = 1e3 * sf.blocked_s, [0] if status == capi.VSF_ERR_CAPACITY else []
    if world > 1:
        cdev = "cpu" if on_gloo else dev
        mine = torch.tensor([elapsed, sf.blocked_s, float(status == capi.VSF_ERR_CAPACITY)], dtype=torch.float64, device=cdev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        every = torch.stack(every).cpu()
        rank_ms = [1e3 * float(v) / args.steps for v in every[:, 0]]
        elapsed = float(every[:, 0].max())  # the step time of the job is its slowest rank's
        blocked_ms = 1e3 * float(every[:, 1].max())
        overflow_ranks = [r for r in range(world) if every[r, 2] > 0]

    # Sustained leg: the timed region above is a fraction of a second of GPU time (the contract's K steps); here >= 600
    # further steps (~5 s) of the SAME loop show what the clocks and the power limit leave of it.  Same count on every
    # rank (derived from the slowest rank's step time, which every rank holds), same barriers, max over ranks.
    sustained = None
    if not args.no_sustained:
        report_stage("sustained leg")
        step_s = elapsed / args.steps
        n_sus = 600 if 600 * step_s > 5.0 else max(600, min(args.sustained_steps, int(5.0 / step_s)))
        for _ in range(3):
            run_step()
        sf.drain()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(n_sus // 100 + 1)]
        ts0, ts0_unix = time.perf_counter(), time.time()
        marks[0].record(stream)
        for i in range(n_sus):
            run_step()
            if (i + 1) % 100 == 0:
                marks[(i + 1) // 100].record(stream)
        sf.drain()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        sus_elapsed = time.perf_counter() - ts0
        per100 = [marks[i].elapsed_time(marks[i + 1]) / 100.0 for i in range(len(marks) - 1)]
        if world > 1:
            mine = torch.tensor([sus_elapsed, per100[0], per100[-1]], dtype=torch.float64, device="cpu" if on_gloo else dev)
            every = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(every, mine)
            every = torch.stack(every).cpu()
            sus_elapsed, first100, last100 = (float(every[:, k].max()) for k in range(3))
        else:
            first100, last100 = per100[0], per100[-1]
        sustained = {"steps": n_sus, "seconds": sus_elapsed, "value": world * B * n_sus / sus_elapsed,
                     "ms_per_step": 1e3 * sus_elapsed / n_sus, "first_100_ms": first100, "last_100_ms": last100,
                     "ms_per_step_by_100": per100, "unix_time": [ts0_unix, ts0_unix + sus_elapsed],
                     "note": "the timed loop continued for %d more steps after the headline's region (same settings, drained "
                             "inside the clock); first/last_100_ms: ms per step over the first / last 100 steps (HIP events on "
                             "the extraction's stream, rank 0%s)" % (n_sus, "; max over ranks for the scalars" if world > 1 else "")}

    counts = sf.counts.cpu().numpy()
    nm = sf.nmatches.cpu().numpy()
    nfeat = sf.nfeat.cpu().numpy()
    payload_bytes = int(sf.local_payload(sf.step_idx - 1)[12:16].view(torch.int32).item())
    if rank == 0:
        total_frames = world * B * args.steps
        value = total_frames / elapsed
        alg = stage_algorithmic_bytes(ctx, 2 * B, B, NF)
        pmc, pmc_stale = committed_counters(W, H, NF, B)
        if pmc_stale:
            pmc = {}  # (counters of other kernels than the ones that just ran are not published)
        ceilings, valu_rate_all4, ceilings_stale = valu_ceilings()
        concurrent = (["gauss_blur7"] if blur_beside else []) + (["pyramid_resize"] if pipeline else [])
        # the dominant stage by its OWN duration (the in-line pass when stages share the chip in the timed steps)
        # (a stage the tail context carries -- the stereo matcher as the tail's first kernel -- is looked up there)
        stages_all = dict(stages)
        for k, v in (tail_stages or {}).items():
            if stages_all.get(k, (0.0, 0))[1] == 0:
                stages_all[k] = v
        if inline_stages:
            dom = max((k for k in inline_stages if stages_all.get(k, (0.0, 0))[1] > 0), key=lambda k: inline_stages[k])
        else:
            dom = max((k for k in stages if k not in concurrent), key=lambda k: stages[k][0])
        dom_ms, dom_launches = stages_all[dom]
        fast_waves = ctx.get_fast_resident()
        # FAST as resident workgroups shares every SIMD with the blur: its stage time in the timed steps is then a span
        # under contention as well; its own duration comes from the in-line pass
        shares = ["gauss_blur7"] if (dom == "fast_score_nms" and blur_beside and fast_waves > 0) else []
        per_launch_bytes = alg[dom] * args.steps / max(dom_launches, 1)
        per_launch_s = dom_ms * 1e-3 / max(dom_launches, 1)
        achieved = per_launch_bytes / per_launch_s / 1e9
        traffic = pmc.get(dom, {}).get("hbm_bytes_per_step")
        if traffic is not None:
            traffic = float(traffic) * args.steps / max(dom_launches, 1)
        # VALU roofline of the same kernel: wave64 VALU instructions per step from the committed SQ_INSTS_VALU pass
        # divided by the stage time measured in THIS run, against 1024 SIMDs x 2.4 GHz / 4 cycles per instruction
        valu = None
        insts = pmc.get(dom, {}).get("valu_wave_insts_per_step")
        if insts:
            a = float(insts) * args.steps / (dom_ms * 1e-3) / 1e9
            peak = ceilings.get(dom, valu_rate_all4)
            valu = {"bound": "valu", "kernel": dom, "achieved": a, "peak": peak, "unit": "G wave-inst/s",
                    "frac": a / peak, "valu_wave_insts_per_launch": float(insts) * args.steps / max(dom_launches, 1),
                    "peak_source": "profiles/valu_ceiling.json: this kernel's opcode mix priced with the issue table "
                                   "measured on MI355X (valu_issue_table.json); %.0f G/s if none of its 2-cycle-class "
                                   "instructions pair up%s" % (valu_rate_all4, " -- STALE against csrc/" if ceilings_stale else ""),
                    "source": "profiles/traffic*.json (rocprofv3 --pmc SQ_INSTS_VALU, own pass, same source digest) / stage "
                              "time of this run"}
            if inline_stages:  # the kernel by itself (see roofline.in_line)
                valu["frac_in_line"] = float(insts) / (inline_stages[dom] * 1e-3) / 1e9 / peak
        hbm_frac = achieved / HBM_PEAK_GBS
        bound = "valu" if valu and valu["frac"] > hbm_frac else "hbm"
        # matcher (K9): pair distances per second of the stereo knn2 launches and the int8 matrix-core rate they imply
        # (a 256-bit distance is 256 int8 multiply-adds on the MFMA pipe = 512 ops)
        knn_ms, knn_launches = stages_all.get("hamming_knn2", (0.0, 0))
        matcher = None
        mean_n = float(counts.mean())
        pairs_per_step = B * mean_n * mean_n  # stereo L->R; the R'->L' and temporal launches are ~1 % of that
        if knn_ms > 0:
            dps = pairs_per_step * args.steps / (knn_ms * 1e-3)
            peak = MFMA_FP4_PEAK_TOPS
            matcher = {"pair_distances_per_s": dps, "mfma_tops": dps * 512 / 1e12,
                       "form": "fp4 (v_mfma_scale_f32_32x32x64_f8f6f4, exact)",
                       "frac_of_mfma_peak": dps * 512 / 1e12 / peak, "peak_tops": peak,
                       "ms_per_step": knn_ms / args.steps,
                       "note": "the stereo L->R launch of each step (B pairs of ~N x N)" if tail_stages is not None else
                               "all knn2 launches of a step (stereo + R'->L' + temporal) over the stereo pair count"}
        step_s = elapsed / args.steps
        step_rooflines = {"hbm_algorithmic": {"achieved": sum(alg.values()) / step_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                              "frac": sum(alg.values()) / step_s / 1e9 / HBM_PEAK_GBS}}
        if pmc:
            tr = sum(float(v.get("hbm_bytes_per_step", 0)) for v in pmc.values())
            vi = sum(float(v.get("valu_wave_insts_per_step", 0)) for v in pmc.values())
            step_rooflines["hbm_counted"] = {"achieved": tr / step_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                             "frac": tr / step_s / 1e9 / HBM_PEAK_GBS}
            step_rooflines["valu"] = {"achieved": vi / step_s / 1e9, "peak": VALU_PEAK_GINST, "unit": "G wave-inst/s",
                                      "frac": vi / step_s / 1e9 / VALU_PEAK_GINST,
                                      "note": "against the measured rate of 4-cycle instructions; FAST by itself issues at "
                                              "its mix ceiling, the other stages wait on latency (NOTES.md section 6)"}
        device_ms = sum(v[0] for k, v in stages.items() if k not in concurrent)
        roofline = {"bound": "hbm", "limited_by": None if (pmc_stale or not pmc) else bound, "kernel": dom, "achieved": achieved,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_frac, "traffic": traffic, "traffic_stale": bool(pmc_stale),
                    "algorithmic_bytes_per_launch": per_launch_bytes, "avg_launch_ms": 1e3 * per_launch_s,
                    "launches": dom_launches, "shares_the_chip_with": shares,
                    "in_line": None if not inline_stages else {
                        "avg_launch_ms": inline_stages[dom] * args.steps / max(dom_launches, 1),
                        "achieved": alg[dom] / (inline_stages[dom] * 1e-3) / 1e9,
                        "frac": alg[dom] / (inline_stages[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "note": "the kernel by itself: 3 untimed steps with every stage back in line on one stream"}}
        if dom == "hamming_knn2":
            # the matcher runs on the matrix cores (at nfeatures = 10000 it is the step's longest stage): its roofline is the
            # dense FP4 MFMA rate -- 512 operations per 256-bit distance -- not HBM, which it leaves idle (the HBM figures stay
            # beside it as "hbm")
            ops_per_launch = pairs_per_step * 512 * args.steps / max(dom_launches, 1)
            tops = ops_per_launch / per_launch_s / 1e12
            hbm_view = {k: roofline[k] for k in ("achieved", "peak", "unit", "frac", "algorithmic_bytes_per_launch")}
            roofline.update({"bound": "mfma", "achieved": tops, "peak": MFMA_FP4_PEAK_TOPS, "unit": "TFLOP/s",
                             "frac": tops / MFMA_FP4_PEAK_TOPS, "algorithmic_flops_per_launch": ops_per_launch,
                             "form": "fp4 (v_mfma_scale_f32_32x32x64_f8f6f4, exact); per launch: the step's stereo pair "
                                     "distances / its matcher launches", "hbm": hbm_view})
            del roofline["algorithmic_bytes_per_launch"]
            if sf.match_on_tail and sf.overlap:  # (its span in the timed steps is a span under contention: see in_line)
                roofline["shares_the_chip_with"] = ["the next step's extraction (the matcher is the tail stream's first kernel)"]
            if inline_stages:
                t_in = pairs_per_step * 512 / (inline_stages[dom] * 1e-3) / 1e12
                roofline["in_line"].update({"achieved": t_in, "frac": t_in / MFMA_FP4_PEAK_TOPS})
        out = {
            "metric": ("stereo frames/s (640x480, 2000 kp/frame)" if (W, H, NF) == (640, 480, 2000)
                       else "stereo frames/s (%dx%d, %d kp/frame)" % (W, H, NF)) +
                      (" from %s files in host memory" % args.ingest.upper() if args.ingest != "hbm" else ""),
            "value": value, "unit": "stereo frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%s: %dx%d stereo stream, nfeatures=%d, ORB(1.04, 50 levels, edge 31, FAST 20, Harris) + "
                                   "Hamming 2-NN + ratio 0.6f; extract(L)+extract(R)+GetMatches(L,R), then RemoveAmbigStereo, "
                                   "GetFeatureMatches x%d, Calculate3DPoints, packed VisionFeature/FeatureMatch payload"
                                   % (cfg["label"] if (W, H, NF) == (cfg["width"], cfg["height"], cfg["nfeatures"])
                                      else "custom", W, H, NF, args.window),
                       "frames_per_step_per_gpu": B, "global_frames_per_step": world * B, "scene": args.scene,
                       "ingest": "frames resident in HBM" if args.ingest == "hbm" else
                                 "per step: %d %s files of %.0f KB from host memory -> upload -> "
                                 "vsf_%s_decode_gray_batch -> Bayer->gray, one step ahead on its own stream "
                                 "(HIP stream priority %d: it fills what the extraction leaves free)"
                                 % (ingest.n_files, "grayscale PNG (libpng level 6)" if args.ingest == "png" else "baseline-JPEG (quality 80)",
                                    ingest.avg_kb, args.ingest, ingest.stream_priority),
                       "parallelism": "frames sharded over %d GPU(s)%s" %
                                      (world, ", all-gather of per-frame means + frame tails, compact payload gather to rank 0 "
                                              "(RCCL)" if world > 1 else ""),
                       "blur_overlap": "gauss_blur7 on its own stream beside fast_score_nms / select_harris_angle (its "
                                       "stages_ms_per_step entry is a wall span; stages_ms_per_step_in_line: 3 untimed "
                                       "steps with every stage back in line)" if blur_beside else "off",
                       "input_rotation": "3 distinct %d-frame batches in turn (%.0f MB of input; the Infinity Cache holds 256 MB)"
                                         % (B, 3 * B * 2 * W * H / 1e6)
                                         if args.ingest == "hbm" else "per-step decode",
                       "fast_resident": ("%s: %s" % ("measured before the run (vsf_tune_fast_resident), same on every rank" if tune
                                         else "default", "one resident workgroup per CU, %d waves per SIMD (the blur "
                                         "finishes inside the FAST pass)" % fast_waves if fast_waves > 0 else
                                         "one workgroup per four cells")) if blur_beside else "off",
                       "fast_tune": tune,
                       "pipeline": "step s + 1's pyramid beside step s's selection / descriptors / matcher (vsf_set_pipeline)"
                                   if pipeline else "off",
                       "tail_overlap": "step s's tail + collectives on a second stream beside step s+1's extraction"
                                       if sf.overlap else "off (one stream)",
                       "stereo_match": "first kernel of the tail" if sf.match_on_tail else "last kernel of the extraction",
                       "mean_keypoints_per_image": float(counts.mean()), "mean_stereo_matches": float(nm.mean()),
                       "mean_features_per_frame": float(nfeat.mean()), "payload_bytes_per_step_per_gpu": payload_bytes,
                       "parity": "bit-exact vs the in-repo oracle (a restatement of OpenCV 3.2; parity with OpenCV itself unpinned)",
                       "capacity_overflow": bool(overflow_ranks)},
            # "bound" names the roofline this object is measured against (the contract knows "hbm" and "mfma");
            # "limited_by" says what the counters show the kernel is actually limited by (see "roofline_valu")
            "roofline": roofline,
            # self-verification of a multi-GPU run: what the process group says it is, the ranks an all-gather of rank ids
            # saw on that backend, every rank's own step time, the longest any rank's host waited inside a gather
            "rccl": dict(handshake, per_rank_ms_per_step={"min": min(rank_ms), "max": max(rank_ms)},
                         max_rank_blocked_in_gather_ms=blocked_ms, capacity_overflow_ranks=overflow_ranks),
            "sustained": sustained,
            "roofline_valu": valu,
            # the whole step against the two rooflines: every stage's algorithmic bytes (and counted HBM bytes, and wave64
            # VALU instructions from the committed counter passes) over the step time of this run
            "step_rooflines": step_rooflines,
            "matcher": matcher,
            # every streaming stage against the same roofline (algorithmic bytes / measured stage time)
            "streaming_stages_gbs": {k: alg[k] / ((inline_stages[k] if inline_stages else stages[k][0] / args.steps) * 1e-3) / 1e9
                                     for k in ("pyramid_resize", "fast_score_nms", "gauss_blur7") if stages[k][0] > 0},
            "stages_ms_per_step": {k: v[0] / args.steps for k, v in stages.items()},
            "concurrent_stages": concurrent,
            "stages_ms_per_step_in_line": inline_stages,
            "device_ms_per_step": device_ms / args.steps,
            # host wall time per step spent issuing work (the step is asynchronous: well below ms_per_step = the GPU is the limit)
            "host_issue_ms_per_step": host_ms,
            "tail_stream_ms_per_step": None if tail_stages is None else
            {k: v[0] / args.steps for k, v in tail_stages.items()},
        }
        # The drop-in API (slam::Frontend::ObserveImage, cc:400-472, through the C++ host class): NOT the benchmarked value,
        # reported beside it, at 2000 features and at the reference's own 10000.
        #   observe_image_ms             synchronous latency per call (median; the call returns with the node booked)
        #   observe_image_unchanged_fps  the reference's driver unchanged: GetSLAMProblem after every node (main.cc:320-321)
        #   observe_image_pipelined_fps  Frontend::set_pipelined(true): frames wait in the context's queue (depth 256) and
        #                                leave for the GPU in batches of up to 128; the reference's driver loop in C++
        #                                (vsfh_time_sequence), median of three runs, all three listed
        out["observe_image"] = None
        report_stage("observe_image leg")
        if world == 1 and not args.no_observe and (W, H) == (640, 480):
            sys.path.insert(0, str(ROOT / "tools"))
            import time_frontend as tf
            obs = {}
            for nf in (2000, 10000):
                ms, _, _, _ = tf.observe_image_ms(nf, True, n_frames=96)
                unchanged = tf.queued_fps(nf, n_frames=160, pipelined=False, read_every=1)[0][0]
                runs = sorted(r[0] for r in tf.queued_fps(nf, repeats=3))
                obs["nfeatures_%d" % nf] = {"observe_image_ms": ms, "observe_image_unchanged_fps": unchanged,
                                            "observe_image_pipelined_fps": runs[1], "observe_image_pipelined_runs": runs}
            obs["note"] = ("640x480, frame_life 10, window full; per stereo frame through slam::Frontend (host/slam_frontend.cc); "
                           "pipelined: queue depth 256, <= 128 frames per batch, 3200 steady frames per run")
            out["observe_image"] = obs
        # BASELINE's other single-GPU configurations, each a short run of this same script in a child process (its own
        # contexts, tune call, timed steps and in-line pass; the parent's GPU work is over): configs[2] = 1920x1080 / 8000
        # features, configs[4] = a temporal window of 8 frames (the multi-query matcher launch).
        out["other_configs"] = None
        report_stage("other configs")
        if (world == 1 and not args.no_other_configs and args.config == "vga" and args.window == 1 and args.ingest == "hbm"
                and args.nfeatures is None and args.batch is None):
            other = {}
            for name, extra in (("1080p_8000", ["--config", "1080p", "--batch", "64"]), ("window8", ["--window", "8"])):
                cmd = [sys.executable, str(ROOT / "bench.py"), "--leg", "--steps", "8", "--warmup", "2", "--tune-steps", "4"] + extra
                t_leg = time.perf_counter()
                try:
                    r = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
                    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
                    j = json.loads(line[-1]) if line else None
                except (subprocess.TimeoutExpired, ValueError) as e:
                    r, j = None, None
                    other[name] = {"error": repr(e)}
                if j is not None:
                    rf = j.get("roofline") or {}
                    other[name] = {"value": j["value"], "unit": j["unit"], "ms_per_step": j["ms_per_step"], "steps": j["steps"],
                                   "config": {k: j["config"].get(k) for k in ("workload", "frames_per_step_per_gpu", "stereo_match",
                                                                               "mean_keypoints_per_image", "mean_features_per_frame")},
                                   "roofline": {"kernel": rf.get("kernel"), "bound": rf.get("bound"), "frac": rf.get("frac"),
                                                "achieved": rf.get("achieved"), "unit": rf.get("unit"),
                                                "frac_in_line": (rf.get("in_line") or {}).get("frac")},
                                   "stages_ms_per_step_in_line": j.get("stages_ms_per_step_in_line"),
                                   "seconds": time.perf_counter() - t_leg}
                elif r is not None:
                    other[name] = {"error": "rc %d: %s" % (r.returncode, (r.stderr or "")[-400:])}
            out["other_configs"] = other
        report_stage("cpu_baseline leg")
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(W, H, NF, synth.BASE_SEED)
            # ... and the same sources with -march=native: a floor (above) and a fairer figure (here) for what the
            # reference's -O3 build (CMakeLists.txt:12-14) with OpenCV's SSE2 / parallel_for_ paths would give
            report_stage("cpu_baseline_native leg")
            out["cpu_baseline_native"] = cpu_baseline_native(W, H, NF, synth.BASE_SEED, out["cpu_baseline"]["digest"])
        else:
            out["cpu_baseline"] = None
            out["cpu_baseline_native"] = None
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    report_stage("teardown")
    sf.close()
    if comm is not None:
        comm.close()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()
    if overflow_ranks:
        if rank == 0:
            print("bench.py: keypoint / match capacity overflowed on rank(s) %s: outputs were truncated, the line above is "
                  "not a valid measurement" % overflow_ranks, file=sys.stderr)
        return 4
    return 0


if __name__ == "__main__":
    sys.exit(main())
