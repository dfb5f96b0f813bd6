#!/usr/bin/env python3
"""Headline benchmark: stereo frames/s through extract(left) + extract(right) + GetMatches(left, right)
(slam_frontend.cc:411-416) on synthetic 640x480 stereo pairs, 2000 keypoints per frame (BASELINE.json
configs[1]), inputs resident in HBM, on N GPUs of one node (one process per GPU, frames sharded, outputs
gathered to rank 0 over RCCL).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of `--batch` stereo frames per GPU.  Rank 0 prints ONE
JSON line (schema: task contract + `roofline` and `cpu_baseline` objects).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


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
    }


def measured_traffic(stage: str, width: int, height: int, nfeatures: int, batch: int):
    """HBM bytes per step of `stage` from the committed rocprofv3 --pmc passes (profiles/traffic.json: FETCH_SIZE and
    WRITE_SIZE collected in their own runs, gfx950 FETCH_SIZE correction applied), if they were taken on this
    configuration; None otherwise.  PMC counters cannot be collected inside the timed run."""
    try:
        t = json.loads((ROOT / "profiles" / "traffic.json").read_text())
    except (OSError, ValueError):
        return None
    c = t.get("config", {})
    if (c.get("width"), c.get("height"), c.get("nfeatures"), c.get("batch")) != (width, height, nfeatures, batch):
        return None
    st = t.get("stages", {}).get(stage)
    return None if st is None else float(st["hbm_bytes_per_step"])


def cpu_baseline(width, height, nfeatures, seed):
    """The CPU oracle (a scalar C++ restatement, kind "port") timed on this box's host cores."""
    from concurrent.futures import ThreadPoolExecutor

    from oracle import binding as ob
    from vision_slam_frontend_amd import synth

    ob.build()
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, 16))
    n = 6 * cores  # ~20 s of CPU work (about 0.2 s per stereo frame and thread)
    frames = synth.bench_batch(n, width, height, seed=seed, n_scenes=min(4, n))

    def one(i):
        a, b = ob.Orb(nfeatures=nfeatures), ob.Orb(nfeatures=nfeatures)
        a.run(frames[i, 0])
        b.run(frames[i, 1])
        _, da = a.result()
        _, db = b.result()
        return len(ob.get_matches(da, db))

    one(0)  # warm (page in, build tables)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(one, range(n)))
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "stereo frames/s", "cores": cores, "kind": "port",
            "sample": "%d synthetic %dx%d stereo frames (%d kp), oracle extract(L)+extract(R)+GetMatches, "
                      "%d threads, %.1f s wall" % (n, width, height, nfeatures, cores, dt)}


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="stereo frames per step per GPU")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--nfeatures", type=int, default=2000)
    ap.add_argument("--lanes", type=int, default=1, help="concurrent half-batches per step (vsf_set_lanes)")
    ap.add_argument("--pipeline", action="store_true",
                    help="overlap a step's pyramid with the previous step's latency-bound tail (vsf_set_pipeline): "
                         "+4.7 %% frames/s, but the per-stage timers then overlap; off for the reported line")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--traffic", type=float, default=None,
                    help="HBM bytes per launch of the dominant kernel from a separate rocprofv3 --pmc run")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    from vision_slam_frontend_amd import capi, synth
    from vision_slam_frontend_amd import distributed as vd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            print("bench.py: --gpus %d needs `python -m torch.distributed.run --nproc-per-node %d`" %
                  (args.gpus, args.gpus), file=sys.stderr)
            return 2
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible (the HIP path has no CPU fallback)", file=sys.stderr)
        return 2
    rehearsal = bool(os.environ.get("VSF_BENCH_ONE_GPU"))  # N > 1 control flow on a one-GPU box: ranks share device 0
    if rehearsal:                                          # and talk over gloo (RCCL refuses two ranks on one GPU)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    B, W, H, NF = args.batch, args.width, args.height, args.nfeatures
    # One explicit (non-default) stream carries everything: buffer initialisation, the HIP kernels (vsf_set_stream),
    # the per-stage hipEvents and, for N > 1, the RCCL gather.  (torch's default stream has handle 0, which
    # vsf_set_stream reads as "use the context's own stream".)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    p = capi.default_params(W, H, max_images=2 * B, nfeatures=NF)
    ctx = capi.Context(p, device=local_rank)
    K = ctx.params.max_keypoints
    frames = synth.bench_batch(B, W, H, seed=synth.BASE_SEED + 100003 * rank)
    d_img = torch.from_numpy(frames).to(dev)  # [B, 2, H, W] uint8, resident in HBM before timing
    d_kp = torch.empty((2 * B, K, 28), dtype=torch.uint8, device=dev)
    d_desc = torch.empty((2 * B, K, 32), dtype=torch.uint8, device=dev)
    d_counts = torch.zeros(2 * B, dtype=torch.int32, device=dev)
    d_matches = torch.empty((B, K, 16), dtype=torch.uint8, device=dev)
    d_nmatches = torch.zeros(B, dtype=torch.int32, device=dev)
    gather_bufs = [None, None]
    if world > 1 and rank == 0:
        payload_bytes = B * K * 28 + 2 * B * 4 + B * K * 16 + B * 4
        gather_bufs = [[torch.empty(payload_bytes, dtype=torch.uint8, device=dev) for _ in range(world)]
                       for _ in range(2)]  # two sets: the gather of step i is in flight while step i + 1 is computed
    pending = []  # (work, payload, recv) of the gathers in flight
    assert stream.cuda_stream != 0
    ctx.set_stream(stream.cuda_stream)
    ctx.set_lanes(args.lanes)
    ctx.set_pipeline(args.pipeline)  # (legal here: the synthetic stream is resident in HBM before every call)
    torch.cuda.synchronize()

    def step():
        ctx.stereo_batch_dev(d_img.data_ptr(), B, W * H, W, d_kp.data_ptr(), d_desc.data_ptr(),
                             d_counts.data_ptr(), d_matches.data_ptr(), d_nmatches.data_ptr())
        if world > 1:
            # VisionFeature (left keypoints) / FeatureMatch payloads of this rank's frames -> rank 0 over RCCL, as an
            # asynchronous collective on the backend's stream: it overlaps the next step's kernels (the payload is a
            # packed copy, so the next step may overwrite the output buffers; at most two gathers are in flight).
            if len(pending) >= 2:
                pending.pop(0)[0].wait()
            left_kp = d_kp.view(B, 2, K, 28)[:, 0].contiguous()
            pending.append(vd.gather_packed_to_root_async(
                {"kp": left_kp, "counts": d_counts, "matches": d_matches, "nmatches": d_nmatches}, dst=0,
                bufs=gather_bufs[step.n & 1]))
            step.n += 1

    step.n = 0

    def drain():
        while pending:
            pending.pop(0)[0].wait()

    for _ in range(args.warmup):
        step()
    drain()
    ctx.sync(allow_capacity=True)
    ctx.profile_enable(True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()  # every gather of the timed steps has completed inside the timed region
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    stages = ctx.profile_read(reset=True)
    ctx.profile_enable(False)
    status = ctx.sync(allow_capacity=True)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    counts = d_counts.cpu().numpy()
    nm = d_nmatches.cpu().numpy()
    if rank == 0:
        total_frames = world * B * args.steps
        value = total_frames / elapsed
        alg = stage_algorithmic_bytes(ctx, 2 * B, B, NF)
        dom = max(stages, key=lambda k: stages[k][0])
        dom_ms, dom_launches = stages[dom]
        per_launch_bytes = alg[dom] * args.steps / max(dom_launches, 1)
        per_launch_s = dom_ms * 1e-3 / max(dom_launches, 1)
        achieved = per_launch_bytes / per_launch_s / 1e9
        traffic = args.traffic
        if traffic is None:
            t_step = measured_traffic(dom, W, H, NF, B)
            if t_step is not None:
                traffic = t_step * args.steps / max(dom_launches, 1)
        device_ms = sum(v[0] for v in stages.values())
        out = {
            "metric": "stereo frames/s (640x480, 2000 kp/frame)" if (W, H, NF) == (640, 480, 2000)
            else "stereo frames/s (%dx%d, %d kp/frame)" % (W, H, NF),
            "value": value, "unit": "stereo frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: %dx%d stereo stream, nfeatures=%d, ORB(1.04, 50 levels, "
                                   "edge 31, FAST 20, Harris) + Hamming 2-NN + ratio 0.6f; extract(L)+extract(R)+"
                                   "GetMatches(L,R)" % (W, H, NF),
                       "frames_per_step_per_gpu": B, "global_frames_per_step": world * B,
                       "parallelism": "frames sharded over %d GPU(s)%s" %
                                      (world, ", RCCL gather of keypoints+matches to rank 0" if world > 1 else ""),
                       "mean_keypoints_per_image": float(counts.mean()), "mean_stereo_matches": float(nm.mean()),
                       "capacity_overflow": bool(status == capi.VSF_ERR_CAPACITY)},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": per_launch_bytes, "avg_launch_ms": 1e3 * per_launch_s,
                         "launches": dom_launches},
            # every streaming stage against the same roofline (algorithmic bytes / measured stage time)
            "streaming_stages_gbs": {k: alg[k] * args.steps / (stages[k][0] * 1e-3) / 1e9
                                     for k in ("pyramid_resize", "fast_score_nms", "gauss_blur7") if stages[k][0] > 0},
            "stages_ms_per_step": {k: v[0] / args.steps for k, v in stages.items()},
            "device_ms_per_step": device_ms / args.steps,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(W, H, NF, synth.BASE_SEED)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    ctx.close()
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
