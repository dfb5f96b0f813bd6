"""BASELINE configs[3] with real kernels: two ranks share ONE GPU (gloo between them -- RCCL refuses two ranks on one
device) and run the sharded hot path of vision_slam_frontend_amd.distributed.ShardedStereoFrontend: all-gather of the
per-frame means -> device thresholds -> RemoveAmbigStereo filter -> tail exchange for the temporal pairs ->
Calculate3DPoints -> compact VisionFeature / FeatureMatch payload -> gather to rank 0.  Rank 0's gathered frames must
be BYTE-IDENTICAL to a single-process run over the same frames -- including the RemoveAmbigStereo threshold that
crosses the rank boundary (slam_frontend.cc:353, 392-398), its NaN case (quirk Q3) and the temporal predecessor that
lives on the other rank."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

ROOT = Path(__file__).resolve().parent.parent
W_IMG, H_IMG, NF = 320, 240, 600
B, STEPS, WORLD, WINDOW = 3, 3, 2, 2
F_RECT = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32)  # l^T F r = y_r - y_l on the rectified synthetic pair


def _calibration():
    from vision_slam_frontend_amd import frontend
    return frontend.default_calibration().set("fundamental", F_RECT)


def _run(frames, frames_per_rank, rank, world, overlap=True, force_collectives=False):
    """All steps of one rank; returns the ShardedStereoFrontend (drained).  overlap: the tail of a step on a second
    stream beside the next step's extraction (the default) or everything on one stream."""
    from vision_slam_frontend_amd import capi
    from vision_slam_frontend_amd import distributed as vd
    dev = torch.device("cuda", 0)
    ctx = capi.Context(capi.default_params(W_IMG, H_IMG, max_images=2 * frames_per_rank, nfeatures=NF))
    sf = vd.ShardedStereoFrontend(ctx, frames_per_rank, W_IMG, H_IMG, _calibration(), window=WINDOW, device=dev,
                                  overlap=overlap, force_collectives=force_collectives)
    local = []
    for s in range(STEPS):
        idx = list(vd.frame_block(s, frames_per_rank, world, rank))
        d_img = torch.from_numpy(np.ascontiguousarray(frames[idx])).to(dev)
        sf.step(d_img)
        if world == 1:
            sf.synchronize()
            local.append((s, [sf.local_payload(s).cpu().clone()]))
    sf.drain()
    assert all(c.sync() == capi.VSF_OK for c in sf.contexts())
    sf.close()
    return sf, local, ctx


def _worker(rank: int, world: int, port: int, frames_path: str, out_path: str, frames_per_rank: int = B):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.path.insert(0, str(ROOT))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        frames = np.load(frames_path)
        sf, _, ctx = _run(frames, frames_per_rank, rank, world)
        if rank == 0:
            assert [c[0] for c in sf.completed] == list(range(STEPS))
            np.savez(out_path, **{"s%d_r%d" % (st, r): pl.cpu().numpy() for st, per in sf.completed
                                  for r, pl in enumerate(per)})
        else:
            assert not sf.completed
        ctx.close()
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_ranks_equal_one_process(tmp_path):
    import torch.multiprocessing as mp

    from vision_slam_frontend_amd import distributed as vd
    from vision_slam_frontend_amd import synth

    n = WORLD * B * STEPS
    frames = synth.stereo_stream(n, W_IMG, H_IMG, n_objects=400)
    frames[B - 1, 1] = 128  # rank 0's last frame of step 0 has no stereo match: rank 1's first frame gets the NaN threshold
    frames_path, out_path = str(tmp_path / "frames.npy"), str(tmp_path / "gathered.npz")
    np.save(frames_path, frames)

    # single process: the same frames, world x B per step
    sf1, local, ctx1 = _run(frames, WORLD * B, 0, 1)
    want_f, want_m = vd.assemble_outputs(local, 1, WORLD * B, WINDOW)
    ctx1.close()
    assert sorted(want_f) == list(range(n))

    mp.spawn(_worker, args=(WORLD, _free_port(), frames_path, out_path), nprocs=WORLD, join=True)
    z = np.load(out_path)
    completed = [(s, [z["s%d_r%d" % (s, r)] for r in range(WORLD)]) for s in range(STEPS)]
    got_f, got_m = vd.assemble_outputs(completed, WORLD, B, WINDOW)

    assert sorted(got_f) == sorted(want_f) and sorted(got_m) == sorted(want_m)
    for g in range(n):
        assert got_f[g].tobytes() == want_f[g].tobytes(), "VisionFeature records of global frame %d" % g
    for key in want_m:
        assert got_m[key].tobytes() == want_m[key].tobytes(), "FeatureMatch records of factor %s" % (key,)
    # the cases this test exists for really occurred
    sizes = [len(want_f[g]) for g in range(n)]
    assert sizes[B - 1] == 0 and sizes[B] == 0, "empty-match frame, then the NaN-threshold frame on the OTHER rank"
    assert sizes[B + 1] > 20 and sizes[0] > sizes[1] > 0, "filtering resumes; later frames are filtered by mean + 2"
    cross = [k for k in want_m if k[0] // B != k[1] // B]  # predecessor on another rank (or the previous step)
    assert len(cross) >= 2 * (WORLD * STEPS - 1) and sum(len(want_m[k]) for k in cross) > 20
    assert sum(len(v) for v in want_m.values()) > 40
    # and the payloads were compact: sized by the counts, not by the capacity
    assert all(int(pl[:16].view(np.uint32)[3]) < sf1.cap // 4 for _, per in completed for pl in per)


def _rccl_worker(rank: int, port: int, frames_path: str, out_path: str):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.path.insert(0, str(ROOT))
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from vision_slam_frontend_amd import distributed as vd
        hs = vd.collective_handshake(torch.device("cuda", 0))  # what bench.py prints as "rccl" for the first 8-GPU run
        assert hs["backend"].startswith("nccl") and hs["world"] == 1 and hs["ranks_seen"] == [0] and hs["nccl_version"]
        frames = np.load(frames_path)
        sf, _, ctx = _run(frames, WORLD * B, 0, 1, overlap=False, force_collectives=True)  # (the one-stream form, for coverage)
        assert sf.dist_on and not sf.host_detour and [c[0] for c in sf.completed] == list(range(STEPS))
        assert sf.blocked_s >= 0.0
        np.savez(out_path, **{"s%d" % st: per[0].cpu().numpy() for st, per in sf.completed})
        ctx.close()
    finally:
        dist.destroy_process_group()


def test_collectives_on_rccl_in_a_world_of_one(tmp_path):
    """The same class on the REAL backend (nccl = RCCL) with device tensors: a one-GPU box cannot host two RCCL ranks, so
    this runs every collective of the step (three all-gathers, the size exchange, the asynchronous sized gather with its
    slot rotation) in a world of one and compares the gathered payloads with the collective-free run."""
    import torch.multiprocessing as mp

    from vision_slam_frontend_amd import distributed as vd
    from vision_slam_frontend_amd import synth

    n = WORLD * B * STEPS
    frames = synth.stereo_stream(n, W_IMG, H_IMG, n_objects=400)
    frames_path, out_path = str(tmp_path / "frames.npy"), str(tmp_path / "gathered.npz")
    np.save(frames_path, frames)
    sf1, local, ctx1 = _run(frames, WORLD * B, 0, 1)
    ctx1.close()
    mp.spawn(_rccl_worker, args=(_free_port(), frames_path, out_path), nprocs=1, join=True)
    z = np.load(out_path)
    for s, per in local:
        want = per[0].numpy()
        total = int(want[:16].view(np.uint32)[3])
        got = z["s%d" % s]
        assert len(got) >= total and got[:total].tobytes() == want[:total].tobytes(), "payload of step %d" % s
    want_f, want_m = vd.assemble_outputs(local, 1, WORLD * B, WINDOW)
    assert sum(len(v) for v in want_f.values()) > 200 and sum(len(v) for v in want_m.values()) > 40


def test_three_ranks_equal_one_process(tmp_path):
    """World of three, two frames per rank, window 2 (= the whole block of a rank): the middle rank has its predecessors on
    rank 0 and its successors on rank 2, every temporal pair of a rank's first frame lives on another rank, rank 0 fetches
    its predecessors from rank 2's tail of the PREVIOUS step, and the NaN threshold (quirk Q3) crosses from rank 1's
    first to its second frame while the finite one after it crosses into rank 2."""
    import torch.multiprocessing as mp

    from vision_slam_frontend_amd import distributed as vd
    from vision_slam_frontend_amd import synth

    world, per = 3, 2
    n = world * per * STEPS
    frames = synth.stereo_stream(n, W_IMG, H_IMG, n_objects=400)
    frames[per, 1] = 128  # rank 1's first frame of step 0 has no stereo match: its second frame meets the NaN threshold
    frames_path, out_path = str(tmp_path / "frames.npy"), str(tmp_path / "gathered.npz")
    np.save(frames_path, frames)
    sf1, local, ctx1 = _run(frames, world * per, 0, 1)
    want_f, want_m = vd.assemble_outputs(local, 1, world * per, WINDOW)
    ctx1.close()
    mp.spawn(_worker, args=(world, _free_port(), frames_path, out_path, per), nprocs=world, join=True)
    z = np.load(out_path)
    completed = [(s, [z["s%d_r%d" % (s, r)] for r in range(world)]) for s in range(STEPS)]
    got_f, got_m = vd.assemble_outputs(completed, world, per, WINDOW)
    assert sorted(got_f) == sorted(want_f) == list(range(n)) and sorted(got_m) == sorted(want_m)
    for g in range(n):
        assert got_f[g].tobytes() == want_f[g].tobytes(), "VisionFeature records of global frame %d" % g
    for key in want_m:
        assert got_m[key].tobytes() == want_m[key].tobytes(), "FeatureMatch records of factor %s" % (key,)
    sizes = [len(want_f[g]) for g in range(n)]
    assert sizes[per] == 0 and sizes[per + 1] == 0 and sizes[per + 2] > 10
    assert (per * 3 - 2, per * 3) in want_m and (per * 3 - 1, per * 3) in want_m  # rank 0, step 1 <- rank 2, step 0


def test_sharded_outputs_follow_the_reference_sequence(oracle):
    """Not a self-comparison: the payloads ShardedStereoFrontend gathers (world of one, two steps of three frames, window
    2) against the reference's sequence assembled from the CPU oracle's pieces -- extract L / R, GetMatches,
    RemoveAmbigStereo with the threshold carried from frame to frame (slam_frontend.cc:353-398), GetFeatureMatches
    against the two previous frames (cc:282-309, 424-434), Calculate3DPoints / UndistortFeaturePoints (cc:117-173,
    323-351).  Indices bit for bit, point3d / pixel to the tolerance of row f2."""
    from vision_slam_frontend_amd import distributed as vd
    from vision_slam_frontend_amd import synth

    steps, per_step = 2, 3
    n = steps * per_step
    frames = synth.stereo_stream(n, W_IMG, H_IMG, n_objects=400)
    frames[1, 1] = 128  # frame 1 has no stereo match: frame 2 meets the NaN threshold (quirk Q3), frame 3 is normal again
    global STEPS
    saved, STEPS = STEPS, steps
    try:
        sf, local, ctx = _run(frames, per_step, 0, 1)
    finally:
        STEPS = saved
    got_f, got_m = vd.assemble_outputs(local, 1, per_step, WINDOW)
    calib = _calibration()
    ctx.close()
    bp = float(np.float32(0.3))
    thr = np.float32(10000.0)
    filtered = []  # filtered left descriptors per frame
    sizes = []
    for g in range(n):
        left, right = frames[g, 0], frames[g, 1]
        ol, orr = oracle.Orb(nfeatures=NF), oracle.Orb(nfeatures=NF)
        ol.run(left)
        orr.run(right)
        kl, dl = ol.result()
        kr, dr = orr.result()
        m = oracle.get_matches(dl, dr)
        keep, _, thr_next, kept = oracle.remove_ambig_stereo(kl, kr, m, F_RECT, float(thr))
        kl2, dl2 = kl[m["queryIdx"][keep]], dl[m["queryIdx"][keep]]
        kr2, dr2 = kr[m["trainIdx"][keep]], dr[m["trainIdx"][keep]]
        want, npts = oracle.vision_features(kl2, dl2, kr2, dr2, calib.get("projection_left"), calib.get("projection_right"),
                                            calib.get("camera_matrix_left"), calib.get("distortion_left"))
        f = got_f[g]
        assert len(f) == len(want) == kept, "frame %d" % g
        np.testing.assert_array_equal(f["feature_idx"], want["feature_idx"])
        assert np.abs(f["pixel"].astype(np.float64) - want["pixel"]).max(initial=0.0) <= 1e-4
        a, b = f["point3d"].astype(np.float64), want["point3d"].astype(np.float64)
        fin = np.isfinite(b)
        assert np.array_equal(np.isfinite(a), fin)
        assert (np.abs(a[fin] - b[fin]) / np.maximum(np.abs(b[fin]), 1e-30)).max(initial=0.0) <= 1e-5
        for w in range(WINDOW, 0, -1):
            if g - w < 0:
                continue
            mm = oracle.sort_and_trim(oracle.get_matches(filtered[g - w], dl2), bp)
            fac = got_m[(g - w, g)]
            np.testing.assert_array_equal(fac["feature_idx_initial"], mm["queryIdx"], err_msg="factor (%d, %d)" % (g - w, g))
            np.testing.assert_array_equal(fac["feature_idx_current"], mm["trainIdx"])
        filtered.append(dl2)
        sizes.append(kept)
        thr = np.float32(thr_next)
    assert sizes[1] == 0 and sizes[2] == 0 and sizes[3] > 20 and sizes[0] > 20
    assert sum(len(v) for v in got_m.values()) > 5


def _run_rank(frames, w, h, nf, frames_per_rank, window, steps, rank, world, comm=None, tune=False):
    """All steps of one rank of a `world`-rank job at any geometry; returns (ShardedStereoFrontend, local payloads, tune)."""
    from vision_slam_frontend_amd import capi
    from vision_slam_frontend_amd import distributed as vd
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    ctx = capi.Context(capi.default_params(w, h, max_images=2 * frames_per_rank, nfeatures=nf))
    sf = vd.ShardedStereoFrontend(ctx, frames_per_rank, w, h, _calibration(), window=window, device=dev, comm=comm)
    local, tuned = [], None
    if tune:
        idx = list(vd.frame_block(0, frames_per_rank, world, rank))
        tuned = sf.tune(torch.from_numpy(np.ascontiguousarray(frames[idx])).to(dev), samples=1)
    for s in range(steps):
        idx = list(vd.frame_block(s, frames_per_rank, world, rank))
        d_img = torch.from_numpy(np.ascontiguousarray(frames[idx])).to(dev)
        sf.step(d_img)
        if world == 1:
            sf.synchronize()
            local.append((s, [sf.local_payload(s).cpu().clone()]))
    sf.drain()
    assert all(c.sync() == capi.VSF_OK for c in sf.contexts())
    sf.close()
    completed = list(sf.completed)
    ctx.close()
    return sf, local, tuned, completed


def test_eight_ranks_one_frame_per_gpu():
    """BASELINE configs[3] literally -- "640x480 stereo batch sharded one-frame-per-GPU", eight ranks -- as far as one GPU
    allows: a world of EIGHT with ONE frame per rank and step, 640x480 / 2000 features, window 1, three steps (24 frames),
    real kernels, byte-identical to a single process over the same 24 frames.  A GPU box admits at most six processes on
    its card, so the eight ranks are eight THREADS of this process, each with its own context, streams and
    ShardedStereoFrontend, exchanging through distributed.ThreadComm (the class takes whatever carries its bytes; the
    torch.distributed carriers are covered by the 2- and 3-process tests above and by the RCCL world of one).  Reference
    dependencies being sharded: the static threshold of RemoveAmbigStereo crossing EVERY frame boundary = rank boundary
    (slam_frontend.cc:353, 392-394), a frame without stereo matches inside a step and at a step's end (its NaN threshold
    lands on the next rank / on rank 0 of the next step, quirk Q3), the temporal predecessor that always lives on another
    rank (cc:424-434).  The explicit tune call takes part as in bench.py: every rank issues its one all-reduce."""
    import threading

    from vision_slam_frontend_amd import distributed as vd
    from vision_slam_frontend_amd import synth

    world, per, window, steps, w, h, nf = 8, 1, 1, 3, 640, 480, 2000
    n = world * per * steps
    frames = synth.stereo_stream(n, w, h, n_objects=400)
    frames[2, 1] = 128   # rank 2, step 0: no stereo match -> rank 3's frame of step 0 meets the NaN threshold
    frames[15, 1] = 128  # rank 7, step 1 -> rank 0's frame of step 2 meets it (across the step boundary)
    _, local, _, _ = _run_rank(frames, w, h, nf, world * per, window, steps, 0, 1)
    want_f, want_m = vd.assemble_outputs(local, 1, world * per, window)
    assert sorted(want_f) == list(range(n))

    shared = vd.ThreadWorld(world)
    shared.barrier = threading.Barrier(world, timeout=300)  # (a rank that dies must not leave seven waiting for ever)
    results, errors = [None] * world, []

    def rank_main(r):
        try:
            results[r] = _run_rank(frames, w, h, nf, per, window, steps, r, world, comm=vd.ThreadComm(shared, r), tune=True)
        except BaseException as e:  # noqa: BLE001 -- reported below, on the main thread
            errors.append((r, repr(e)))
            shared.barrier.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    tunes = [results[r][2] for r in range(world)]
    assert len({t["fast_resident"] for t in tunes}) == 1 and all(t["agreed_over_ranks"] == world for t in tunes)
    for r in range(1, world):
        assert not results[r][3], "only rank 0 receives payloads"
    completed = results[0][3]
    assert [c[0] for c in completed] == list(range(steps)) and all(len(per_rank) == world for _, per_rank in completed)
    got_f, got_m = vd.assemble_outputs(completed, world, per, window)
    assert sorted(got_f) == list(range(n)) and sorted(got_m) == sorted(want_m)
    for g in range(n):
        assert got_f[g].tobytes() == want_f[g].tobytes(), "VisionFeature records of global frame %d" % g
    for key in want_m:
        assert got_m[key].tobytes() == want_m[key].tobytes(), "FeatureMatch records of factor %s" % (key,)
    sizes = [len(want_f[g]) for g in range(n)]
    assert sizes[2] == 0 and sizes[3] == 0 and sizes[4] > 20, "empty frame, NaN-threshold frame on the next rank, then normal"
    assert sizes[15] == 0 and sizes[16] == 0 and sizes[17] > 20, "the same across the step boundary (rank 7 -> rank 0)"
    assert all(k[1] - k[0] == 1 for k in want_m) and len(want_m) == n - 1, "every temporal pair crosses a rank boundary"
    assert sum(len(v) for v in want_m.values()) > 200


def test_bench_bare_form_two_ranks_on_one_gpu():
    """`python3 bench.py --gpus 2 ...` started EXACTLY like that (no torch.distributed.run, no WORLD_SIZE in the
    environment): the parent launches its two ranks as fresh children before it touches torch or the GPU, relays rank 0's
    one JSON line and returns 0.  On this one-GPU box the ranks share device 0 and talk over gloo (VSF_BENCH_ONE_GPU=1);
    the launch path is the one an 8-GPU node takes.  The sustained leg runs with a small count."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["VSF_BENCH_ONE_GPU"] = "1"
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--batch", "8", "--steps", "3", "--warmup", "1",
                        "--width", "320", "--height", "240", "--nfeatures", "600", "--sustained-steps", "600"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl"]["ranks_seen"] == [0, 1] and out["rccl"]["world"] == 2
    assert out["config"]["global_frames_per_step"] == 16 and out["value"] > 0
    sus = out["sustained"]
    assert sus["steps"] >= 600 and sus["value"] > 0 and sus["first_100_ms"] > 0 and sus["last_100_ms"] > 0
    assert out["config"]["capacity_overflow"] is False


def test_tune_steps_leave_no_trace():
    """ShardedStereoFrontend.tune(steps > 0) runs real steps on the object (bench.py's set-up); afterwards the threshold
    chain must start at 10000 again (slam_frontend.cc:353), the first step must have no temporal predecessors and rank 0's
    bookkeeping must be empty: the payloads of the run that follows are byte-identical to a run without tune.  A batch of
    32 images makes the tuning eligible (fewer: the library reports 0 / 0 and runs no steps)."""
    from vision_slam_frontend_amd import capi, synth
    from vision_slam_frontend_amd import distributed as vd
    w, h, nf, per, window, steps = 320, 240, 600, 16, 2, 3
    frames = synth.stereo_stream(per * steps, w, h, n_objects=300)
    frames[5, 1] = 128  # a frame without stereo matches: the NaN threshold must land where it does without tune
    dev = torch.device("cuda", 0)

    def run(tune_steps):
        ctx = capi.Context(capi.default_params(w, h, max_images=2 * per, nfeatures=nf))
        sf = vd.ShardedStereoFrontend(ctx, per, w, h, _calibration(), window=window, device=dev, force_collectives=False)
        tuned = None
        if tune_steps:
            other = torch.from_numpy(np.ascontiguousarray(frames[per:2 * per][::-1])).to(dev)  # NOT the run's first batch
            tuned = sf.tune([other], samples=1, steps=tune_steps)
            assert sf.step_idx == 0 and not sf.completed and not sf.inflight and sf.next_gather == 0
            assert float(sf.thr_state.item()) == vd.INITIAL_STEREO_AMBIG_CONSTRAINT
        out = []
        for s in range(steps):
            sf.step(torch.from_numpy(np.ascontiguousarray(frames[s * per:(s + 1) * per])).to(dev))
            sf.synchronize()
            pl = sf.local_payload(s).cpu().clone()
            out.append(pl[:int(pl[12:16].view(torch.int32).item())].numpy().tobytes())
        sf.drain()
        assert all(c.sync() == capi.VSF_OK for c in sf.contexts())
        sf.close()
        ctx.close()
        return out, tuned

    plain, _ = run(0)
    tuned_run, tuned = run(2)
    assert tuned["timed_on"].startswith("2 whole steps"), tuned
    assert plain == tuned_run


def test_stereo_match_on_the_tail_stream_gives_the_same_payloads():
    """ShardedStereoFrontend(match_on_tail=True): the stereo GetMatches (cc:414) runs as the first kernel of the step's tail
    (vsf_extract_batch_dev on the extraction's stream, vsf_match_batch_dev on the tail's) instead of inside
    vsf_stereo_batch_dev -- a scheduling choice for many features per frame; the payloads are byte-identical."""
    from vision_slam_frontend_amd import capi, synth
    from vision_slam_frontend_amd import distributed as vd
    w, h, nf, per, window, steps = 320, 240, 600, 4, 2, 3
    frames = synth.stereo_stream(per * steps, w, h, n_objects=300)
    frames[6, 1] = 128
    dev = torch.device("cuda", 0)

    def run(on_tail):
        ctx = capi.Context(capi.default_params(w, h, max_images=2 * per, nfeatures=nf))
        sf = vd.ShardedStereoFrontend(ctx, per, w, h, _calibration(), window=window, device=dev, match_on_tail=on_tail)
        assert sf.match_on_tail == on_tail
        out = []
        for s in range(steps):
            sf.step(torch.from_numpy(np.ascontiguousarray(frames[s * per:(s + 1) * per])).to(dev))
            sf.synchronize()
            pl = sf.local_payload(s).cpu().clone()
            out.append(pl[:int(pl[12:16].view(torch.int32).item())].numpy().tobytes())
        sf.drain()
        assert all(c.sync() == capi.VSF_OK for c in sf.contexts())
        sf.close()
        ctx.close()
        return out

    a, b = run(False), run(True)
    assert a == b and sum(len(x) for x in a) > 3000
