"""N > 1 host path on CPU: two gloo ranks shard frames, exchange the threshold chain and gather outputs to rank 0
exactly as bench.py / the production path do over RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vision_slam_frontend_amd import distributed as vd

B, K = 3, 16  # frames per rank per step, keypoint capacity


def _fake_outputs(frame: int):
    rng = np.random.default_rng(1000 + frame)
    n = int(rng.integers(1, K))
    kp = np.zeros((K, 28), np.uint8)
    kp[:n] = rng.integers(0, 256, (n, 28), dtype=np.uint8)
    nm = int(rng.integers(0, n + 1))
    m = np.zeros((K, 16), np.uint8)
    m[:nm] = rng.integers(0, 256, (nm, 16), dtype=np.uint8)
    mean = np.float32(rng.uniform(0, 5)) if nm else np.float32("nan")
    return kp, n, m, nm, mean


def _worker(rank: int, world: int, port: int, results):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        all_ok = True
        hs = vd.collective_handshake()  # bench.py's self-verification: every rank sees every rank through the backend
        all_ok &= hs["world"] == world and hs["ranks_seen"] == list(range(world)) and hs["backend"] == "gloo"
        for step in range(2):
            frames = vd.frame_block(step, B, world, rank)
            outs = [_fake_outputs(f) for f in frames]
            t = {
                "kp": torch.from_numpy(np.stack([o[0] for o in outs])),
                "counts": torch.tensor([o[1] for o in outs], dtype=torch.int32),
                "matches": torch.from_numpy(np.stack([o[2] for o in outs])),
                "nmatches": torch.tensor([o[3] for o in outs], dtype=torch.int32),
            }
            means = torch.tensor([o[4] for o in outs], dtype=torch.float32)
            allm = vd.allgather_frame_means(means)
            assert allm.shape == (world, B)
            flat = vd.time_ordered(allm.unsqueeze(-1)).squeeze(-1).numpy()
            expect = np.array([_fake_outputs(f)[4] for r in range(world) for f in vd.frame_block(step, B, world, r)],
                              np.float32)
            all_ok &= bool(np.array_equal(flat, expect, equal_nan=True))
            gp = vd.gather_packed_to_root(t, dst=0)  # one collective; must equal the per-tensor gather
            work, payload, recv = vd.gather_packed_to_root_async(t, dst=0)  # the overlapped form bench.py uses
            g = vd.gather_to_root(t, dst=0)
            work.wait()
            if rank == 0:
                for r in range(world):
                    got = vd.unpack(recv[r], t)
                    for k in t:
                        all_ok &= bool(torch.equal(got[k], gp[r][k]))
            else:
                all_ok &= recv is None
            if rank == 0:
                for r in range(world):
                    for k in t:
                        all_ok &= bool(torch.equal(gp[r][k], g[k][r]))
            else:
                all_ok &= gp is None
            if rank == 0:
                for r in range(world):
                    for i, f in enumerate(vd.frame_block(step, B, world, r)):
                        kp, n, m, nm, _ = _fake_outputs(f)
                        all_ok &= bool(np.array_equal(g["kp"][r][i].numpy(), kp))
                        all_ok &= int(g["counts"][r][i]) == n and int(g["nmatches"][r][i]) == nm
                        all_ok &= bool(np.array_equal(g["matches"][r][i].numpy(), m))
                        all_ok &= vd.owner_of(f, B, world) == r
            else:
                assert g is None
        results[rank] = all_ok
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 8])
def test_gloo_gather_and_threshold_exchange(world):
    """World of two, and of EIGHT (BASELINE configs[3]'s rank count: the handshake must see ranks 0..7, the means and the
    payloads must arrive rank-major): real processes, gloo, no GPU."""
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), results), nprocs=world, join=True)
    assert dict(results) == {r: True for r in range(world)}


def test_thread_world_carries_the_same_exchanges():
    """distributed.ThreadComm (ranks = threads of one process: the vehicle of the eight-rank GPU rehearsal) against the
    definitions: all-gather is rank-major, the gather lands on the root only, the all-reduce is a maximum -- for eight
    ranks, repeated so that slot reuse is exercised."""
    import threading
    world = 8
    shared = vd.ThreadWorld(world)
    ok = [False] * world

    class _Stream:  # (ThreadComm synchronises the producer's stream before it reads: nothing to wait for on the CPU)
        def synchronize(self):
            pass

    def main(r):
        comm = vd.ThreadComm(shared, r)
        good = True
        for it in range(5):
            inp = torch.arange(3, dtype=torch.float32) + 10 * r + 100 * it
            out = torch.zeros(world * 3)
            comm.all_gather(out, inp, _Stream())
            want = torch.cat([torch.arange(3, dtype=torch.float32) + 10 * q + 100 * it for q in range(world)])
            good &= bool(torch.equal(out, want))
            send = torch.full((4,), r + it, dtype=torch.uint8)
            recv = [torch.zeros(4, dtype=torch.uint8) for _ in range(world)] if r == 0 else None
            work, _ = comm.gather_async(send, recv, dst=0)
            work.wait()
            if r == 0:
                good &= all(int(recv[q][0]) == q + it for q in range(world))
            good &= comm.all_reduce_max([float(r), float(-r)], "cpu") == [float(world - 1), 0.0]
        ok[r] = good

    threads = [threading.Thread(target=main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert all(ok)


def test_handshake_without_a_process_group():
    assert vd.collective_handshake() == {"backend": "none", "world": 1, "nccl_version": None, "ranks_seen": [0]}


def test_frame_blocks_partition_the_stream():
    world, steps = 4, 3
    seen = []
    for s in range(steps):
        for r in range(world):
            seen.extend(vd.frame_block(s, B, world, r))
    assert seen == list(range(world * steps * B))


def test_stereo_threshold_chain():
    thr = vd.stereo_thresholds([1.5, float("nan"), 0.25, 3.0])
    assert thr.dtype == np.float32
    assert thr[[0, 1, 3]].tolist() == [10000.0, 3.5, 2.25] and np.isnan(thr[2])  # NaN for exactly one frame (quirk Q3)
    assert vd.stereo_thresholds([], 7.0).tolist() == []
    # float32 arithmetic, like `avg_constraint / n + padding_from_average` in the reference
    m = np.float32(0.1)
    assert vd.stereo_thresholds([m, m])[1] == np.float32(m + np.float32(2.0))


@pytest.mark.parametrize("world,B,W", [(1, 4, 2), (2, 3, 2), (3, 2, 2), (4, 5, 1), (2, 4, 4), (8, 1, 1), (8, 2, 2)])
def test_temporal_pair_schedule_reaches_the_right_frames(world, B, W):
    """The static (query set, train set) schedule of ShardedStereoFrontend resolves, on every rank and step, to the
    global frames (g - w, g) of slam_frontend.cc:424-434 -- through local sets, the previous rank's tail of the same
    step, or (rank 0) the last rank's tail of the previous step."""
    steps = 4
    empty = 2 * B + 2 * world * W
    where = {}  # (step parity region, slot) -> global frame held there after that step's tail all-gather
    seen = set()
    for s in range(steps):
        parity = s & 1
        for r in range(world):  # what the all-gather of step s leaves in region `parity`
            blk = list(vd.frame_block(s, B, world, r))
            for j in range(W):
                where[(parity, r * W + j)] = blk[B - W + j]
        for r in range(world):
            blk = list(vd.frame_block(s, B, world, r))
            q, t = vd.temporal_pair_sets(B, W, world, r, parity, s == 0)
            assert len(q) == len(t) == B * W
            k = 0
            for i in range(B):
                for w in range(W, 0, -1):
                    g = blk[i]
                    assert t[k] == 2 * i
                    if q[k] == empty:
                        assert g - w < 0
                    elif q[k] < 2 * B:
                        assert q[k] % 2 == 0 and blk[q[k] // 2] == g - w
                    else:
                        reg = (q[k] - 2 * B) // (world * W)
                        slot = (q[k] - 2 * B) % (world * W)
                        # a same-step region must have been filled by a LOWER rank (its tail exists before ours is needed)
                        assert reg == parity and slot // W < r or reg == 1 - parity and r == 0
                        assert where[(reg, slot)] == g - w
                    if g - w >= 0:
                        seen.add((g - w, g))
                    k += 1
    n = world * B * steps
    assert seen == {(g - w, g) for g in range(n) for w in range(1, W + 1) if g - w >= 0}
