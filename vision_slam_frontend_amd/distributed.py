"""Frame sharding and the RCCL exchange of the multi-GPU path.

The hot path shards by stereo frame: extract(L), extract(R) and the stereo GetMatches of frame k depend only on
frame k's two images (slam_frontend.cc:411-416), so each rank (one process per GPU) owns a block of frames and
there is NO collective on the data path.  Two small exchanges remain:

* the RemoveAmbigStereo threshold chain (slam_frontend.cc:353, 392-394): the threshold applied to frame k is
  mean epipolar residual over ALL raw stereo matches of frame k-1, plus 2.0 -- one float per frame, independent of
  frame k-1's own threshold, so an all-gather of the per-frame means lets every rank filter locally;
* the gather of the per-frame outputs (VisionFeature / FeatureMatch payloads: left keypoints, counts, matches) to
  rank 0, which assembles the SLAMProblem.

Works on any torch.distributed backend: "nccl" (= RCCL over xGMI) with device tensors in production, "gloo" with
CPU tensors in the CPU test-suite.
"""
from __future__ import annotations

import time
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist

INITIAL_STEREO_AMBIG_CONSTRAINT = 10000.0  # slam_frontend.cc:353
STEREO_AMBIG_PADDING = 2.0                 # slam_frontend.cc:392


def frame_block(step: int, frames_per_rank: int, world: int, rank: int) -> range:
    """Global frame indices rank `rank` owns in step `step` (contiguous block per rank, step-major)."""
    start = (step * world + rank) * frames_per_rank
    return range(start, start + frames_per_rank)


def owner_of(frame: int, frames_per_rank: int, world: int) -> int:
    return (frame // frames_per_rank) % world


def gather_to_root(tensors: Dict[str, torch.Tensor], dst: int = 0,
                   bufs: Optional[Dict[str, List[torch.Tensor]]] = None) -> Optional[Dict[str, List[torch.Tensor]]]:
    """Gathers each named tensor (same shape/dtype on every rank) to `dst`.  Returns {name: [per-rank tensor]} on
    dst (rank order == frame-block order), None elsewhere.  `bufs` lets the caller reuse receive buffers."""
    world, rank = dist.get_world_size(), dist.get_rank()
    out = None
    if rank == dst:
        out = bufs if bufs is not None else {k: [torch.empty_like(v) for _ in range(world)] for k, v in tensors.items()}
    for k in sorted(tensors):
        dist.gather(tensors[k].contiguous(), out[k] if rank == dst else None, dst=dst)
    return out


def pack(tensors: Dict[str, torch.Tensor]) -> torch.Tensor:
    """One contiguous uint8 payload of the named tensors (sorted by name): a step's outputs travel in ONE collective
    instead of one per tensor (at ~6 MB per rank and step the gather is launch-latency bound, not link bound)."""
    return torch.cat([tensors[k].contiguous().view(torch.uint8).reshape(-1) for k in sorted(tensors)])


def unpack(payload: torch.Tensor, like: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Inverse of pack(): views into `payload` with the shapes / dtypes of `like`."""
    out, off = {}, 0
    for k in sorted(like):
        n = like[k].numel() * like[k].element_size()
        out[k] = payload[off:off + n].view(like[k].dtype).reshape(like[k].shape)
        off += n
    return out


def gather_packed_to_root(tensors: Dict[str, torch.Tensor], dst: int = 0,
                          bufs: Optional[List[torch.Tensor]] = None) -> Optional[List[Dict[str, torch.Tensor]]]:
    """gather_to_root with a single collective.  Returns [per-rank {name: tensor}] on dst, None elsewhere.  `bufs`:
    per-rank uint8 receive buffers of the payload size (reused across steps)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    payload = pack(tensors)
    if payload.is_cuda and dist.get_backend() == "gloo":  # rehearsal on CPU collectives: gloo gathers host tensors
        payload, bufs = payload.cpu(), None
    recv = None
    if rank == dst:
        recv = bufs if bufs is not None else [torch.empty_like(payload) for _ in range(world)]
    dist.gather(payload, recv, dst=dst)
    return None if rank != dst else [unpack(b, tensors) for b in recv]


def gather_packed_to_root_async(tensors: Dict[str, torch.Tensor], dst: int = 0,
                                bufs: Optional[List[torch.Tensor]] = None):
    """gather_packed_to_root as an asynchronous collective: returns (work, payload, recv).  The collective runs on the
    backend's own stream behind the producer of `tensors`; the caller keeps computing and calls work.wait() before it
    reuses `bufs` (recv is None off the root).  A step's outputs then travel while the next step is computed."""
    world, rank = dist.get_world_size(), dist.get_rank()
    payload = pack(tensors)
    if payload.is_cuda and dist.get_backend() == "gloo":  # rehearsal on CPU collectives: gloo gathers host tensors
        payload, bufs = payload.cpu(), None
    recv = None
    if rank == dst:
        recv = bufs if bufs is not None else [torch.empty_like(payload) for _ in range(world)]
    work = dist.gather(payload, recv, dst=dst, async_op=True)
    return work, payload, recv


def allgather_frame_means(local_means: torch.Tensor) -> torch.Tensor:
    """All ranks receive every rank's per-frame mean residuals, shape (world, frames_per_rank), rank-major."""
    world = dist.get_world_size()
    parts = [torch.empty_like(local_means) for _ in range(world)]
    dist.all_gather(parts, local_means.contiguous())
    return torch.stack(parts)


def stereo_thresholds(frame_means: Sequence[float], first: float = INITIAL_STEREO_AMBIG_CONSTRAINT) -> np.ndarray:
    """Threshold applied to each frame of a time-ordered sequence given every frame's mean residual:
    thr[0] = `first` (the static's value before the sequence), thr[k] = float32(mean[k-1] + 2.0f).
    A frame without stereo matches has mean NaN (0/0 in the reference, quirk Q3): the frame after it is filtered against
    NaN (keeps nothing), the one after that against a finite threshold again -- as the reference's static behaves."""
    m = np.asarray(frame_means, np.float32)
    thr = np.empty(len(m), np.float32)
    if len(m):
        thr[0] = np.float32(first)
        thr[1:] = m[:-1] + np.float32(STEREO_AMBIG_PADDING)
    return thr


def time_ordered(per_rank: torch.Tensor) -> torch.Tensor:
    """(world, frames_per_rank, ...) of ONE step -> (world * frames_per_rank, ...) in global frame order."""
    return per_rank.reshape((-1,) + tuple(per_rank.shape[2:]))


# ---------------------------------------------------------------------------------------------------------------------
# The sharded hot path (BASELINE configs[3]; slam_frontend.cc:400-443 per frame)
# ---------------------------------------------------------------------------------------------------------------------

def tail_region(frames_per_rank: int, window: int, world: int, parity: int) -> int:
    """First set index of the gathered-tail region of step parity `parity` in ShardedStereoFrontend's set array:
    sets 0 .. 2B-1 are the local filtered frames (2f = left, 2f+1 = right), then two regions of world x window tail
    sets (rank-major; a rank's tail = its last `window` filtered LEFT frames in time order), then one empty set."""
    return 2 * frames_per_rank + parity * world * window


def temporal_pair_sets(frames_per_rank: int, window: int, world: int, rank: int, parity: int, first_step: bool):
    """(query sets, train sets) of one rank's temporal GetFeatureMatches calls of one step (slam_frontend.cc:424-434):
    pair i*window + k matches local frame i (train = set 2i) against the frame `window - k` before it, oldest first as
    frame_list_ is walked.  A predecessor inside the rank's block is a local set; one before the block lives in the
    previous rank's tail of the same step (region `parity`) -- for rank 0 in the LAST rank's tail of the previous step
    (region 1 - parity), or nowhere when the stream starts (the empty set)."""
    B, W = frames_per_rank, window
    empty = 2 * B + 2 * world * W
    q, t = [], []
    for i in range(B):
        for w in range(W, 0, -1):
            past = i - w
            if past >= 0:
                qs = 2 * past
            elif rank > 0:
                qs = tail_region(B, W, world, parity) + (rank - 1) * W + (W + past)
            elif first_step:
                qs = empty
            else:
                qs = tail_region(B, W, world, 1 - parity) + (world - 1) * W + (W + past)
            q.append(qs)
            t.append(2 * i)
    return q, t


def collective_handshake(device=None) -> dict:
    """What a multi-GPU run needs in order to verify itself: the backend, the world size the process group reports, the
    collective library's version and -- by an all-gather of the rank ids on the backend that will carry the step's
    collectives (device tensors on RCCL) -- the ranks that actually took part.  Call once, after init_process_group."""
    if not (dist.is_available() and dist.is_initialized()):
        return {"backend": "none", "world": 1, "nccl_version": None, "ranks_seen": [0]}
    backend, world, rank = dist.get_backend(), dist.get_world_size(), dist.get_rank()
    on_device = backend == "nccl"
    dev = device if (on_device and device is not None) else (torch.device("cuda", torch.cuda.current_device()) if on_device else "cpu")
    mine = torch.tensor([rank], dtype=torch.int32, device=dev)
    seen = torch.full((world,), -1, dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(seen, mine) if on_device else dist.all_gather(list(seen.split(1)), mine)
    version = None
    if on_device:
        try:
            version = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:  # noqa: BLE001 -- a version string is a nicety, never a reason to fail the run
            version = "unknown"
    return {"backend": backend + (" (RCCL)" if on_device and getattr(torch.version, "hip", None) else ""), "world": world,
            "nccl_version": version, "ranks_seen": [int(v) for v in seen.cpu().tolist()]}


class _Done:
    """A finished collective (interface of torch's Work object)."""

    def wait(self):
        return True


class TorchComm:
    """The step's exchanges on torch.distributed: "nccl" (= RCCL over xGMI, device tensors, stream-ordered) in production,
    "gloo" (tensors take a detour through the host) for one-GPU rehearsals and the CPU-side tests."""

    def __init__(self):
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self.host_detour = dist.get_backend() == "gloo"
        self.name = "torch.distributed/" + dist.get_backend()

    def all_gather(self, out: torch.Tensor, inp: torch.Tensor, stream):
        if self.host_detour:
            stream.synchronize()
            h_in = inp.contiguous().cpu().reshape(-1)
            parts = [torch.empty_like(h_in) for _ in range(self.world)]
            dist.all_gather(parts, h_in)
            out.view(-1).copy_(torch.cat(parts).to(out.device))
        else:
            dist.all_gather_into_tensor(out.view(-1), inp.contiguous().reshape(-1))

    def gather_async(self, send: torch.Tensor, recv, dst: int = 0):
        """recv: per-rank receive tensors on dst, None elsewhere.  Returns (work, the tensor that must stay alive)."""
        if self.host_detour:
            send = send.cpu()
        return dist.gather(send, recv, dst=dst, async_op=True), send

    def all_reduce_max(self, values, device):
        t = torch.tensor(list(values), dtype=torch.float64, device="cpu" if self.host_detour else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(v) for v in t.cpu()]


class CapiComm:
    """The same exchanges through the C ABI (include/vsf.h: vsf_comm_create / vsf_allgather_dev / vsf_gather_payload_dev),
    i.e. on librccl directly, stream-ordered on the stream of the context that issues them -- the route a C++ host (the
    reference's driver, slam_frontend_main.cc:251, 132) takes; tools/time_sharded.cc is that host.  `comm_id`: the 128 bytes
    of vsf_comm_unique_id() of ONE rank, carried to the others by the caller (here: torch.distributed's broadcast)."""

    def __init__(self, ctx, comm_id: bytes, rank: int, world: int):
        import ctypes as C

        from . import capi
        self._capi, self._C = capi, C
        self.world, self.rank, self.host_detour = int(world), int(rank), False
        self.ctx = ctx
        h = C.c_void_p()
        buf = (C.c_uint8 * 128).from_buffer_copy(comm_id)
        st = capi.lib().vsf_comm_create(ctx._h, buf, self.rank, self.world, C.byref(h))
        if st != capi.VSF_OK:
            raise capi.VsfError(st, "vsf_comm_create")
        self._h = h
        v = C.c_int()
        capi.lib().vsf_comm_info(self._h, None, None, C.byref(v))
        self.rccl_version = v.value
        self.name = "C ABI (vsf_allgather_dev / vsf_gather_payload_dev on librccl %d)" % v.value

    @staticmethod
    def unique_id() -> bytes:
        import ctypes as C

        from . import capi
        buf = (C.c_uint8 * 128)()
        st = capi.lib().vsf_comm_unique_id(buf)
        if st != capi.VSF_OK:
            raise capi.VsfError(st, "vsf_comm_unique_id")
        return bytes(buf)

    def bind(self, ctx):
        """The context whose stream orders the exchanges (ShardedStereoFrontend binds its tail context)."""
        self.ctx = ctx

    def close(self):
        if self._h:
            self._capi.lib().vsf_comm_destroy(self._h)
            self._h = None

    def _check(self, st, where):
        if st != self._capi.VSF_OK:
            raise self._capi.VsfError(st, where, self._capi.lib().vsf_last_hip_error(self.ctx._h))

    def all_gather(self, out: torch.Tensor, inp: torch.Tensor, stream):
        inp = inp.contiguous()
        self._check(self._capi.lib().vsf_allgather_dev(self.ctx._h, self._h, inp.data_ptr(), out.data_ptr(),
                                                       inp.numel() * inp.element_size()), "vsf_allgather_dev")
        self._keep = inp  # (alive until the next exchange has been queued behind it on the same stream)

    def gather_async(self, send: torch.Tensor, recv, dst: int = 0):
        nbytes = send.numel() * send.element_size()
        base, stride = 0, nbytes
        if self.rank == dst:
            base = recv[0].data_ptr()
            stride = recv[1].data_ptr() - base if self.world > 1 else nbytes
            assert all(r.data_ptr() == base + i * stride for i, r in enumerate(recv)), "receive views must be equally spaced"
        self._check(self._capi.lib().vsf_gather_payload_dev(self.ctx._h, self._h, send.data_ptr(), nbytes, base, stride, dst),
                    "vsf_gather_payload_dev")
        return _Done(), send

    def all_reduce_max(self, values, device):
        mine = torch.tensor(list(values), dtype=torch.float64, device=device)
        every = torch.zeros(self.world * len(mine), dtype=torch.float64, device=device)
        torch.cuda.current_stream(device).synchronize()
        self.all_gather(every, mine, None)
        self.ctx.sync()
        return [float(v) for v in every.view(self.world, -1).max(0).values.cpu()]

    def ranks_seen(self, device):
        mine = torch.tensor([self.rank], dtype=torch.int32, device=device)
        every = torch.full((self.world,), -1, dtype=torch.int32, device=device)
        torch.cuda.current_stream(device).synchronize()
        self.all_gather(every, mine, None)
        self.ctx.sync()
        return [int(v) for v in every.cpu()]


class ThreadWorld:
    """Shared state of a world whose ranks are THREADS of one process (ThreadComm): a barrier and one slot per rank."""

    def __init__(self, world: int):
        import threading
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world


class ThreadComm:
    """The same exchanges between ranks that are threads of ONE process on ONE GPU -- a rehearsal vehicle: a GPU box admits
    only a handful of processes on its card, so a world of eight cannot be eight processes there, but the sharding logic
    (frame blocks, threshold chain, tail exchange, sized gather) does not care what carries the bytes.  Every exchange is
    a rendezvous through host memory: publish, barrier, read, barrier."""

    def __init__(self, shared: ThreadWorld, rank: int):
        self.shared, self.world, self.rank = shared, shared.world, rank
        self.host_detour = True
        self.name = "threads of one process (rehearsal)"

    def _exchange(self, mine):
        sh = self.shared
        sh.slots[self.rank] = mine
        sh.barrier.wait()
        every = list(sh.slots)
        sh.barrier.wait()
        return every

    def all_gather(self, out: torch.Tensor, inp: torch.Tensor, stream):
        stream.synchronize()
        parts = self._exchange(inp.contiguous().cpu().reshape(-1))
        out.view(-1).copy_(torch.cat(parts).to(out.device))

    def gather_async(self, send: torch.Tensor, recv, dst: int = 0):
        parts = self._exchange(send.cpu())
        if self.rank == dst:
            for r, part in zip(recv, parts):
                r.copy_(part)
        return _Done(), send

    def all_reduce_max(self, values, device):
        every = self._exchange(list(values))
        return [max(col) for col in zip(*every)]


class ShardedStereoFrontend:
    """One rank's share of a time-ordered stereo stream, one step = `frames_per_rank` frames on this GPU.

    Global frame order is step-major, rank-major (frame_block()).  Per step, on the device and stream-ordered:

      1. vsf_stereo_batch_dev                extract(L), extract(R), GetMatches(L, R)          (cc:411-416)   local
      2. vsf_stereo_residuals_batch_dev      |l^T F r| and the per-frame mean                  (cc:369-383)   local
      3. all-gather of the means             one float per frame                                              RCCL
      4. vsf_stereo_thresholds_dev           thr[g] = mean[g-1] + 2 over the step's world x B frames (cc:392-394)
      5. vsf_stereo_filter_batch_dev         RemoveAmbigStereo's re-indexing                   (cc:384-397)   local
      6. all-gather of every rank's last `window` filtered left frames (descriptors + counts): the temporal
         predecessors of the next rank's first frames (rank 0 uses the last rank's tail of the PREVIOUS step)      RCCL
      7. vsf_feature_matches_batch_dev       GetFeatureMatches(past, current), `window` per frame (cc:424-434) local
      8. vsf_vision_features_batch_dev       Calculate3DPoints + UndistortFeaturePoints         (cc:437-443)   local
      9. vsf_pack_outputs_dev                compact VisionFeature / FeatureMatch payload, counts first
     10. all-gather of the payload sizes (4 bytes per rank), then -- one step later, when the sizes have reached the
         host without stalling it -- an asynchronous gather of the payloads to rank 0, sized by the counts        RCCL

    Steps 2-10 (the "tail": short latency-bound kernels and every collective) of step s run on a SECOND, high-priority
    stream with a small context of their own while step s + 1's extraction fills the chip on the main stream: the raw
    outputs of step 1 are double-buffered and two events per buffer order the streams (raw ready -> tail may read; tail
    done -> the extraction after next may overwrite).  The tail's kernels leave the vector ALUs idle most of the time, so
    this hides them -- and the collectives' rendezvous between ranks -- behind the VALU-bound extraction.
    With world == 1 the same kernels run and the collectives degenerate to local copies, so bench.py measures the same
    per-GPU work at every N.  Works on "nccl" (RCCL, device tensors) and, for one-GPU rehearsals and the CPU-side
    tests of the exchange logic, on "gloo" (tensors take a detour through the host).
    """

    PAYLOAD_SLOTS = 3

    def __init__(self, ctx, frames_per_rank: int, width: int, height: int, calib, *, window: int = 1,
                 best_percent: float = 0.3, device=None, stream=None, overlap: bool = True,
                 force_collectives: bool = False, comm=None, match_on_tail: bool = False):
        from . import capi  # (ctx is a capi.Context)

        self.ctx, self.B, self.W = ctx, int(frames_per_rank), int(window)
        self.width, self.height = int(width), int(height)
        self.calib = calib
        self.best_percent = float(np.float32(best_percent))
        # What carries the exchanges: `comm` (TorchComm, ThreadComm, or the C-ABI route CapiComm) -- by default
        # torch.distributed when a process group of more than one rank exists (force_collectives runs every collective even
        # in a world of one: the RCCL code path on a one-GPU box).
        if comm is None and dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force_collectives):
            comm = TorchComm()
        self.comm = comm
        self.dist_on = comm is not None
        self.world = comm.world if comm else 1
        self.rank = comm.rank if comm else 0
        self.host_detour = bool(comm and comm.host_detour)
        if not 0 <= self.W <= self.B:
            raise ValueError("window must be in [0, frames_per_rank]")
        dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.dev = dev
        self.stream = stream if stream is not None else torch.cuda.Stream(device=dev)
        self.overlap = bool(overlap)
        # the stereo GetMatches (cc:414) as the first kernel of the tail instead of the last of the extraction: its matrix-core
        # work then runs beside the next step's pyramid and FAST (a choice for many features per frame; needs the tail stream)
        self.match_on_tail = bool(match_on_tail) and self.overlap
        if self.overlap:
            # the tail's own stream (high priority: its few small workgroups should not queue behind the thousands of
            # the extraction kernels) and context (scratch buffers and stream are per context)
            self.tail_stream = torch.cuda.Stream(device=dev, priority=-1)
            tail_params = capi.VsfParams.from_buffer_copy(ctx.params)
            tail_params.max_images = 2
            self.tail_ctx = capi.Context(tail_params, device=ctx.device)
            assert self.tail_ctx.params.max_keypoints == ctx.params.max_keypoints
        else:
            self.tail_stream, self.tail_ctx = self.stream, ctx
        self.NRAW = 2 if self.overlap else 1
        B, W, world, K = self.B, self.W, self.world, ctx.params.max_keypoints
        self.K = K
        self.F = calib.get("fundamental")
        u8 = torch.uint8
        with torch.cuda.stream(self.stream):
            z = lambda *shape, dtype=u8: torch.zeros(shape, dtype=dtype, device=dev)  # noqa: E731
            self.raw = [dict(kp=z(2 * B, K, 28), desc=z(2 * B, K, 32), counts=z(2 * B, dtype=torch.int32),
                             matches=z(B, K, 16), nmatches=z(B, dtype=torch.int32)) for _ in range(self.NRAW)]
            self.means = z(B, dtype=torch.float32)
            self.means_all = z(world * B, dtype=torch.float32)
            self.thr_all = z(world * B, dtype=torch.float32)
            self.thr_state = torch.full((1,), INITIAL_STEREO_AMBIG_CONSTRAINT, dtype=torch.float32, device=dev)
            # filtered frames: sets 0 .. 2B-1 (2f = left, 2f+1 = right of local frame f), then two regions (step
            # parity) of world x W gathered tail sets, then one set that stays empty
            self.tail0 = 2 * B
            self.empty_set = 2 * B + 2 * world * W
            nsets = self.empty_set + 1
            self.kpf = z(2 * B, K, 28)
            self.descf, self.countsf = z(nsets, K, 32), z(nsets, dtype=torch.int32)
            self.feat, self.nfeat = z(B, K, 28), z(B, dtype=torch.int32)
            self.NP = B * W
            self.pairs = z(max(self.NP, 1), K, 2, dtype=torch.int64)
            self.npairs = z(max(self.NP, 1), dtype=torch.int32)
            self.cap = ctx.packed_outputs_capacity(B, self.NP)
            self.payload = [z(self.cap) for _ in range(self.PAYLOAD_SLOTS)]
            self.sizes_dev = [z(world, dtype=torch.int32) for _ in range(self.PAYLOAD_SLOTS)]
            self.tail_idx = torch.tensor([2 * (B - W + j) for j in range(W)], dtype=torch.long, device=dev)
            # (query set, train set) of temporal pair i*W + (W - w): frame i against the frame `w` before it
            self.pair_sets = {key: self._pair_sets(*key) for key in ((0, True), (1, True), (0, False), (1, False))}
        self.sizes_host = [torch.zeros(world, dtype=torch.int32).pin_memory() for _ in range(self.PAYLOAD_SLOTS)]
        self.size_events = [torch.cuda.Event() for _ in range(self.PAYLOAD_SLOTS)]
        self.raw_ready = [torch.cuda.Event() for _ in range(self.NRAW)]  # extraction of the buffer's step has finished
        self.raw_free = [torch.cuda.Event() for _ in range(self.NRAW)]   # the tail has read it: may be overwritten
        self.recv = None
        if self.rank == 0 and self.dist_on:
            # receive buffers sized by the largest possible payload (one allocation per slot, a row per rank: the C-ABI
            # gather takes a base and a stride); each gather uses a prefix sized by the counts
            rdev = "cpu" if self.host_detour else dev
            self.recv = [list(torch.empty((world, self.cap), dtype=u8, device=rdev).unbind(0))
                         for _ in range(self.PAYLOAD_SLOTS)]
        self.step_idx = 0
        self.next_gather = 0  # first step whose payload has not been handed to a gather yet
        self.inflight = []   # (step, work, send tensor, receive views) of the gathers not yet waited for
        self.completed = []  # on rank 0: (step, [uint8 tensor per rank]) in step order
        self.keep_outputs = True
        self.blocked_s = 0.0  # host wall time spent waiting inside _issue_gather / _retire_through (bench.py reports it)
        ctx.set_stream(self.stream.cuda_stream)
        if self.overlap:
            self.tail_ctx.set_stream(self.tail_stream.cuda_stream)
        # the tail's scratch for B frames and B x W pairs, sized now (set-up) instead of inside the first step
        self.tail_ctx.reserve(B, max(self.NP, 1))
        if comm is not None and hasattr(comm, "bind"):
            comm.bind(self.tail_ctx)  # the exchanges are ordered on the tail's stream
        self.stream.synchronize()

    # the raw outputs of the most recent step (bench.py reports their counts)
    @property
    def counts(self):
        return self.raw[(self.step_idx - 1) % self.NRAW]["counts"]

    @property
    def nmatches(self):
        return self.raw[(self.step_idx - 1) % self.NRAW]["nmatches"]

    def synchronize(self):
        """Waits for everything issued so far on both streams (not for outstanding gathers: drain())."""
        self.stream.synchronize()
        self.tail_stream.synchronize()

    def contexts(self):
        """The vsf contexts whose per-stage timers together cover a step."""
        return [self.ctx] + ([self.tail_ctx] if self.overlap else [])

    def close(self):
        if self.overlap and self.tail_ctx is not None:
            self.tail_stream.synchronize()
            self.tail_ctx.close()
            self.tail_ctx = None

    # ---- the one measured launch choice, made the same on every rank ----
    def tune(self, d_img, samples: int = 3, steps: int = 0, step_fn=None, warm: int = 3) -> dict:
        """Explicit and blocking (set-up, never inside a timed region): vsf_tune_fast_resident times the two forms of the
        FAST launch on this rank's own batch (median of `samples` runs each), then ONE all-reduce (max over ranks) of the
        two medians makes the choice common: the step time of the job is its slowest rank's, and ranks that ran different
        forms would hand each other a persistent per-step skew through the means all-gather.  Every rank issues exactly
        the same collectives whatever its own measurement returned (no rank-dependent control flow).
        steps > 0 (bench.py): the two forms are timed on whole STEPS of this class instead -- `warm` untimed steps (three: with
        one, the form measured first paid for the pipeline filling -- 6.53 against 6.28 ms where both take 6.35 -- and at
        10 000 features six timed steps could not tell 11.6 from 12.0 ms), then `steps` timed ones per form, on `d_img` (a batch, or a list of batches taken in turn as the caller's own loop will) -- because
        what the form is worth shows in the composed, pipelined step (the blur beside FAST, the next step's pyramid beside
        this step's tail), where it is twice what an extraction by itself shows; an ineligible batch (the library reports
        0 / 0: fewer than 32 images, blur in line) runs no steps.  step_fn: the caller's own way to run one step (a callable
        that ends in self.step(...), e.g. with frames arriving from a decoder) instead of self.step(batch).  The tuning steps are real step() calls; reset() afterwards
        puts the threshold chain, the step counter and the gather bookkeeping back, so the run that follows produces the
        same bytes as one without tune (tests/test_gpu_sharded.py::test_tune_steps_leave_no_trace)."""
        batches = list(d_img) if isinstance(d_img, (list, tuple)) else [d_img]
        d_img = batches[0]
        raw = self.raw[0]
        with torch.cuda.stream(self.stream):
            g, r = self.ctx.tune_fast_resident(d_img.data_ptr(), 2 * self.B, self.width * self.height, self.width,
                                               raw["kp"].data_ptr(), raw["desc"].data_ptr(), raw["counts"].data_ptr(),
                                               samples)
        extraction_ms = (g, r)
        if steps > 0 and g > 0.0 and r > 0.0:
            per_form = []
            for form in (0, 3):
                self.ctx.set_fast_resident(form)
                one = (lambda i: step_fn()) if step_fn is not None else (lambda i: self.step(batches[i % len(batches)]))
                for i in range(warm):  # (the pipeline -- and a decoder in front of the step -- take three steps to fill)
                    one(-1 - i)
                self.drain()
                t0 = time.perf_counter()
                for i in range(steps):
                    one(i)
                self.drain()
                per_form.append(1e3 * (time.perf_counter() - t0) / steps)
            g, r = per_form
            self.reset()  # the tuning steps must leave no trace: the caller's first step is step 0 of the stream
        mine = (g, r)
        if self.dist_on:
            g, r = self.comm.all_reduce_max([g, r], self.dev)
        choice = 3 if (r > 0.0 and r < g) else 0
        self.ctx.set_fast_resident(choice)
        return {"ms_grid": g, "ms_resident": r, "this_rank_ms": list(mine), "fast_resident": choice, "samples": samples,
                "timed_on": "%d whole steps per form" % steps if (steps > 0 and extraction_ms[0] > 0.0) else "the extraction by itself",
                "extraction_ms": list(extraction_ms), "agreed_over_ranks": self.world}

    def reset(self):
        """Back to the state of a fresh object (everything issued so far is drained first): RemoveAmbigStereo's static
        threshold at its initial 10000 (slam_frontend.cc:353), the next step is step 0 again (no temporal predecessors, first
        payload slot), nothing gathered, nothing in flight.  tune(steps > 0) ends with it, so set-up steps cannot shift the
        reference sequence of the run that follows.  Every rank calls it at the same point (it issues no collective)."""
        self.drain()
        with torch.cuda.stream(self.tail_stream):
            self.thr_state.fill_(INITIAL_STEREO_AMBIG_CONSTRAINT)
            self.countsf.zero_()
        self.tail_stream.synchronize()
        self.step_idx = 0
        self.next_gather = 0
        self.inflight.clear()
        self.completed.clear()
        self.blocked_s = 0.0

    # ---- static schedule of the temporal pairs ----
    def _pair_sets(self, parity: int, first_step: bool):
        q, t = temporal_pair_sets(self.B, self.W, self.world, self.rank, parity, first_step)
        mk = lambda v: torch.tensor(v if v else [0], dtype=torch.int32, device=self.dev)  # noqa: E731
        return mk(q), mk(t)

    def pair_frames(self, step: int):
        """(global past frame, global current frame) of each temporal pair of this rank's payload, -1 = none."""
        out = []
        for i in range(self.B):
            g = (step * self.world + self.rank) * self.B + i
            for w in range(self.W, 0, -1):
                out.append((g - w if g - w >= 0 else -1, g))
        return out

    # ---- collectives (host detour on gloo) ----
    def _all_gather(self, out: torch.Tensor, inp: torch.Tensor):
        """out: contiguous (world * inp.numel()) view; rank-major."""
        if not self.dist_on:
            out.view(-1).copy_(inp.reshape(-1))
            return
        self.comm.all_gather(out, inp, self.tail_stream)

    # ---- one step ----
    def step(self, d_img: torch.Tensor, input_event=None):
        """d_img: (B, 2, H, W) uint8 resident in HBM: this rank's frames of step `step_idx`, in time order.
        input_event: a recorded torch.cuda.Event (or a raw hipEvent_t handle) behind the producer of d_img -- a decoder on
        another stream, say; the extraction, its pipelined pyramid included, waits for it on the GPU (vsf_set_input_event)."""
        ctx, tctx, B, W, K, world = self.ctx, self.tail_ctx, self.B, self.W, self.K, self.world
        s = self.step_idx
        parity, slot = s & 1, s % self.PAYLOAD_SLOTS
        raw = self.raw[s % self.NRAW]
        p = lambda t: t.data_ptr()  # noqa: E731
        with torch.cuda.stream(self.stream):
            if self.overlap and s >= self.NRAW:
                self.stream.wait_event(self.raw_free[s % self.NRAW])  # the tail of step s - 2 has read this buffer
            if input_event is not None:
                ctx.set_input_event(getattr(input_event, "cuda_event", input_event))
            if self.match_on_tail:
                ctx.extract_batch_dev(p(d_img), 2 * B, self.width * self.height, self.width, p(raw["kp"]), p(raw["desc"]),
                                      p(raw["counts"]))
            else:
                ctx.stereo_batch_dev(p(d_img), B, self.width * self.height, self.width, p(raw["kp"]), p(raw["desc"]),
                                     p(raw["counts"]), p(raw["matches"]), p(raw["nmatches"]))
            if self.overlap:
                self.raw_ready[s % self.NRAW].record(self.stream)
        with torch.cuda.stream(self.tail_stream):
            if self.overlap:
                self.tail_stream.wait_event(self.raw_ready[s % self.NRAW])
            if self.match_on_tail:  # (left, right) of every frame: sets 2 f / 2 f + 1 of the raw descriptors
                tctx.match_batch_dev(p(raw["desc"]), p(raw["counts"]), K * 32, 0, 0, B, 0, 0, p(raw["matches"]), p(raw["nmatches"]))
            if self.dist_on:  # payload slot `slot` (and the root's receive set) was last used by step s - PAYLOAD_SLOTS
                self._retire_through(s - self.PAYLOAD_SLOTS)
            tctx.stereo_residuals_batch_dev(p(raw["kp"]), p(raw["matches"]), p(raw["nmatches"]), B, self.F, p(self.means))
            self._all_gather(self.means_all, self.means)
            tctx.stereo_thresholds_dev(p(self.means_all), world * B, p(self.thr_state), p(self.thr_all))
            tctx.stereo_filter_batch_dev(p(raw["kp"]), p(raw["desc"]), p(raw["matches"]), p(raw["nmatches"]), B,
                                         p(self.thr_all) + 4 * self.rank * B, p(self.kpf), p(self.descf), p(self.countsf))
            if self.overlap:
                self.raw_free[s % self.NRAW].record(self.tail_stream)  # nothing below reads the raw outputs
            if W > 0:
                r0 = self.tail0 + parity * world * W
                self._all_gather(self.descf[r0:r0 + world * W], self.descf.index_select(0, self.tail_idx))
                self._all_gather(self.countsf[r0:r0 + world * W], self.countsf.index_select(0, self.tail_idx))
                q_set, t_set = self.pair_sets[(parity, s == 0)]
                tctx.feature_matches_batch_dev(p(self.descf), p(self.countsf), K * 32, p(q_set), p(t_set), self.NP,
                                               self.best_percent, p(self.pairs), p(self.npairs))
            tctx.vision_features_batch_dev(self.calib, p(self.kpf), p(self.descf), p(self.countsf), B, p(self.feat),
                                           p(self.nfeat), 0)
            tctx.pack_outputs_dev(p(self.feat), p(self.nfeat), B, p(self.pairs), p(self.npairs), self.NP,
                                  p(self.payload[slot]), self.cap)
            if self.dist_on:
                # sizes of every rank's payload -> host, without stalling it: read one step later
                self._all_gather(self.sizes_dev[slot], self.payload[slot][12:16].view(torch.int32))
                self.sizes_host[slot].copy_(self.sizes_dev[slot], non_blocking=True)
                self.size_events[slot].record(self.tail_stream)
                while self.next_gather < s:
                    self._issue_gather(self.next_gather)
        self.step_idx += 1

    def _issue_gather(self, step: int):
        slot = step % self.PAYLOAD_SLOTS
        t0 = time.perf_counter()
        self.size_events[slot].synchronize()  # (long done: the GPU is at least one step ahead of this point)
        self.blocked_s += time.perf_counter() - t0
        nbytes = int(self.sizes_host[slot].max())
        nbytes = min((nbytes + 15) & ~15, self.cap)
        recv = [b[:nbytes] for b in self.recv[slot]] if self.rank == 0 else None
        if self.host_detour:
            self.tail_stream.synchronize()  # (the payload is about to be read by the host)
        work, send = self.comm.gather_async(self.payload[slot][:nbytes], recv, dst=0)
        self.inflight.append((step, work, send, recv))
        self.next_gather = step + 1

    def _retire_through(self, step: int):
        """Waits (stream-ordered on RCCL, on the host with gloo) for the gathers of all steps <= `step`."""
        while self.inflight and self.inflight[0][0] <= step:
            st, work, _send, recv = self.inflight.pop(0)
            t0 = time.perf_counter()
            work.wait()
            self.blocked_s += time.perf_counter() - t0
            if self.rank == 0 and self.keep_outputs:
                self.completed.append((st, [r.clone() for r in recv]))

    def drain(self):
        """Issues and completes every outstanding gather (call inside the timed region), then waits for the stream."""
        if self.dist_on:
            with torch.cuda.stream(self.tail_stream):
                while self.next_gather < self.step_idx:
                    self._issue_gather(self.next_gather)
                self._retire_through(self.step_idx)
        self.tail_stream.synchronize()
        self.stream.synchronize()

    def local_payload(self, step: int) -> torch.Tensor:
        """This rank's packed payload of `step` (valid until PAYLOAD_SLOTS further steps have been issued)."""
        return self.payload[step % self.PAYLOAD_SLOTS]


def assemble_outputs(completed, world: int, frames_per_rank: int, window: int):
    """Rank 0's view of the gathered payloads: ({global frame: VisionFeature records}, {(past frame, current frame):
    FeatureMatch records}) from ShardedStereoFrontend.completed (or [(step, [payload])] of a single-process run)."""
    from . import capi

    feats, factors = {}, {}
    for step, per_rank in completed:
        for r, payload in enumerate(per_rank):
            raw = payload.cpu().numpy() if isinstance(payload, torch.Tensor) else np.asarray(payload)
            fl, ml = capi.unpack_outputs(raw)
            assert len(fl) == frames_per_rank and len(ml) == frames_per_rank * window
            base = (step * world + r) * frames_per_rank
            for i in range(frames_per_rank):
                feats[base + i] = fl[i]
                for k, w in enumerate(range(window, 0, -1)):
                    if base + i - w >= 0:
                        factors[(base + i - w, base + i)] = ml[i * window + k]
                    else:
                        assert len(ml[i * window + k]) == 0
    return feats, factors
