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
    A frame without stereo matches (mean is NaN: 0/0 in the reference, quirk Q3) leaves the threshold unchanged."""
    m = np.asarray(frame_means, np.float32)
    thr = np.empty(len(m), np.float32)
    cur = np.float32(first)
    for k in range(len(m)):
        thr[k] = cur
        if not np.isnan(m[k]):
            cur = np.float32(m[k] + np.float32(STEREO_AMBIG_PADDING))
    return thr


def time_ordered(per_rank: torch.Tensor) -> torch.Tensor:
    """(world, frames_per_rank, ...) of ONE step -> (world * frames_per_rank, ...) in global frame order."""
    return per_rank.reshape((-1,) + tuple(per_rank.shape[2:]))
