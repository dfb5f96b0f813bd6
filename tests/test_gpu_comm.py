"""The multi-GPU exchange behind the C ABI (round-3 review, "what's missing" 2): vsf_comm_create / vsf_allgather_dev /
vsf_gather_payload_dev on librccl directly, stream-ordered on the context's stream (include/vsf.h; SURVEY.md 8(b)'s seam
"vsf_gather_*"; callers in the reference: slam_frontend_main.cc:251, 132).  A one-GPU box hosts one RCCL rank, so both tests
run a world of one in which EVERY exchange of the step still goes through RCCL:

  * ShardedStereoFrontend with distributed.CapiComm (the Python composition over the C-ABI route) gathers the same payload
    bytes as the collective-free run;
  * tools/time_sharded.cc -- one thread per GPU, the steps of DESIGN.md section 7 (NOTES.md section 7 lists all ten) through include/vsf.h only, no Python
    -- gathers the same payload bytes as the Python composition over the same frames."""
import os
import socket
import struct
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

ROOT = Path(__file__).resolve().parent.parent
W_IMG, H_IMG, NF = 320, 240, 600
B, STEPS, WINDOW = 6, 3, 2


def _calibration():
    from vision_slam_frontend_amd import frontend
    return frontend.default_calibration().set("fundamental", [0, 0, 0, 0, 0, -1, 0, 1, 0])


def _run(frames, comm_factory=None):
    from vision_slam_frontend_amd import capi
    from vision_slam_frontend_amd import distributed as vd
    dev = torch.device("cuda", 0)
    ctx = capi.Context(capi.default_params(W_IMG, H_IMG, max_images=2 * B, nfeatures=NF))
    comm = comm_factory(ctx) if comm_factory else None
    sf = vd.ShardedStereoFrontend(ctx, B, W_IMG, H_IMG, _calibration(), window=WINDOW, device=dev, comm=comm)
    local = []
    for s in range(STEPS):
        d_img = torch.from_numpy(np.ascontiguousarray(frames[s * B:(s + 1) * B])).to(dev)
        sf.step(d_img)
        if comm is None:
            sf.synchronize()
            p = sf.local_payload(s).cpu().numpy()
            local.append(p[:int(p[:16].view(np.uint32)[3])].copy())
    sf.drain()
    assert all(c.sync() == capi.VSF_OK for c in sf.contexts())
    if comm is not None:
        assert sf.dist_on and not sf.host_detour and [c[0] for c in sf.completed] == list(range(STEPS))
        for _, per in sf.completed:
            p = per[0].cpu().numpy()
            local.append(p[:int(p[:16].view(np.uint32)[3])].copy())
        tuned = sf.tune(torch.from_numpy(np.ascontiguousarray(frames[:B])).to(dev), samples=1)  # its one exchange too
        assert tuned["agreed_over_ranks"] == 1
        assert comm.ranks_seen(dev) == [0] and comm.rccl_version > 20000
    sf.close()
    if comm is not None:
        comm.close()
    ctx.close()
    return local


def _capi_worker(frames_path, out_path):
    sys.path.insert(0, str(ROOT))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from vision_slam_frontend_amd import distributed as vd
    torch.cuda.set_device(0)
    frames = np.load(frames_path)
    got = _run(frames, lambda ctx: vd.CapiComm(ctx, vd.CapiComm.unique_id(), 0, 1))
    np.savez(out_path, **{"s%d" % i: p for i, p in enumerate(got)})


@pytest.fixture(scope="module")
def frames():
    from vision_slam_frontend_amd import synth
    f = synth.stereo_stream(B * STEPS, W_IMG, H_IMG, n_objects=400)
    f[2, 1] = 128  # a frame without stereo matches: the next one meets the NaN threshold (quirk Q3)
    return f


@pytest.fixture(scope="module")
def reference_payloads(frames):
    want = _run(frames)
    assert len(want) == STEPS and all(len(p) > 2000 for p in want)
    return want


def test_capi_collectives_on_rccl_in_a_world_of_one(tmp_path, frames, reference_payloads):
    import torch.multiprocessing as mp
    frames_path, out_path = str(tmp_path / "frames.npy"), str(tmp_path / "got.npz")
    np.save(frames_path, frames)
    ctx = mp.get_context("spawn")  # (a fresh process: the communicator's life cycle from load to destroy)
    p = ctx.Process(target=_capi_worker, args=(frames_path, out_path))
    p.start()
    p.join(600)
    assert p.exitcode == 0
    z = np.load(out_path)
    for s, want in enumerate(reference_payloads):
        assert z["s%d" % s].tobytes() == want.tobytes(), "payload of step %d through the C-ABI route" % s


def test_cpp_sharded_driver_gathers_the_same_payloads(tmp_path, frames, reference_payloads):
    """No Python in the loop: tools/time_sharded (built by `make -C tools time_sharded`; one thread per visible GPU)."""
    exe = ROOT / "tools" / "time_sharded"
    r = subprocess.run(["make", "-s", "-C", str(ROOT / "tools"), "time_sharded"], capture_output=True, text=True)
    assert r.returncode == 0 and exe.exists(), r.stderr[-2000:]
    raw, out = tmp_path / "frames.raw", tmp_path / "payloads.bin"
    np.ascontiguousarray(frames).tofile(raw)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([str(exe), str(raw), str(W_IMG), str(H_IMG), str(len(frames)), str(NF), str(B), str(WINDOW), str(STEPS),
                        str(out), "1"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-2000:])
    import json
    info = json.loads(r.stdout.strip().splitlines()[-1])
    assert info["n_gpus"] == 1 and info["ranks_seen"] == [0] and info["rccl_version"] > 20000
    blob = out.read_bytes()
    off, got = 0, []
    while off < len(blob):
        (n,) = struct.unpack_from("<I", blob, off)
        got.append(blob[off + 4:off + 4 + n])
        off += 4 + n
    assert len(got) == STEPS
    for s, want in enumerate(reference_payloads):
        assert got[s] == want.tobytes(), "payload of step %d from the C++ driver" % s
