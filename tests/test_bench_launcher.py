"""`python3 bench.py --gpus N` with no launcher around it: the parent starts its N ranks as fresh child processes, relays
rank 0's one JSON line, returns the worst exit code and kills a hung world (bench.py: launch_ranks).  No GPU here: the
ranks are a stub program (VSF_BENCH_CHILD_CMD), which proves the spawn / relay / exit-code / watchdog plumbing; the real
thing is tests/test_gpu_sharded.py::test_bench_bare_form_two_ranks_on_one_gpu."""
import json
import os
import subprocess
import sys
import textwrap
import time
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent

STUB = textwrap.dedent('''
    import json, os, sys, time
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    mode = os.environ.get("STUB_MODE", "ok")
    def stage(name):
        with open(os.environ["VSF_BENCH_STAGE_FILE"], "w") as f:
            f.write("%.3f %s\\n" % (time.time(), name))
    stage("import torch")
    assert "torch" not in sys.modules
    if mode == "needs_gloo" and os.environ.get("VSF_BENCH_BACKEND", "nccl") != "gloo":
        sys.stderr.write("stub rank %d: pretend ncclCommInitRank failed\\n" % rank)
        sys.exit(7)
    if mode == "fail_rank1" and rank == 1:
        sys.stderr.write("stub rank 1: failing\\n")
        sys.exit(5)
    if mode == "fail_rank1" and rank == 0:
        stage("init_process_group")
        time.sleep(600)           # waits for the dead rank in a rendezvous, as a real world would
    if mode == "hang":
        stage("init_process_group")
        time.sleep(600)
    stage("timed steps")
    if rank == 0:
        print("[Gloo] Rank 0 is connected to 1 peer ranks.")  # what gloo really prints on stdout
        print(json.dumps({"metric": "stub", "n_gpus": world, "argv": sys.argv[1:],
                          "env": {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                                 "HSA_ENABLE_IPC_MODE_LEGACY", "VSF_BENCH_BACKEND",
                                                                 "VSF_BENCH_FALLBACK_REASON")}}), flush=True)
    else:
        sys.stderr.write("stub rank %d of %d done\\n" % (rank, world))
    sys.exit(0)
''')


def run_parent(tmp_path, mode, *extra, gpus=2, one_gpu=True, timeout=120):
    stub = tmp_path / "stub_rank.py"
    stub.write_text(STUB)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                            "VSF_BENCH_BACKEND")}
    env.update(STUB_MODE=mode, VSF_BENCH_CHILD_CMD=json.dumps([sys.executable, str(stub)]))
    if one_gpu:
        env["VSF_BENCH_ONE_GPU"] = "1"
    else:
        env.pop("VSF_BENCH_ONE_GPU", None)
    t0 = time.time()
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", str(gpus), "--steps", "3", *extra], env=env,
                       capture_output=True, text=True, timeout=timeout)
    return r, time.time() - t0


def test_parent_spawns_ranks_relays_one_json_line_and_returns_zero(tmp_path):
    r, _ = run_parent(tmp_path, "ok", "--batch", "8", gpus=3)
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 3 and out["argv"] == ["--gpus", "3", "--steps", "3", "--batch", "8"]
    e = out["env"]
    assert (e["RANK"], e["LOCAL_RANK"], e["WORLD_SIZE"], e["MASTER_ADDR"]) == ("0", "0", "3", "127.0.0.1")
    assert int(e["MASTER_PORT"]) > 0 and e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert "stub rank 1 of 3 done" in r.stderr and "stub rank 2 of 3 done" in r.stderr
    assert "[Gloo] Rank 0" in r.stderr  # a library's chatter on rank 0's stdout is diagnostics, not the result


def test_parent_returns_the_failing_ranks_code_and_takes_the_others_down(tmp_path):
    r, dt = run_parent(tmp_path, "fail_rank1")
    assert r.returncode == 5, (r.returncode, r.stderr)
    assert r.stdout.strip() == ""
    assert "rank 0: init_process_group" in r.stderr and "rank(s) [1] exited non-zero" in r.stderr
    assert dt < 60


def test_watchdog_kills_a_hung_world_and_names_the_stage(tmp_path):
    r, dt = run_parent(tmp_path, "hang", "--launch-stall", "3")
    assert r.returncode == 124, (r.returncode, r.stderr)
    assert "no rank reported a new stage for 3 s" in r.stderr
    assert r.stderr.count("init_process_group") >= 2  # both ranks' last stage
    assert dt < 40
    r, dt = run_parent(tmp_path, "hang", "--launch-deadline", "2")
    assert r.returncode == 124 and "no end after 2 s" in r.stderr


def test_failed_rccl_set_up_is_retried_once_on_gloo_and_labelled(tmp_path):
    r, _ = run_parent(tmp_path, "needs_gloo", one_gpu=False)
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout.strip())
    assert out["env"]["VSF_BENCH_BACKEND"] == "gloo" and "rc 7" in out["env"]["VSF_BENCH_FALLBACK_REASON"]
    assert "ONE retry with the collectives on gloo" in r.stderr
    r, _ = run_parent(tmp_path, "needs_gloo", "--no-gloo-retry", one_gpu=False)
    assert r.returncode == 7


def test_under_a_launcher_the_world_size_must_match(tmp_path):
    """WORLD_SIZE in the environment = started by torch.distributed.run: no second launch, and a mismatch is an error."""
    pytest.importorskip("torch")
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "--gpus 2 but WORLD_SIZE=4" in r.stderr


def test_parent_never_imports_torch():
    """The launching parent must not initialise a GPU runtime: bench.py's module level imports nothing of the kind."""
    code = "import sys; sys.argv=['bench.py']; import bench; assert 'torch' not in sys.modules and 'numpy' not in sys.modules"
    subprocess.run([sys.executable, "-c", code], cwd=ROOT, check=True, timeout=60)
