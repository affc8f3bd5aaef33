"""bench.py's launch contract (the driver types ``python bench.py --gpus N``; classification/README.md:32 is the reference's
``torch.distributed.launch`` line): with N > 1 and no WORLD_SIZE the parent spawns the ranks itself as a child process —
or refuses, from the parent and before anything touches a GPU, when fewer than N GPUs are visible."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args, env=None):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None); e.pop("RANK", None); e.pop("LOCAL_RANK", None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), env=e, capture_output=True, text=True, timeout=300)


def test_parent_refuses_more_gpus_than_visible():
    n = torch.cuda.device_count() + 1 if torch.cuda.device_count() >= 1 else 2
    r = _run("--gpus", str(n))
    assert r.returncode == 2
    assert "--gpus %d needs %d GPUs, %d visible" % (n, n, torch.cuda.device_count()) in r.stderr
    assert "Traceback" not in r.stderr and r.stdout.strip() == ""


def test_mismatch_under_a_launcher_is_still_refused():
    # under torchrun (WORLD_SIZE set) a --gpus that disagrees with the world must not silently measure another size
    r = _run("--gpus", "4", env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    if torch.cuda.is_available():
        assert "WORLD_SIZE 1 under nccl" in r.stderr


@pytest.mark.gpu
def test_self_launch_two_ranks_on_one_gpu_rehearsal():
    """``python bench.py --gpus 2`` WITHOUT torchrun: the parent starts two ranks (gloo, both on GPU 0: the one-GPU
    rehearsal) and rank 0's JSON line comes back through it."""
    import json
    r = _run("--gpus", "2", "--backend", "gloo", "--device-index", "0", "--steps", "2", "--warmup", "1", "--model", "resnet32",
             "--image", "32", "--classes", "100", "--batch", "32", "--no-cpu-baseline", "--no-fp32-step")
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1
    out = json.loads(line[0])
    assert out["n_gpus"] == 2 and out["reducer"]["world"] == 2 and out["config"]["global_batch"] == 64
