"""Two data-parallel ranks of the native engine on the GPU (train.py:230-234 replaced by ArenaReducer):
bucketed all-reduce launched from inside backward, weight-gradient side stream, 1/world folded into SGD."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_two_ranks_same_batch_equal_single_process_bit_for_bit(tmp_path):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(HERE, "ddp_gpu_worker.py"), str(tmp_path), "4"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    sys.path.insert(0, HERE)
    import ddp_gpu_worker
    ddp_gpu_worker.run(str(tmp_path / "single.pt"), 4, False)
    single = torch.load(tmp_path / "single.pt")
    for rk in (0, 1):
        got = torch.load(tmp_path / ("rank%d.pt" % rk))
        assert got["losses"] == single["losses"]
        assert torch.equal(got["params"], single["params"])       # (g + g) * 0.5 == g exactly
