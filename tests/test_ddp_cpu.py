"""world_size-2 rehearsal of the multi-rank path on CPU (gloo): bucketed arena
all-reduce driven in backward order, parameter broadcast, metric all-reduce."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from iif_amd import resnet_cifar
        from iif_amd.ddp import ArenaReducer, broadcast_parameters
        from iif_amd.utils import SmoothedValue
        torch.manual_seed(100 + rank)                       # different init per rank
        net = resnet_cifar.resnet20(num_classes=10, device="cpu", compute_dtype=torch.float32)
        broadcast_parameters(net, src=0)
        ref = [torch.zeros_like(net.param_arena) for _ in range(world)]
        dist.all_gather(ref, net.param_arena)
        assert all(torch.equal(r, ref[0]) for r in ref)     # replicas identical after broadcast
        # views still alias the arena after the in-place broadcast
        assert net.conv1.weight.data_ptr() == net.param_arena.data_ptr()

        red = net.make_reducer(bucket_bytes=64 << 10)        # several buckets on this small net
        assert len(red.buckets) >= 3
        covered = sorted(red.buckets)
        assert covered[0][0] == 0 and covered[-1][1] == net.grad_arena.numel()
        assert all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
        g = net.grad_arena
        n = g.numel()
        base = torch.arange(n, dtype=torch.float32) % 97
        g.copy_(base * (rank + 1))
        offs = net.block_offsets()
        red.begin()
        red.gradients_ready_from(offs["head"])
        for bi in range(len(offs["blocks"]) - 1, -1, -1):     # backward order
            red.gradients_ready_from(offs["blocks"][bi])
        red.finish()
        expect = base * sum(r + 1 for r in range(world))
        assert torch.equal(g, expect)
        assert red.grad_scale == 1.0 / world

        m = SmoothedValue()
        m.update(10.0 * (rank + 1), n=4)
        m.synchronize_between_processes()
        assert m.count == 4 * world and abs(m.total - 4 * 10.0 * sum(r + 1 for r in range(world))) < 1e-9
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_arena_reduce_broadcast_and_metrics():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {0: "ok", 1: "ok"}


def test_single_process_reducer_is_a_no_op():
    from iif_amd.ddp import ArenaReducer
    g = torch.ones(1000)
    r = ArenaReducer(g, [0, 100, 500], bucket_bytes=400)
    r.begin(); r.gradients_ready_from(500); r.finish()
    assert torch.equal(g, torch.ones(1000)) and r.grad_scale == 1.0
