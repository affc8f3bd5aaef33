"""Worker of test_ddp_gpu.py: one data-parallel rank of the native engine.  Every rank sits on cuda:0
(the test box has one GPU; RCCL refuses two ranks on one device, so the rehearsal uses gloo) and feeds
the SAME batch, so that the averaged gradient equals the single-process gradient bit for bit."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(out_path, steps, with_reducer):
    from iif_amd import resnet_cifar
    from iif_amd.custom import IIFLoss
    from iif_amd.ddp import broadcast_parameters
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    torch.manual_seed(5)
    net = resnet_cifar.resnet20(num_classes=10, use_norm="None", device=dev, compute_dtype=torch.float32)
    net.train()
    reducer = None
    if with_reducer:
        broadcast_parameters(net)
        reducer = net.make_reducer(bucket_bytes=256 << 10)     # several buckets on a 0.27M-parameter net
        assert len(reducer.buckets) >= 3

    class _D(object):
        def get_cls_num_list(self):
            return [500, 300, 200, 120, 80, 50, 30, 20, 10, 5]
    crit = IIFLoss(_D(), variant="raw", reduction="mean", device=dev)
    g = torch.Generator().manual_seed(77)
    x = torch.randn(16, 3, 32, 32, generator=g).to(dev)
    y = torch.randint(0, 10, (16,), generator=g).to(dev)
    scale = reducer.grad_scale if reducer is not None else 1.0
    losses = []
    for it in range(steps):
        loss, _ = net.loss_and_backward(x, y, crit, reducer=reducer)
        net.sgd_step(0.05, 0.9, 1e-4, grad_scale=scale)
        losses.append(float(loss.item()))
    torch.cuda.synchronize()
    torch.save({"params": net.param_arena.detach().cpu(), "losses": losses}, out_path)


if __name__ == "__main__":
    out_dir, steps = sys.argv[1], int(sys.argv[2])
    dist.init_process_group("gloo")
    run(os.path.join(out_dir, "rank%d.pt" % dist.get_rank()), steps, True)
    dist.barrier()
    dist.destroy_process_group()
