"""Body of __graft_entry__.smoke(): one small invocation of the hot path on
cuda:0 — ResNet32 forward, fused IIF loss, backward, fused SGD, all through the
native kernels — checked against the CPU oracle (same ReLU decisions replayed)."""
import torch

from oracle import iif_oracle as O
from oracle import resnet_oracle as R


def run():
    from iif_amd import resnet_cifar
    from iif_amd.custom import IIFLoss
    dev = torch.device("cuda", 0)
    counts = O.img_num_per_cls(100, 50000, "exp", 0.01)

    class DS:
        def get_cls_num_list(self):
            return counts
    sd = R.init_cifar("resnet32", 100, seed=0)
    net = resnet_cifar.resnet32(num_classes=100, use_norm="None", device=dev, compute_dtype=torch.float32)
    net.load_state_dict(sd)
    net.train()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(16, 3, 32, 32, generator=g)
    prior = torch.tensor(counts, dtype=torch.float64)
    y = torch.multinomial(prior / prior.sum(), 16, replacement=True, generator=g)
    crit = IIFLoss(DS(), variant="raw")
    table = O.iif_tables(counts)["raw"]
    bufs, ref = {}, {k: v.clone() for k, v in sd.items()}
    for it in range(2):
        lr = 0.1 * O.warmup_factor(it, 1000)
        loss, _ = net.loss_and_backward(x.to(dev), y.to(dev), crit)
        plan = net._saved
        masks = R.ReluMasks([(t > 0).permute(0, 3, 1, 2).cpu()
                             for t in [plan.stem.y] + [u.y for b in plan.blocks for u in b["units"]]])
        net.sgd_step(lr, 0.9, 1e-4)
        ref_loss, _ = R.train_step(ref, bufs, x, y, table, "resnet32", lr, relu_masks=masks)
        err = abs(loss.item() - ref_loss.item()) / abs(ref_loss.item())
        assert err < 1e-4, (it, loss.item(), ref_loss.item())
        print("smoke: step %d ResNet32+IIF loss %.6f (oracle %.6f) rel err %.1e" % (it, loss.item(), ref_loss.item(), err))
    # bf16 performance mode runs too
    netb = resnet_cifar.resnet32(num_classes=100, use_norm="None", device=dev, compute_dtype=torch.bfloat16)
    netb.load_state_dict(sd)
    netb.train()
    lb, _ = netb.loss_and_backward(x.to(dev), y.to(dev), crit)
    assert torch.isfinite(lb).item()
    print("smoke: bf16 mode loss %.4f" % lb.item())
