"""Training entry point with the command line of the reference's
``classification/train.py`` (flags :288-392, loop :39-92, evaluate :95-119,
main :168-285), running on the native MI355X engine.

    python -m iif_amd.train --model resnet32 --dset_name cifar100 --classif iif --iif raw -b 128 --epochs 2
    python -m torch.distributed.run --nproc-per-node 8 -m iif_amd.train --model resnet50 --dset_name imagenet_lt ...

Differences, all forced by the environment or by the MI355X-first design:
  * datasets are synthetic long-tailed sets (no torchvision / network);
  * the step is the fused native one — forward, fused IIF loss (+mixup), backward,
    bucketed RCCL all-reduce overlapped with backward, ONE fused SGD launch — instead of
    autograd + torch.optim + DistributedDataParallel; schedules are evaluated on the host
    and passed to the kernel, momentum buffers live in the model's momentum arena;
  * metrics are read back every ``--print-freq`` iterations (the reference syncs the host
    three times per iteration, train.py:87-92);
  * new flags: ``--compute-dtype {bf16,f32}``, ``--max-iters``.
"""
import argparse
import datetime
import math
import os
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")     # distinct hardware queues for the main / wgrad / RCCL streams

import torch   # noqa: E402

from . import custom, initialisers, resnet_cifar, resnet_pytorch, utils
from .ddp import broadcast_parameters, sync_buffers


def lr_at(args, epoch, it, iters_per_epoch):
    """Learning rate of iteration ``it`` of ``epoch``: MultiStep / cosine per epoch
    (train.py:223-228) times the linear warm-up of the first epoch (train.py:52-56)."""
    if args.cosine_scheduler:
        base = args.lr * (1 + math.cos(math.pi * epoch / args.epochs)) / 2
    else:
        base = args.lr * (args.lr_gamma ** sum(1 for m in args.milestones if epoch >= m))
    if epoch < 1:
        warm_iters = min(1000, iters_per_epoch - 1)
        base *= utils.warmup_factor(it, warm_iters, 1.0 / 1000) if warm_iters > 0 else 1.0
    return base


def scheduler_state_dict(args, last_epoch):
    """State of the reference's per-epoch scheduler (train.py:223-228) after ``last_epoch`` steps."""
    lr = lr_at(args, last_epoch, 10 ** 9, 10 ** 9)
    sd = {"last_epoch": last_epoch, "_step_count": last_epoch + 1, "base_lrs": [args.lr], "_last_lr": [lr],
          "_get_lr_called_within_step": False, "verbose": False}
    if args.cosine_scheduler:
        sd.update({"T_max": args.epochs, "eta_min": 0})
    else:
        from collections import Counter
        sd.update({"milestones": Counter(args.milestones), "gamma": args.lr_gamma})
    return sd


def train_one_epoch(model, criterion, data_loader, device, epoch, args, reducer=None):
    model.train()
    logger = utils.MetricLogger(delimiter="  ")
    logger.add_meter("lr", utils.SmoothedValue(window_size=1, fmt="{value}"))
    logger.add_meter("img/s", utils.SmoothedValue(window_size=10, fmt="{value}"))
    header = "Epoch: [{}]".format(epoch)
    n_iters = len(data_loader)
    nesterov = args.opt.lower() == "nesterov"
    scale = reducer.grad_scale if reducer is not None else 1.0
    mix = custom.Mixup(criterion, alpha=args.mixup) if args.mixup is not None else None
    it = 0
    t_last, imgs_since = time.time(), 0
    for image, target in logger.log_every(data_loader, args.print_freq, header):
        image = image.to(device, non_blocking=True)
        target = target.to(device, non_blocking=True)
        lr = lr_at(args, epoch, it, n_iters)
        probe = reducer is not None and args.bf16_buckets and not getattr(reducer, "probed", False)
        red = None if probe else reducer
        if mix is not None:
            image, ta, tb, lam = mix(image, target)
            loss, output = model.loss_and_backward(image, ta, criterion, targets_b=tb, lam=lam, reducer=red)
        else:
            loss, output = model.loss_and_backward(image, target, criterion, reducer=red)
        if probe:
            # first step: gradients are still local; measure what bf16 buckets would do to their average, switch
            # only if that stays within one bf16 rounding, then reduce this step (un-overlapped)
            worst = reducer.probe_bf16()
            reducer.probed = True
            try:
                reducer.set_bucket_dtype(torch.bfloat16)
                print("bf16 gradient buckets enabled (probe: %.2e relative L2)" % worst)
            except RuntimeError:
                print("bf16 gradient buckets REFUSED (probe: %.2e relative L2 > tolerance); staying with fp32" % worst)
            reducer.begin()
            reducer.finish()
        model.sgd_step(lr, args.momentum, args.weight_decay, nesterov, grad_scale=scale)
        imgs_since += image.shape[0]
        if it % args.print_freq == 0 or it == n_iters - 1:
            acc1, acc5 = utils.accuracy(output, target, topk=(1, min(5, output.shape[1])))
            now = time.time()                                   # .item() below is the only host sync
            logger.update(loss=loss.item(), lr=lr)
            model.check_labels()                                # out-of-range targets raise here (device assert in the reference)
            logger.meters["acc1"].update(acc1.item(), n=image.shape[0])
            logger.meters["acc5"].update(acc5.item(), n=image.shape[0])
            logger.meters["img/s"].update(imgs_since / max(now - t_last, 1e-9))
            t_last, imgs_since = time.time(), 0
        it += 1
        if args.max_iters and it >= args.max_iters:
            break


def evaluate(model, criterion, data_loader, device, print_freq=100):
    model.eval()
    logger = utils.MetricLogger(delimiter="  ")
    with torch.no_grad():
        for image, target in logger.log_every(data_loader, print_freq, "Test:"):
            image = image.to(device, non_blocking=True)
            target = target.to(device, non_blocking=True)
            output = model(image)
            if hasattr(criterion, "iif"):
                output = criterion(output, infer=True)
            acc1, acc5 = utils.accuracy(output, target, topk=(1, min(5, output.shape[1])))
            n = image.shape[0]
            logger.meters["acc1"].update(acc1.item(), n=n)
            logger.meters["acc5"].update(acc5.item(), n=n)
    logger.synchronize_between_processes()
    print(" * Acc@1 {top1.global_avg:.3f} Acc@5 {top5.global_avg:.3f}".format(top1=logger.acc1, top5=logger.acc5))
    return logger.acc1.global_avg


def build_model(args, num_classes):
    """train.py:184-187: look the constructor up in resnet_pytorch, then resnet_cifar."""
    cdt = torch.bfloat16 if args.compute_dtype == "bf16" else torch.float32
    kw = dict(num_classes=num_classes, use_norm=str(args.classif_norm), device=args.device, compute_dtype=cdt)
    if hasattr(resnet_pytorch, args.model):
        return getattr(resnet_pytorch, args.model)(pretrained=str(args.pretrained), **kw)
    if hasattr(resnet_cifar, args.model):
        return getattr(resnet_cifar, args.model)(**kw)
    raise AttributeError("unknown model %r" % (args.model,))


def main(args):
    if args.output_dir:
        utils.mkdir(args.output_dir)
    utils.init_distributed_mode(args)
    if not torch.cuda.is_available():
        raise SystemExit("iif_amd.train needs an MI355X: the native engine has no CPU path")
    device = torch.device(args.device)
    print(args)
    if getattr(args, "auto_augment", None) and not getattr(args, "data_path", None):
        import warnings
        warnings.warn("--auto-augment %r applies to the list datasets (--data-path); the synthetic long-tailed data of this run "
                      "is not augmented" % (args.auto_augment,))
    dataset, num_classes, data_loader, data_loader_test, train_sampler = initialisers.get_data(args)
    print("Creating model")
    model = build_model(args, num_classes)
    criterion = initialisers.get_criterion(args, dataset, model, num_classes)
    if args.sync_bn and args.distributed:
        model.enable_sync_bn()                # train.py:190-191: batch statistics over all ranks
    if args.opt.lower() not in ("sgd", "nesterov"):
        raise RuntimeError("Invalid optimizer {}. Only SGD and RMSprop are supported.".format(args.opt))
    if args.decoup:
        model.select_training_param()       # train.py:123-145: classifier-only stage
    reducer = None
    if args.distributed:
        broadcast_parameters(model)
        reducer = model.make_reducer(mode=args.reduce_mode)
    if args.resume:
        # the reference's checkpoint dict (train.py:265-271): model / optimizer / lr_scheduler / epoch / args
        ckpt = torch.load(args.resume, map_location="cpu", weights_only=False)
        model.load_state_dict(ckpt["model"])
        if "optimizer" in ckpt:
            model.load_optimizer_state_dict(ckpt["optimizer"])
        args.start_epoch = ckpt["epoch"] + 1
    if args.load_from:
        model.load_state_dict(torch.load(args.load_from, map_location="cpu", weights_only=False)["model"])
    if args.test_only:
        evaluate(model, criterion, data_loader_test, device=device)
        return
    print("Start training")
    start_time = time.time()
    best_acc = 0
    for epoch in range(args.start_epoch, args.epochs):
        if args.distributed:
            train_sampler.set_epoch(epoch)
        train_one_epoch(model, criterion, data_loader, device, epoch, args, reducer)
        if args.distributed:
            sync_buffers(model)               # rank 0's BN statistics everywhere, as DDP's broadcast_buffers
        acc = evaluate(model, criterion, data_loader_test, device=device)
        best_acc = max(best_acc, acc)
        if args.output_dir:
            nxt = lr_at(args, epoch + 1, 10 ** 9, 10 ** 9)
            ckpt = {"model": model.state_dict(),
                    "optimizer": model.optimizer_state_dict(nxt, args.momentum, args.weight_decay,
                                                            args.opt.lower() == "nesterov", initial_lr=args.lr),
                    "lr_scheduler": scheduler_state_dict(args, epoch + 1),
                    "epoch": epoch, "args": args}
            utils.save_on_master(ckpt, os.path.join(args.output_dir, "model_{}.pth".format(epoch)))
            utils.save_on_master(ckpt, os.path.join(args.output_dir, "checkpoint.pth"))
    print("Training time {}".format(datetime.timedelta(seconds=int(time.time() - start_time))))
    print("best acc is:", best_acc)


def get_args_parser(add_help=True):
    p = argparse.ArgumentParser(description="IIF classification training on MI355X", add_help=add_help)
    p.add_argument("--data-path", default="", help="dataset root of the list files; empty = synthetic long-tailed sets")
    p.add_argument("--train-txt", dest="train_txt", default=None, help="training list (default: the reference's path for --dset_name)")
    p.add_argument("--eval-txt", dest="eval_txt", default=None, help="evaluation list (default: the reference's path)")
    p.add_argument("--image-size", dest="image_size", default=224, type=int)
    p.add_argument("--dset_name", default="cifar100", help="cifar10|cifar100|imagenet_lt|places_lt|inat18 (synthetic)")
    p.add_argument("--rand_number", default=0, type=int)
    p.add_argument("--imb_type", default="exp", type=str)
    p.add_argument("--imb_factor", default=0.01, type=float)
    p.add_argument("--model", default="resnet32")
    p.add_argument("--device", default="cuda")
    p.add_argument("-b", "--batch-size", default=32, type=int)
    p.add_argument("--epochs", default=400, type=int, metavar="N")
    p.add_argument("-j", "--workers", default=4, type=int, metavar="N")
    p.add_argument("--opt", default="sgd", type=str)
    p.add_argument("--lr", default=0.1, type=float)
    p.add_argument("--cosine_scheduler", action="store_true")
    p.add_argument("--momentum", default=0.9, type=float, metavar="M")
    p.add_argument("--wd", "--weight-decay", default=1e-4, type=float, metavar="W", dest="weight_decay")
    p.add_argument("--milestones", nargs="+", default=[360, 380], type=int)
    p.add_argument("--lr-gamma", default=0.1, type=float)
    p.add_argument("--print-freq", default=100, type=int)
    p.add_argument("--output-dir", default="", help="path where to save")
    p.add_argument("--resume", default="")
    p.add_argument("--load_from", default="")
    p.add_argument("--classif", default="ce", type=str)
    p.add_argument("--classif_norm", default=None, type=str)
    p.add_argument("--gamma", default=0.0, type=float)
    p.add_argument("--alpha", default=None, type=float)
    p.add_argument("--iif", default="raw", type=str)
    p.add_argument("--iif_norm", default=0, type=int)
    p.add_argument("--decoup", action="store_true")
    p.add_argument("--mixup", default=None, type=float)
    p.add_argument("--sampler", default="random", type=str)
    p.add_argument("--reduction", default="mean", type=str)
    p.add_argument("--start-epoch", default=0, type=int, metavar="N")
    p.add_argument("--cache-dataset", dest="cache_dataset", action="store_true")
    p.add_argument("--sync-bn", dest="sync_bn", action="store_true",
                   help="cross-replica BN statistics (reference train.py:190); rehearsed over gloo and one RCCL rank only")
    p.add_argument("--test-only", dest="test_only", action="store_true")
    p.add_argument("--pretrained", dest="pretrained", default=None, type=str)
    p.add_argument("--deffered", action="store_true")
    p.add_argument("--auto-augment", default=None)
    p.add_argument("--random-erase", default=0.0, type=float)
    p.add_argument("--apex", action="store_true")
    p.add_argument("--apex-opt-level", default="O2", type=str)
    p.add_argument("--world-size", default=1, type=int)
    p.add_argument("--dist-url", default="env://")
    p.add_argument("--record-result", dest="record_result", action="store_true")
    # MI355X-native additions
    p.add_argument("--compute-dtype", default="bf16", choices=["bf16", "f32"])
    p.add_argument("--reduce-mode", dest="reduce_mode", default="allreduce", choices=["allreduce", "rs_ag"],
                   help="gradient buckets: one all_reduce each, or reduce_scatter + all_gather")
    p.add_argument("--bf16-buckets", dest="bf16_buckets", action="store_true",
                   help="reduce bf16 copies of the gradient buckets (refused unless the first step's probe stays in tolerance)")
    p.add_argument("--max-iters", default=0, type=int, help="stop each epoch after this many iterations (0 = all)")
    p.add_argument("--synthetic-scale", dest="synthetic_scale", default=1.0, type=float)
    return p


if __name__ == "__main__":
    main(get_args_parser().parse_args())
