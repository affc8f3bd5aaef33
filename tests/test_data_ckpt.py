"""Host-side rows of SURVEY §8f rank 4: list-file datasets, class-balanced sampling, shot accuracy and
checkpoint interoperability with the reference's dict (classification/train.py:265-271).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import iif_oracle as O


def test_lt_dataset_list_parsing_and_eval_remap(tmp_path):
    from iif_amd.imbalanced_dataset import LT_Dataset, LT_Dataset_Eval
    rng = np.random.RandomState(0)
    C = 7
    labels = rng.choice(C, size=300, p=np.array([1, 2, 4, 8, 16, 32, 64]) / 127.0)
    train = tmp_path / "train.txt"
    train.write_text("".join("img/%05d.jpg %d\n" % (i, l) for i, l in enumerate(labels)))
    ds = LT_Dataset("/data", str(train), C, loader=lambda p: p)
    cmap, remapped, counts = O.lt_class_map(labels.tolist(), C, kind="stable")
    assert ds.class_map == cmap and ds.targets == remapped and ds.get_cls_num_list() == counts
    assert counts == sorted(counts, reverse=True)                       # descending-frequency renumbering
    assert len(ds) == 300 and ds[3] == ("/data/img/00003.jpg", remapped[3])
    assert [len(c) for c in ds.class_data] == counts
    val = tmp_path / "val.txt"
    val.write_text("v/0.jpg 6\nv/1.jpg 0\n\n")
    ev = LT_Dataset_Eval("/data", str(val), ds.class_map, C, loader=lambda p: p)
    assert ev.targets == [cmap[6], cmap[0]] and len(ev) == 2


def test_lt_dataset_without_pil_fails_loudly(tmp_path):
    from iif_amd.imbalanced_dataset import LT_Dataset
    f = tmp_path / "t.txt"
    f.write_text("a.jpg 0\n")
    ds = LT_Dataset(str(tmp_path), str(f), 2)
    with pytest.raises((RuntimeError, FileNotFoundError)):
        ds[0]


def test_shot_acc_known_answer():
    from iif_amd.per_shot_acc import shot_acc
    train = [0] * 150 + [1] * 50 + [2] * 5
    labels = np.array([0, 0, 0, 0, 1, 1, 2, 2])
    preds = np.array([0, 0, 0, 1, 1, 0, 2, 2])
    many, med, low, per = shot_acc(preds, labels, train, acc_per_cls=True)
    assert (many, med, low) == (0.75, 0.5, 1.0) and per == [0.75, 0.5, 1.0]
    m2 = shot_acc(torch.tensor(preds), torch.tensor(labels), np.array(train))
    assert m2 == (0.75, 0.5, 1.0)
    with pytest.raises(TypeError):
        shot_acc(list(preds), labels, train)


@pytest.mark.parametrize("mode,per", [("downsampling", 5), ("upsampling", 60), (17, 17)])
def test_balance_class_sampler(mode, per):
    from iif_amd.samplers import BalanceClassSampler
    labels = [0] * 60 + [1] * 20 + [2] * 5
    s = BalanceClassSampler(labels, mode=mode)
    np.random.seed(0)
    idx = list(s)
    assert len(s) == len(idx) == 3 * per
    got = np.bincount(np.array(labels)[idx], minlength=3)
    assert got.tolist() == [per, per, per]
    for c, n in ((0, 60), (1, 20), (2, 5)):                 # without replacement whenever the class is large enough
        mine = [i for i in idx if labels[i] == c]
        if per <= n:
            assert len(set(mine)) == len(mine)


def test_distributed_sampler_wrapper_shards():
    from iif_amd.samplers import BalanceClassSampler, DistributedSamplerWrapper
    labels = [0] * 30 + [1] * 10
    shards = []
    for r in range(2):
        np.random.seed(1)
        w = DistributedSamplerWrapper(BalanceClassSampler(labels, "upsampling"), num_replicas=2, rank=r, shuffle=False)
        w.set_epoch(0)
        shards.append(list(w))
    assert len(shards[0]) == len(shards[1]) == 30
    np.random.seed(1)
    full = list(BalanceClassSampler(labels, "upsampling"))
    assert shards[0] == full[0::2] and shards[1] == full[1::2]


def test_optimizer_state_interoperates_with_torch_sgd():
    from iif_amd import resnet_cifar
    net = resnet_cifar.resnet20(num_classes=10, device="cpu", compute_dtype=torch.float32)
    g = torch.Generator().manual_seed(0)
    net._mom_arena.copy_(torch.randn(net._mom_arena.shape, generator=g))
    sd = net.optimizer_state_dict(0.05, 0.9, 1e-4, False, initial_lr=0.1)
    # a reference-side optimizer over same-shaped parameters accepts it as is
    params = [torch.nn.Parameter(torch.zeros_like(p)) for p in net.parameters()]
    opt = torch.optim.SGD(params, lr=0.1, momentum=0.9, weight_decay=1e-4)
    opt.load_state_dict(sd)
    assert opt.param_groups[0]["lr"] == 0.05
    for i, (p, v) in enumerate(zip(params, net._arena_views(net._mom_arena))):
        assert torch.equal(opt.state[p]["momentum_buffer"], v), i
    # and a torch optimizer's state comes back into the arena
    for p in params:
        opt.state[p]["momentum_buffer"].mul_(2.0)
    net2 = resnet_cifar.resnet20(num_classes=10, device="cpu", compute_dtype=torch.float32)
    assert net2.load_optimizer_state_dict(opt.state_dict()) == []
    for v1, v2 in zip(net._arena_views(net._mom_arena), net2._arena_views(net2._mom_arena)):
        assert torch.equal(v1 * 2.0, v2)


def test_scheduler_state_matches_torch():
    import argparse
    from iif_amd.train import scheduler_state_dict
    for cosine in (False, True):
        args = argparse.Namespace(lr=0.1, cosine_scheduler=cosine, epochs=20, lr_gamma=0.1, milestones=[8, 14])
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.SGD([p], lr=0.1, momentum=0.9)
        sch = (torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=20) if cosine
               else torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[8, 14], gamma=0.1))
        for e in range(10):
            opt.step(); sch.step()
        mine, ref = scheduler_state_dict(args, 10), sch.state_dict()
        assert mine["last_epoch"] == ref["last_epoch"] and abs(mine["_last_lr"][0] - ref["_last_lr"][0]) < 1e-12
        sch2 = (torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=20) if cosine
                else torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[8, 14], gamma=0.1))
        sch2.load_state_dict(mine)                                       # the reference's resume path (train.py:241)
        assert sch2.last_epoch == 10


def test_shot_acc_matches_per_class_scan():
    """The bincount formulation against a direct per-class scan on random data (present classes only, empty splits -> 0)."""
    from iif_amd.per_shot_acc import shot_acc
    rng = np.random.RandomState(3)
    for trial in range(5):
        C = 30
        train = rng.choice(C, size=4000, p=np.r_[np.full(5, 0.12), np.full(25, 0.016)])
        labels = rng.choice(np.arange(2, C), size=500)                 # classes 0, 1 absent from the test labels
        preds = np.where(rng.rand(500) < 0.6, labels, rng.choice(C, size=500))
        many, med, low, per = shot_acc(preds, labels, train, many_shot_thr=100, low_shot_thr=60, acc_per_cls=True)
        buckets = {"many": [], "med": [], "low": []}
        want_per = []
        for c in sorted(set(labels.tolist())):
            sel = labels == c
            acc = float((preds[sel] == c).mean())
            want_per.append(acc)
            n = int((train == c).sum())
            buckets["many" if n > 100 else ("low" if n < 60 else "med")].append(acc)
        want = [np.mean(v) if v else 0.0 for v in (buckets["many"], buckets["med"], buckets["low"])]
        assert np.allclose([many, med, low], want, rtol=0, atol=1e-12)
        assert np.allclose(per, want_per, rtol=0, atol=1e-12)
    assert shot_acc(np.array([1, 1]), np.array([1, 1]), [1] * 500) == (1.0, 0, 0)


def test_get_data_reads_list_files(tmp_path):
    """``--data-path`` reaches LT_Dataset / LT_Dataset_Eval through get_data (initialisers.py:83-100,
    imbalanced_dataset.py:177-259): class map by descending frequency, evaluation list remapped, tensors of the
    reference's geometry."""
    import types
    from iif_amd import initialisers
    rng = np.random.RandomState(0)
    (tmp_path / "img").mkdir()
    labels = [0] * 2 + [1] * 7 + [2] * 4
    lines = []
    for i, l in enumerate(labels):
        np.save(tmp_path / "img" / ("%d.npy" % i), rng.randint(0, 255, size=(40 + i, 50, 3), dtype=np.uint8))
        lines.append("img/%d.npy %d" % (i, l))
    (tmp_path / "train.txt").write_text("\n".join(lines) + "\n")
    (tmp_path / "eval.txt").write_text("\n".join(lines[:5]) + "\n")
    args = types.SimpleNamespace(dset_name="places_lt", data_path=str(tmp_path), train_txt=str(tmp_path / "train.txt"),
                                 eval_txt=str(tmp_path / "eval.txt"), image_size=32, rand_number=0, sampler="random",
                                 distributed=False, batch_size=4, workers=0)
    ds, C, loader, loader_test, sampler = initialisers.get_data(args)
    assert ds.cls_num_list[:3] == [7, 4, 2] and ds.class_map[:3] == [2, 0, 1]
    x, y = next(iter(loader))
    assert tuple(x.shape) == (4, 3, 32, 32) and x.dtype == torch.float32 and y.dtype == torch.int64
    xt, yt = next(iter(loader_test))
    assert tuple(xt.shape) == (4, 3, 32, 32) and yt.tolist() == [2, 2, 0, 0]       # raw 0,0,1,1 through the training class map


# ------------------------------------------------------------ a checkpoint written by the reference's own classes (golden G18)
def _g18():
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g18_ckpt_resnet32.pth")
    return path, torch.load(path, map_location="cpu", weights_only=False)      # the dict pickles an argparse.Namespace (train.py:270)


def test_reference_checkpoint_loads_into_the_native_model():
    """tests/golden/g18_ckpt_resnet32.pth was saved by classification/resnet_cifar.resnet32 + torch.optim.SGD + MultiStepLR through
    the dict of classification/train.py:265-271 (tests/golden/make_golden.py::g18_ckpt_resnet32).  `--resume`
    (train.py:236-241): model strict-loads, the momentum buffers land in the momentum arena, the scheduler state the native
    loop would have written equals the reference's, the next epoch's learning rate is the reference's.  `--load_from` /
    `pretrained=<path>` with another class count (resnet_pytorch.py:383-397 `_mismatched_classifier`): backbone taken,
    classifier left alone."""
    from iif_amd import resnet_cifar
    from iif_amd.resnet_pytorch import _load_mismatched
    from iif_amd.train import lr_at, scheduler_state_dict
    path, ckpt = _g18()
    assert set(ckpt) == {"model", "optimizer", "lr_scheduler", "epoch", "args"} and ckpt["epoch"] == 0
    net = resnet_cifar.resnet32(num_classes=100, use_norm="None", device="cpu", compute_dtype=torch.float32)
    net.load_state_dict(ckpt["model"])                                   # strict: same keys, same shapes
    own = net.state_dict()
    assert list(own.keys()) == list(ckpt["model"].keys())
    for k, v in ckpt["model"].items():
        assert torch.equal(own[k].cpu(), v), k
    assert net.load_optimizer_state_dict(ckpt["optimizer"]) == []
    for i, v in enumerate(net._arena_views(net._mom_arena)):
        assert torch.equal(v, ckpt["optimizer"]["state"][i]["momentum_buffer"]), i
    args = ckpt["args"]
    mine, ref = scheduler_state_dict(args, ckpt["epoch"] + 1), ckpt["lr_scheduler"]
    assert mine["last_epoch"] == ref["last_epoch"] and mine["_last_lr"] == ref["_last_lr"] and mine["milestones"] == ref["milestones"]
    assert lr_at(args, ckpt["epoch"] + 1, 10 ** 9, 10 ** 9) == ckpt["optimizer"]["param_groups"][0]["lr"]
    # and the native model's own checkpoint of this state is one the reference's optimizer accepts
    sd = net.optimizer_state_dict(ref["_last_lr"][0], args.momentum, args.weight_decay, False, initial_lr=args.lr)
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros_like(p)) for p in net.parameters()], lr=args.lr, momentum=args.momentum)
    opt.load_state_dict(sd)
    # another class count: everything but the classifier
    net10 = resnet_cifar.resnet32(num_classes=10, use_norm="None", device="cpu", compute_dtype=torch.float32)
    w_before, b_before = net10.linear.weight.detach().clone(), net10.linear.bias.detach().clone()
    _load_mismatched(net10, path)
    own10 = net10.state_dict()
    for k, v in ckpt["model"].items():
        if k.startswith("linear."):
            continue
        assert torch.equal(own10[k].cpu(), v), k
    assert torch.equal(net10.linear.weight.detach(), w_before) and torch.equal(net10.linear.bias.detach(), b_before)


@pytest.mark.gpu
def test_resume_from_reference_checkpoint_reproduces_the_reference_run(golden):
    """The two steps the REFERENCE took after writing the checkpoint (epoch 1, lr 0.01 behind the first milestone), taken by the
    native fp32 step after `--resume`: same losses (1e-6 for the first: weights and BN buffers alone; 1e-5 for the second: the
    momentum buffers and the update too), same final classifier rows and running statistics."""
    from iif_amd import resnet_cifar
    from iif_amd.custom import IIFLoss
    from iif_amd.train import lr_at
    g = golden("g18_ckpt_resnet32")
    path, ckpt = _g18()
    args = ckpt["args"]

    class DS:
        def get_cls_num_list(self):
            return [int(c) for c in g["counts"]]
    net = resnet_cifar.resnet32(num_classes=100, use_norm="None", compute_dtype=torch.float32)
    net.load_state_dict(ckpt["model"])
    assert net.load_optimizer_state_dict(ckpt["optimizer"]) == []
    epoch = ckpt["epoch"] + 1
    lr = lr_at(args, epoch, 10 ** 9, 10 ** 9)
    assert abs(lr - float(g["lrs"][2])) < 1e-15 and abs(lr - float(g["lrs"][3])) < 1e-15
    crit = IIFLoss(DS(), variant="raw")
    x, y = torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["y"]).cuda()
    net.train()
    for i, tol in ((2, 1e-6), (3, 1e-5)):
        loss, _ = net.loss_and_backward(x, y, crit)
        net.sgd_step(lr, args.momentum, args.weight_decay)
        assert abs(loss.item() - float(g["losses"][i])) <= tol * abs(float(g["losses"][i])), (i, loss.item(), float(g["losses"][i]))
    sd = net.state_dict()
    assert (sd["linear.weight"][:4].cpu() - torch.from_numpy(g["final_linear"])).abs().max().item() <= 1e-5 * np.abs(g["final_linear"]).max()
    assert (sd["bn1.running_mean"].cpu() - torch.from_numpy(g["final_bn1_rm"])).abs().max().item() <= 1e-5 * max(np.abs(g["final_bn1_rm"]).max(), 1e-3)
