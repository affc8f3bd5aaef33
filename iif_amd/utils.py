"""Runtime helpers with the surface of the reference's ``classification/utils.py``:
top-k accuracy (native hit-count kernel), smoothed meters, the first-epoch
warm-up schedule, torch.distributed bring-up over RCCL and rank-0 checkpoint
saving.  Citations are relative to /root/reference/classification/.
"""
import datetime
import os
import time
from collections import defaultdict, deque

import torch
import torch.distributed as dist

from . import _lib


# ------------------------------------------------------------------- accuracy
def topk_hit_counts(output, target, topk=(1,), table=None):
    """int32 device tensor: rows whose target is within the top-k of
    ``output`` (optionally of ``output*table`` without materialising it)."""
    _lib.require_gpu(output, target, table)
    x = output.detach()
    if x.stride(-1) != 1:
        x = x.contiguous()
    B, C = x.shape
    ks = (torch.tensor(list(topk), dtype=torch.int32)).contiguous()
    hits = torch.zeros(len(topk), dtype=torch.int32, device=x.device)
    tab = None if table is None else table.to(torch.float32).contiguous()
    rc = _lib.lib().iif_topk_hits(_lib.ptr(x), _lib.dtype_code(x), x.stride(0) if B else C, _lib.ptr(tab),
                                  _lib.ptr(target.to(torch.int64).contiguous()), B, C, ks.data_ptr(), len(topk),
                                  _lib.ptr(hits), _lib.stream_ptr())
    _lib.check(rc, "iif_topk_hits")
    return hits


def accuracy(output, target, topk=(1,)):
    """Top-k accuracies in percent, one 0-dim float32 device tensor per k
    (utils.py:165-179).  No host sync: callers ``.item()`` when they log."""
    hits = topk_hit_counts(output, target, topk)
    pct = hits.to(torch.float32) * (100.0 / target.size(0))
    return [pct[i] for i in range(len(topk))]


# --------------------------------------------------------------------- meters
class SmoothedValue(object):
    """One logged scalar: the last ``window_size`` values (median / mean / max / last) and the running total over everything
    seen (``global_avg``).  Same public surface as the reference's meter (utils.py:13-73: ``update``, the five read-only
    statistics, ``synchronize_between_processes``, ``str()`` through ``fmt``); plain Python floats inside, no tensors."""

    _FIELDS = ("median", "avg", "global_avg", "max", "value")

    def __init__(self, window_size=20, fmt=None):
        self.fmt = "{median:.4f} ({global_avg:.4f})" if fmt is None else fmt
        self.deque = deque(maxlen=window_size)      # (the attribute name is part of the surface: checkpoints pickle it)
        self.total, self.count = 0.0, 0

    def update(self, value, n=1):
        self.deque.append(value)
        self.count += n
        self.total += value * n

    def synchronize_between_processes(self):
        """count and total become global sums (fp64 SUM all-reduce behind a barrier, utils.py:31-43); the window stays local."""
        if not is_dist_avail_and_initialized():
            return
        pair = torch.tensor([self.count, self.total], dtype=torch.float64,
                            device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.barrier()
        dist.all_reduce(pair)
        self.count, self.total = int(pair[0].item()), pair[1].item()

    @property
    def median(self):
        w = sorted(self.deque)
        return float(w[(len(w) - 1) // 2])          # the LOWER middle element for even counts, as torch.median gives

    @property
    def avg(self):
        return float(sum(self.deque)) / len(self.deque)

    @property
    def global_avg(self):
        return self.total / self.count

    @property
    def max(self):
        return max(self.deque)

    @property
    def value(self):
        return self.deque[-1]

    def __str__(self):
        return self.fmt.format(**{k: getattr(self, k) for k in self._FIELDS})


class MetricLogger(object):
    """A dictionary of meters that prints itself, plus the progress-printing loop wrapper of the training scripts
    (utils.py:76-162): ``for batch in logger.log_every(loader, freq, header)``."""

    def __init__(self, delimiter="\t"):
        self.meters = defaultdict(SmoothedValue)
        self.delimiter = delimiter

    def add_meter(self, name, meter):
        self.meters[name] = meter

    def update(self, **kwargs):
        for name, v in kwargs.items():
            v = v.item() if isinstance(v, torch.Tensor) else v
            if not isinstance(v, (float, int)):
                raise TypeError("meter %r fed with %r" % (name, type(v).__name__))
            self.meters[name].update(v)

    def __getattr__(self, attr):              # logger.loss -> the meter called "loss"
        meters = self.__dict__.get("meters")
        if meters is not None and attr in meters:
            return meters[attr]
        raise AttributeError("'%s' object has no attribute '%s'" % (type(self).__name__, attr))

    def __str__(self):
        return self.delimiter.join("%s: %s" % item for item in self.meters.items())

    def synchronize_between_processes(self):
        for meter in self.meters.values():
            meter.synchronize_between_processes()

    def _progress_line(self, header, i, n, step_time, load_time):
        remaining = datetime.timedelta(seconds=int(step_time.global_avg * (n - i)))
        parts = [header, "[%*d/%d]" % (len(str(n)), i, n), "eta: %s" % remaining, str(self), "time: %s" % step_time,
                 "data: %s" % load_time]
        if torch.cuda.is_available():
            parts.append("max mem: %.0f" % (torch.cuda.max_memory_allocated() / float(1 << 20)))
        return self.delimiter.join(parts)

    def log_every(self, iterable, print_freq, header=None):
        header = header or ""
        n = len(iterable)
        step_time, load_time = SmoothedValue(fmt="{avg:.4f}"), SmoothedValue(fmt="{avg:.4f}")
        began = mark = time.time()
        for i, item in enumerate(iterable):
            load_time.update(time.time() - mark)
            yield item
            step_time.update(time.time() - mark)
            if i % print_freq == 0:
                print(self._progress_line(header, i, n, step_time, load_time))
            mark = time.time()
        print("%s Total time: %s" % (header, datetime.timedelta(seconds=int(time.time() - began))))


# ------------------------------------------------------------------- schedules
def warmup_factor(it, warmup_iters, warmup_factor0):
    """Multiplier of the first-epoch linear warm-up (utils.py:182-189)."""
    if it >= warmup_iters:
        return 1
    a = float(it) / warmup_iters
    return warmup_factor0 * (1 - a) + a


def warmup_lr_scheduler(optimizer, warmup_iters, warmup_factor0):
    return torch.optim.lr_scheduler.LambdaLR(optimizer, lambda it: warmup_factor(it, warmup_iters, warmup_factor0))


# ---------------------------------------------------------------- distributed
def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def is_main_process():
    return get_rank() == 0


def mkdir(path):
    os.makedirs(path, exist_ok=True)


def save_on_master(*args, **kwargs):
    if is_main_process():
        torch.save(*args, **kwargs)


def setup_for_distributed(is_master):
    """Silence ``print`` on non-zero ranks unless ``force=True`` (utils.py:199-211)."""
    import builtins
    base = builtins.print

    def rank0_print(*a, **kw):
        if kw.pop("force", False) or is_master:
            base(*a, **kw)

    builtins.print = rank0_print


def init_distributed_mode(args):
    """One process per GPU from the launcher's env (utils.py:243-266).

    RANK / WORLD_SIZE / LOCAL_RANK (torchrun) or SLURM_PROCID.  Backend
    ``nccl`` — which is RCCL on ROCm, running over xGMI inside a node — when a
    GPU is present, ``gloo`` otherwise (CPU rehearsal of the multi-rank path).
    """
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:
        args.rank = int(os.environ["RANK"])
        args.world_size = int(os.environ["WORLD_SIZE"])
        args.gpu = int(os.environ.get("LOCAL_RANK", 0))
    elif "SLURM_PROCID" in os.environ:
        args.rank = int(os.environ["SLURM_PROCID"])
        args.gpu = args.rank % max(torch.cuda.device_count(), 1)
    elif hasattr(args, "rank"):
        pass
    else:
        print("Not using distributed mode")
        args.distributed = False
        return
    args.distributed = True
    # IIF_REHEARSE_ONE_GPU=1: every rank on GPU 0 with gloo carrying the collectives (RCCL refuses two ranks on one
    # device) — the multi-rank control flow of train.py on a one-GPU box (tests/test_ddp_gpu.py); never a measured path
    rehearse = bool(os.environ.get("IIF_REHEARSE_ONE_GPU"))
    if rehearse:
        args.gpu = 0
    if torch.cuda.is_available():
        torch.cuda.set_device(args.gpu)
        args.dist_backend = "gloo" if rehearse else "nccl"
    else:
        args.dist_backend = "gloo"
    print("| distributed init (rank %d): %s" % (args.rank, args.dist_url), flush=True)
    kw = {}
    if args.dist_backend == "nccl":
        kw["device_id"] = torch.device("cuda", args.gpu)
    dist.init_process_group(backend=args.dist_backend, init_method=args.dist_url, world_size=args.world_size,
                            rank=args.rank, **kw)
    setup_for_distributed(args.rank == 0)
