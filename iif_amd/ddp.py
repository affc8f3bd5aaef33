"""Data-parallel gradient reduction over the flat gradient arena.

Replaces ``torch.nn.parallel.DistributedDataParallel(model)`` at
classification/train.py:230-234 for the native engine: one process per GPU
(``torch.distributed``, backend ``nccl`` = RCCL over xGMI on ROCm), parameters
replicated, every rank back-propagates its own batch and the gradients are
summed across ranks and divided by the world size.

Because all gradients live in ONE contiguous fp32 arena laid out in forward
order, and backward completes it from the end towards the start, buckets are
plain contiguous slices: as soon as backward has passed a bucket's lowest
offset, the slice is all-reduced on a side stream while the remaining backward
kernels keep the compute stream busy.  xGMI is point-to-point (per-link bound),
so buckets are large (default 32 MB) — few, big collectives.  The 1/world
factor is folded into the fused SGD launch (``grad_scale``), so no extra pass
touches the gradients.
"""
import torch
import torch.distributed as dist


class ArenaReducer(object):
    def __init__(self, grad_arena, boundaries, bucket_bytes=32 << 20, process_group=None, tail_bytes=2 << 20):
        """grad_arena: flat fp32 tensor; boundaries: sorted arena offsets where a bucket
        may start (tensor starts, in elements).  The LAST bucket (the arena's head: stem and
        first blocks) can only start when backward has finished, so its reduction is exposed:
        it is kept below ``tail_bytes`` by one extra cut."""
        self.arena = grad_arena
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        n = grad_arena.numel()
        cuts = sorted(set(int(b) for b in boundaries if 0 < int(b) < n))
        # walk from the END of the arena (first gradients to complete) towards the start
        target = max(bucket_bytes // 4, 1)
        self.buckets = []          # (lo, hi) in completion order
        hi = n
        for c in reversed(cuts):
            if hi - c >= target:
                self.buckets.append((c, hi))
                hi = c
        if hi > 0:
            tail = max(tail_bytes // 4, 1)
            small = [c for c in cuts if c < hi and c <= tail]
            if hi > tail and small:
                self.buckets.append((small[-1], hi))
                hi = small[-1]
            self.buckets.append((0, hi))
        self.use_streams = grad_arena.is_cuda
        self.comm_stream = torch.cuda.Stream(device=grad_arena.device) if self.use_streams else None
        self._next = 0
        self._works = []
        self.force = False         # exercise the bucket/stream machinery even with a single rank (tests)
        self.extra_streams = []    # other producer streams of gradients (the engine's wgrad stream)

    # ---- called by the engine -------------------------------------------------
    def begin(self):
        self._next = 0
        self._works = []

    def gradients_ready_from(self, offset):
        """All gradients at arena offsets >= ``offset`` are final on the compute stream."""
        if self.world == 1 and not self.force:
            return
        while self._next < len(self.buckets) and self.buckets[self._next][0] >= offset:
            self._launch(self.buckets[self._next])
            self._next += 1

    def finish(self):
        """Launch whatever is left and make the compute stream wait for every reduction."""
        if self.world == 1 and not self.force:
            return
        self.gradients_ready_from(0)
        if self.use_streams:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        else:
            for w in self._works:
                w.wait()
        self._works = []

    def finish_tail(self, offset):
        """Frozen-backbone step: only arena[offset:] carries gradients; reduce exactly that slice."""
        if self.world == 1 and not self.force:
            return
        self._launch((int(offset), self.arena.numel()))
        self._next = len(self.buckets)
        if self.use_streams:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        else:
            for w in self._works:
                w.wait()
        self._works = []

    # ---- internals ------------------------------------------------------------
    def _launch(self, bucket):
        lo, hi = bucket
        view = self.arena[lo:hi]
        if self.use_streams:
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            for s in self.extra_streams:
                self.comm_stream.wait_stream(s)
            with torch.cuda.stream(self.comm_stream):
                dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)
        else:
            self._works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    @property
    def grad_scale(self):
        """Factor the optimizer applies to the summed gradients (DDP averages)."""
        return 1.0 / self.world


def broadcast_parameters(net, src=0, process_group=None):
    """Replicate rank ``src``'s parameters and BN buffers (what DDP does at wrap time)."""
    if not dist.is_initialized() or dist.get_world_size(process_group) == 1:
        return
    dist.broadcast(net.param_arena, src, group=process_group)
    dist.broadcast(net._rstat, src, group=process_group)
    dist.broadcast(net._nbt, src, group=process_group)
