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
    """``mode``:
      * ``"allreduce"`` (default): one ``all_reduce(SUM)`` per bucket;
      * ``"rs_ag"``: ``reduce_scatter_tensor`` into this rank's 1/world shard of the bucket followed by
        ``all_gather_into_tensor`` back into the arena (on the fully connected xGMI node the
        7 peer shards of either phase travel over the 7 links concurrently instead of around one ring).  Bucket
        bounds are multiples of 16 elements (arena tensors are 64-byte aligned), which divides by world sizes 2, 4, 8
        and 16 only; for any other world size (3, 5, 6, 7 ...) the constructor keeps just the cuts whose offset is a
        multiple of the world size (and refuses an arena whose length is not).  The sum arrives in a different
        association than the ring's, fp32-rounding apart.
      Both modes, the bf16 staging buckets and the SyncBatchNorm collectives have been rehearsed over gloo (CPU, and two
      ranks on one GPU) and over RCCL with ONE rank only: no multi-GPU node has been available to the build, so the
      stream-ordered ``nccl`` branch with world > 1 has not run on hardware yet.
    ``bucket_dtype=torch.bfloat16`` halves the bytes on the links by reducing a bf16 copy of each bucket (fp32
    accumulation is lost: it is refused until ``probe_bf16`` has measured, on real gradients, that the averaged
    gradient stays within a stated tolerance of the fp32 reduction)."""

    def __init__(self, grad_arena, boundaries, bucket_bytes=32 << 20, process_group=None, tail_bytes=2 << 20,
                 mode="allreduce", bucket_dtype=torch.float32, cu_budget=None):
        """grad_arena: flat fp32 tensor; boundaries: sorted arena offsets where a bucket
        may start (tensor starts, in elements).  The LAST bucket (the arena's head: stem and
        first blocks) can only start when backward has finished, so its reduction is exposed:
        it is kept below ``tail_bytes`` by one extra cut."""
        if mode not in ("allreduce", "rs_ag"):
            raise ValueError("unknown reducer mode %r" % (mode,))
        self.arena = grad_arena
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self.mode = mode
        self.bucket_dtype = torch.float32
        self._bf16_cleared = False
        n = grad_arena.numel()
        cuts = sorted(set(int(b) for b in boundaries if 0 < int(b) < n))
        if mode == "rs_ag":
            if n % max(self.world, 1):
                raise ValueError("rs_ag needs the arena length (%d) to divide by the world size (%d)" % (n, self.world))
            cuts = [c for c in cuts if c % self.world == 0]
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
        self._staging = {}         # bucket -> bf16 staging / shard buffers
        self.force = False         # exercise the bucket/stream machinery even with a single rank (tests)
        self.extra_streams = []    # other producer streams of gradients (the engine's wgrad stream)
        self.launched = 0          # collectives enqueued since construction (bench / tests read it)
        self.steps = 0
        # Compute units the engine's persistent kernels (one resident block per CU) may take while this reducer's collectives
        # are in flight (iif_set_cu_budget, include/iif_amd.h): RCCL's reduction kernels are ordinary workgroups that need a
        # free CU each, and a grid that holds all 256 makes them queue until a block exits.  Default with world > 1: 240 (two CUs
        # per XCD left over; RCCL runs up to 32 channels of one workgroup); 0 = no reservation.  IIF_REDUCER_CUS / cu_budget=
        # override.  The price of the reservation on one GPU is in profiles/r5_reducer_cu_budget.txt; what it buys needs >= 2 GPUs.
        import os
        if cu_budget is None:
            env = os.environ.get("IIF_REDUCER_CUS")
            cu_budget = int(env) if env is not None else (240 if self.world > 1 else 0)
        self.cu_budget = int(cu_budget)
        if bucket_dtype != torch.float32:
            self.set_bucket_dtype(bucket_dtype)

    # ---- configuration --------------------------------------------------------
    def probe_bf16(self, tolerance=4e-3):
        """Measure, on the gradients currently in the arena (call after one backward, before the reduction),
        what reducing bf16 copies would do to the averaged gradient: per bucket, the relative L2 distance between
        the bf16-bucket average and the fp32 average.  Clears ``set_bucket_dtype(bfloat16)`` when every bucket stays
        within ``tolerance`` (bf16 has 8 mantissa bits: 2^-8 = 3.9e-3 is one rounding); returns the worst value.
        Leaves the arena untouched."""
        worst = 0.0
        for lo, hi in self.buckets:
            ref = self.arena[lo:hi].clone()
            low = ref.to(torch.bfloat16)
            if self.world > 1:
                dist.all_reduce(ref, group=self.group)
                dist.all_reduce(low, group=self.group)
            err = (low.float() - ref).norm() / ref.norm().clamp_min(1e-30)
            worst = max(worst, float(err))
        self._bf16_cleared = worst <= tolerance
        return worst

    def set_bucket_dtype(self, dtype):
        if dtype == torch.float32:
            self.bucket_dtype = dtype
            return
        if dtype != torch.bfloat16:
            raise ValueError("bucket dtype must be float32 or bfloat16")
        if not self._bf16_cleared:
            raise RuntimeError("bf16 gradient buckets are refused until probe_bf16() has shown that the averaged gradient "
                               "stays within tolerance of the fp32 reduction on this model's gradients")
        self.bucket_dtype = dtype

    def describe(self):
        """What the collective library is asked to do per step (goes into the bench JSON line)."""
        esz = 2 if self.bucket_dtype == torch.bfloat16 else 4
        return {"world": self.world, "mode": self.mode, "bucket_dtype": "bf16" if esz == 2 else "f32",
                "buckets": len(self.buckets), "bucket_bytes": [int((hi - lo) * esz) for lo, hi in self.buckets],
                "payload_bytes_per_step": int(self.arena.numel() * esz), "tail_bucket_bytes": int((self.buckets[-1][1] - self.buckets[-1][0]) * esz),
                "collectives_launched": self.launched, "steps_reduced": self.steps, "cu_budget": self.cu_budget}

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

    def _join(self):
        if self.use_streams:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        else:
            for w in self._works:
                w()
        self._works = []
        self.steps += 1

    def finish(self):
        """Launch whatever is left and make the compute stream wait for every reduction."""
        if self.world == 1 and not self.force:
            return
        self.gradients_ready_from(0)
        self._join()

    def finish_tail(self, offset):
        """Frozen-backbone step: only arena[offset:] carries gradients; reduce exactly that slice."""
        if self.world == 1 and not self.force:
            return
        offset = int(offset)
        if self.mode == "rs_ag":
            offset -= offset % self.world            # shard-divisible slice (a few frozen, zero gradients ride along)
        self._launch((offset, self.arena.numel()))
        self._next = len(self.buckets)
        self._join()

    # ---- internals ------------------------------------------------------------
    def _collective(self, bucket, sync):
        """Enqueue the reduction of one bucket; returns a completion callable for the host-waited (CPU) path."""
        lo, hi = bucket
        view = self.arena[lo:hi]
        low = self.bucket_dtype == torch.bfloat16
        buf = view
        if low:
            buf = self._staging.get(("low", bucket))
            if buf is None:
                buf = self._staging[("low", bucket)] = torch.empty(hi - lo, dtype=torch.bfloat16, device=view.device)
            buf.copy_(view)
        waits = []
        if self.mode == "rs_ag" and self.world > 1:
            per = (hi - lo) // self.world
            shard = self._staging.get(("shard", bucket, buf.dtype))
            if shard is None:
                shard = self._staging[("shard", bucket, buf.dtype)] = torch.empty(per, dtype=buf.dtype, device=view.device)
            w1 = dist.reduce_scatter_tensor(shard, buf, op=dist.ReduceOp.SUM, group=self.group, async_op=not sync)
            if not sync:
                w1.wait()                             # gloo: the gather reads what the scatter wrote
            w2 = dist.all_gather_into_tensor(buf, shard, group=self.group, async_op=not sync)
            waits = [] if sync else [w2]
            self.launched += 2
        else:
            w = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=not sync)
            waits = [] if sync else [w]
            self.launched += 1

        def done():
            for w in waits:
                w.wait()
            if low:
                view.copy_(buf)
        if sync and low:
            view.copy_(buf)                           # same stream, after the collective
        return done

    def _launch(self, bucket):
        if self.use_streams:
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            for s in self.extra_streams:
                self.comm_stream.wait_stream(s)
            with torch.cuda.stream(self.comm_stream):
                self._collective(bucket, sync=True)
        else:
            self._works.append(self._collective(bucket, sync=False))

    @property
    def grad_scale(self):
        """Factor the optimizer applies to the summed gradients (DDP averages)."""
        return 1.0 / self.world


def broadcast_parameters(net, src=0, process_group=None):
    """Replicate rank ``src``'s parameters and BN buffers (what DDP does at wrap time)."""
    if not dist.is_initialized() or dist.get_world_size(process_group) == 1:
        return
    dist.broadcast(net.param_arena, src, group=process_group)
    sync_buffers(net, src, process_group)


def sync_buffers(net, src=0, process_group=None):
    """BN running statistics and counters of rank ``src`` on every rank.  DistributedDataParallel broadcasts the
    buffers before every forward (``broadcast_buffers=True``, classification/train.py:232), so the reference
    evaluates and checkpoints with rank 0's statistics; training-mode forwards never read them, so broadcasting
    right before ``evaluate`` / a checkpoint is equivalent and costs nothing per step."""
    if not dist.is_initialized() or dist.get_world_size(process_group) == 1:
        return
    dist.broadcast(net._rstat, src, group=process_group)
    dist.broadcast(net._nbt, src, group=process_group)
