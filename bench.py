#!/usr/bin/env python3
"""Headline benchmark: images/sec of a full ResNet50 + IIF training step on
synthetic ImageNet-LT-shaped batches, bs=256 per GPU, bf16 storage / fp32
accumulate (BASELINE.json configs[1]; weak scaling over N GPUs of one node).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = forward + fused IIF loss + backward + gradient all-reduce (N>1) + fused
SGD, all through the hand-written gfx950 kernels (libiif_amd.so); inputs are
resident in HBM before the timed region.  Rank 0 prints ONE JSON line.

Extra objects in that line:
  roofline     - the step against both roofs, from THIS run's wall clock (nothing in it can exceed
                 ms_per_step): algorithmic FLOPs of the convolution launches of one step / ms_per_step
                 against the dense bf16 MFMA peak (`frac`), and `hbm` = algorithmic bytes of one step
                 (every convolution operand and result once + the BN / pooling / SGD passes this
                 dataflow cannot avoid, formula in `hbm.formula`) / ms_per_step against 8 TB/s, which
                 is the roof that binds.  `brackets` keeps the HIP-event table per launch kind; those
                 durations OVERLAP (three streams) and are labelled so.  profiles/README.md holds the
                 recipe that recomputes every number from profiles/r3_*.
  cpu_baseline - the CPU oracle (torch-CPU restatement of the reference step,
                 oracle/) timed on the host cores of this box on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

# The step runs on three HIP streams (main, weight gradients, RCCL).  The runtime maps streams onto
# GPU_MAX_HW_QUEUES hardware queues (default 4); when two of ours share one, they serialise and a wait on one
# blocks the other (measured: -11 % with the process group's extra streams).  Must be set before HIP starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch                       # noqa: E402
import torch.distributed as dist   # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0      # MI355X dense bf16 (guides/MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0


def lt_counts(C, top, bottom=5):
    return [max(int(top * (bottom / top) ** (i / max(C - 1.0, 1.0))), 1) for i in range(C)]


class _Counts(object):
    def __init__(self, c):
        self.c = c

    def get_cls_num_list(self):
        return self.c


class ConvTimer(object):
    """HIP-event brackets around every MFMA convolution launch (on the stream the
    launch goes to: the kernels are enqueued on torch's current stream)."""

    def __init__(self):
        self.records = []      # (kind, flops, ideal_bytes, start_event, stop_event)
        self.shapes = []
        self.active = True     # brackets are recorded on a sample of the timed steps only (they cost ~3 % when on)
        self.sampled_steps = 0

    def wrap(self, ops):
        timer = self
        orig = {k: getattr(ops, k) for k in ("conv_forward", "conv_dgrad", "conv_wgrad", "conv_forward_bnstats",
                                             "conv_dgrad_bnbwd", "conv_dgrad_masksum", "conv_dgrad2_bnbwd", "wgrad1x1_stacked",
                                             "conv_forward_stats_acc", "conv_forward_bn_relu2", "conv_dgrad_masksum_rx",
                                             "conv_forward_bnstats_pro", "conv_dgrad_masksum_rx_pg")}

        def alg_k(r, s, stride, pad, cin):
            # algorithmic K of one output: the space-to-depth stem (4x4/1 pad 2 on 16 padded channels) is charged
            # for the 7*7*3 MACs of the convolution it implements, not for its zero padding
            return 147.0 if (r, s, stride, pad) == (4, 4, 1, 2) and cin in (16, 32) else float(r * s * cin)

        def flops_fwd(x, w, r, s, stride, pad, groups=1, out_hw=None, **kw):
            # algorithmic FLOPs: a chunked grouped conv is charged for its in-chunk MACs only
            n, h, wd, cin = x.shape
            ho, wo = out_hw or ops.conv_out_hw(h, wd, r, s, stride, pad)
            return 2.0 * n * ho * wo * w.shape[0] * alg_k(r, s, stride, pad, cin) / groups

        def nbytes(*ts):
            # ideal HBM bytes of a launch: every operand / result tensor moved exactly once
            return float(sum(t.numel() * t.element_size() for t in ts if t is not None))

        def conv_forward(x, w, r, s, stride, pad, **kw):
            n, h, wd, _ = x.shape
            ho, wo = kw.get("out_hw") or ops.conv_out_hw(h, wd, r, s, stride, pad)
            out_b = n * ho * wo * w.shape[0] * (kw["out"].element_size() if kw.get("out") is not None else x.element_size())
            by = nbytes(x, w, kw.get("res")) + out_b
            return timer._timed("fwd", flops_fwd(x, w, r, s, stride, pad, **{k: v for k, v in kw.items() if k in ("groups", "out_hw")}),
                                by, orig["conv_forward"], x, w, r, s, stride, pad, **kw)

        def conv_dgrad(dy, wt, r, s, stride, pad, in_hw, **kw):
            n, ho, wo, cout = dy.shape
            fl = 2.0 * n * ho * wo * cout * r * s * wt.shape[0] / kw.get("groups", 1)
            by = nbytes(dy, wt, kw.get("res")) + n * in_hw[0] * in_hw[1] * wt.shape[0] * dy.element_size()
            return timer._timed("dgrad", fl, by, orig["conv_dgrad"], dy, wt, r, s, stride, pad, in_hw, **kw)

        def conv_wgrad(x, dy, r, s, stride, pad, **kw):
            n, ho, wo, cout = dy.shape
            fl = 2.0 * n * ho * wo * cout * alg_k(r, s, stride, pad, x.shape[3]) / kw.get("groups", 1)
            if x.data_ptr() == dy.data_ptr():
                fl = 0.0           # a2^T a2 of the algebraic BN3 backward: work this design adds, not algorithmic
            by = nbytes(x, dy) + 4.0 * cout * r * s * x.shape[3] / kw.get("groups", 1)
            return timer._timed("wgrad", fl, by, orig["conv_wgrad"], x, dy, r, s, stride, pad, **kw)

        def conv_forward_bnstats(x, w, r, s, stride, pad, out, partial, **kw):
            fl = flops_fwd(x, w, r, s, stride, pad, out_hw=(out.shape[1], out.shape[2]), **kw)
            return timer._timed("fwd", fl, nbytes(x, w, out), orig["conv_forward_bnstats"], x, w, r, s, stride, pad, out, partial, **kw)

        def conv_dgrad_bnbwd(dy, wt, r, s, stride, pad, in_hw, out, up_x, up_bits, up_stats, partial, **kw):
            n, ho, wo, cout = dy.shape
            fl = 2.0 * n * ho * wo * cout * r * s * wt.shape[0] / kw.get("groups", 1)
            by = nbytes(dy, wt, kw.get("res"), up_x) + out.numel() * out.element_size()
            return timer._timed("dgrad", fl, by, orig["conv_dgrad_bnbwd"], dy, wt, r, s, stride, pad, in_hw, out, up_x, up_bits,
                                up_stats, partial, **kw)

        def conv_dgrad_masksum(dy, wt, in_hw, out, up_bits, partial, **kw):
            n, ho, wo, cout = dy.shape
            fl = 2.0 * n * ho * wo * cout * wt.shape[0]
            by = nbytes(dy, wt, kw.get("res"), kw.get("up_x"), out)
            return timer._timed("dgrad", fl, by, orig["conv_dgrad_masksum"], dy, wt, in_hw, out, up_bits, partial, **kw)

        def conv_dgrad_masksum_rx(dy, wt, in_hw, out, up_bits, partial, up_a2, up_w3, up_stats, **kw):
            # the producer that recomputes the upstream conv3 tile: charged the data gradient's FLOPs (the recomputation is work this
            # design adds), bytes: its operands plus the narrow a2 and the weights it recomputes from
            n, ho, wo, cout = dy.shape
            fl = 2.0 * n * ho * wo * cout * wt.shape[0]
            by = nbytes(dy, wt, kw.get("res"), out, up_a2, up_w3)
            return timer._timed("dgrad", fl, by, orig["conv_dgrad_masksum_rx"], dy, wt, in_hw, out, up_bits, partial, up_a2, up_w3, up_stats, **kw)

        def conv_dgrad_masksum_rx_pg(dy, wt, in_hw, out, up_bits, partial, up_a2, up_w3, up_stats, pg_slabs, pg_ld, **kw):
            # ... that also leaves P = g~^T a2 (the upstream conv3's weight-gradient GEMM: its FLOPs are charged here, to this
            # launch's kind) and Gram behind; bytes: as above plus the fp32 slabs it writes (one per resident block)
            n, ho, wo, cout = dy.shape
            fl = 2.0 * n * ho * wo * cout * wt.shape[0] + 2.0 * n * ho * wo * up_a2.shape[3] * wt.shape[0]
            by = nbytes(dy, wt, kw.get("res"), out, up_a2, up_w3) + 256 * (wt.shape[0] + up_a2.shape[3]) * pg_ld * 4
            return timer._timed("dgrad", fl, by, orig["conv_dgrad_masksum_rx_pg"], dy, wt, in_hw, out, up_bits, partial, up_a2, up_w3, up_stats,
                                pg_slabs, pg_ld, **kw)

        def conv_dgrad2_bnbwd(src, src2, wt, bias, out, *a, **kw):
            # the data gradient of conv3 with BN3's backward folded into the weights: charged the FLOPs of the plain
            # data gradient (K = src's channels); the second K source is overhead of the design, its bytes are counted
            n, h, wd, c1 = src.shape
            fl = 2.0 * n * h * wd * c1 * wt.shape[0]
            by = nbytes(src, src2, wt, out, a[0] if a else kw.get("up_x"))
            return timer._timed("dgrad", fl, by, orig["conv_dgrad2_bnbwd"], src, src2, wt, bias, out, *a, **kw)

        def wgrad1x1_stacked(x2d, dy2d, dy2_2d, out, workspace, **kw):
            # [g~ | a2]^T a2: P is conv3's weight gradient (charged as conv_wgrad charges it), the Gram rows are work this design
            # adds (0 FLOPs, as for conv_wgrad(a2, a2)); bytes: a2 and g~ once (dy2 IS x), the fp32 result
            fl = 2.0 * x2d.shape[0] * x2d.shape[1] * dy2d.shape[1]
            by = nbytes(x2d, dy2d) + 4.0 * (dy2d.shape[1] + dy2_2d.shape[1]) * x2d.shape[1]
            return timer._timed("wgrad", fl, by, orig["wgrad1x1_stacked"], x2d, dy2d, dy2_2d, out, workspace, **kw)

        def conv_forward_stats_acc(x, w, partial):
            # pass 1 of the never-stored forward: the convolution repeated for its statistics - work this design adds (0 FLOPs
            # charged), bytes: the narrow input and the weights
            return timer._timed("fwd", 0.0, nbytes(x, w), orig["conv_forward_stats_acc"], x, w, partial)

        def conv_forward_bn_relu2(x, w, out, stats, relu_bits, **kw):
            # pass 2: THE convolution (charged once), BN + residual + ReLU in its epilogue: input, weights, residual, block output
            fl = 2.0 * x.shape[0] * x.shape[1] * x.shape[2] * x.shape[3] * w.shape[0]
            return timer._timed("fwd", fl, nbytes(x, w, kw.get("res"), out), orig["conv_forward_bn_relu2"], x, w, out, stats, relu_bits, **kw)

        def conv_forward_bnstats_pro(x_raw, x_stats, act_out, act_bits, w, out, partial, **kw):
            # conv3 (out given: charged its FLOPs) or its statistics pass (out None: 0) with bn2 + ReLU in the operand path: reads the raw
            # conv2 output, writes the activation as a by-product (the bn_apply pass it replaces is taken off streaming_pass_bytes)
            fl = 0.0 if out is None else 2.0 * x_raw.shape[0] * x_raw.shape[1] * x_raw.shape[2] * x_raw.shape[3] * w.shape[0]
            return timer._timed("fwd", fl, nbytes(x_raw, w, act_out, out), orig["conv_forward_bnstats_pro"], x_raw, x_stats, act_out, act_bits,
                                w, out, partial, **kw)

        ops.conv_forward_bnstats_pro = conv_forward_bnstats_pro
        ops.conv_forward_stats_acc, ops.conv_forward_bn_relu2 = conv_forward_stats_acc, conv_forward_bn_relu2
        ops.conv_dgrad_masksum_rx = conv_dgrad_masksum_rx
        ops.conv_dgrad_masksum_rx_pg = conv_dgrad_masksum_rx_pg
        ops.wgrad1x1_stacked = wgrad1x1_stacked
        ops.conv_forward, ops.conv_dgrad, ops.conv_wgrad = conv_forward, conv_dgrad, conv_wgrad
        ops.conv_dgrad_masksum, ops.conv_dgrad2_bnbwd = conv_dgrad_masksum, conv_dgrad2_bnbwd
        ops.conv_forward_bnstats = conv_forward_bnstats
        ops.conv_dgrad_bnbwd = conv_dgrad_bnbwd
        self._orig, self._ops = orig, ops

    def unwrap(self):
        for k, v in self._orig.items():
            setattr(self._ops, k, v)

    def _timed(self, kind, flops, ideal_bytes, fn, *a, **kw):
        if not self.active:
            return fn(*a, **kw)
        self.shapes.append("%s %s x %s k%s s%s" % (kind, tuple(a[0].shape), tuple(a[1].shape), a[2] if isinstance(a[2], int) else 1,
                                                   a[4] if len(a) > 4 and isinstance(a[4], int) else 1))
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn(*a, **kw)
        e1.record()
        self.records.append((kind, flops, ideal_bytes, e0, e1))
        return out

    def per_shape(self):
        agg = {}
        for (kind, fl, _by, e0, e1), sh in zip(self.records, self.shapes):
            a = agg.setdefault(sh, [0, 0.0, 0.0])
            a[0] += 1; a[1] += e0.elapsed_time(e1); a[2] += fl
        return agg

    def summary(self):
        tot_ms, tot_fl, by = 0.0, 0.0, {}
        self.sol_ms = 0.0      # sum over launches of max(MFMA time at peak, HBM time at peak)
        for kind, fl, nb, e0, e1 in self.records:
            ms = e0.elapsed_time(e1)
            tot_ms += ms; tot_fl += fl
            self.sol_ms += 1e3 * max(fl / (MFMA_BF16_PEAK_TFLOPS * 1e12), nb / (HBM_PEAK_GBS * 1e9))
            k = by.setdefault(kind, [0, 0.0, 0.0, 0.0])
            k[0] += 1; k[1] += ms; k[2] += fl; k[3] += nb
        return tot_ms, tot_fl, by


def streaming_pass_bytes(net, batch, image):
    """Algorithmic HBM bytes per step of everything that is NOT a convolution launch, for the dataflow this design
    cannot go below: per conv+BN unit with an output of E elements (s = element size)
        forward  normalise (+ReLU):   read y, write a                     2 s E   (batch statistics ride on the conv epilogue: 0)
                                                                                  (round 6, never-stored conv3 units: 0 - the normalise sits in the
                                                                                   convolution's epilogue; their residual read is charged there)
        backward normalise:           read g, read y, write dy            3 s E   (the two BN-backward sums ride on the dgrad epilogue: 0;
                                                                                   units routed through the algebraic BN3 backward: 0 + a column-sum
                                                                                   pass over the unit's input)
        block outputs:                + read the residual                  1 s E
    plus the stem's max-pool (forward: the stem's activation is never stored, only the pooled tensor is written; backward:
    the pooled gradient and the index bytes are read inside the stem's backward normalise), the global average pool, SGD = 20 B / parameter (read p, g, m; write p, m)
    and the fused IIF loss (B*C*(4+4) + 8B + 4C + 4).  ReLU decisions are 1 bit per element (s E / 16 bytes, written once,
    read once).  Returns (bytes, formula text)."""
    plan = net._plan(batch, image, image)
    s = 2 if plan.dt == torch.bfloat16 else 4
    total = 0.0
    alg3 = getattr(plan, "alg3_units", set())
    for u in plan.units:
        e = float(u.n * u.ho * u.wo * u.conv.cout)
        total += (2 + 3) * s * e + 2 * e / 8.0
        if u in alg3:
            # BN3 backward by algebra: no backward normalise pass; one column-sum pass over the unit's input
            # (its Gram matrix and the second K source of the data gradient are convolution launches, counted there)
            total += -3 * s * e + s * float(u.n * u.ho * u.wo * u.conv.cin)
    nostore = getattr(plan, "nostore_units", set())
    for u in nostore:
        # round 6: conv3's raw output is never stored - no forward normalise pass (BN + residual + ReLU sit in the convolution's
        # epilogue, whose launch is charged the residual and the block output; its statistics pass is charged its input)
        total -= 2 * s * float(u.n * u.ho * u.wo * u.conv.cout)
    for u3, pro in getattr(plan, "pro_units", {}).items():
        # round 6: bn2's forward normalise runs inside conv3's launch (charged there: raw input read, activation written)
        u2 = pro[0]
        total -= 2 * s * float(u2.n * u2.ho * u2.wo * u2.conv.cout)
        if pro[1] is not None:
            total -= s * float(u3.n * u3.ho * u3.wo * u3.conv.cin)             # and so do the column sums of a2 the algebra route needs
    for b in plan.blocks:
        last = b["units"][-1]
        if last in nostore:
            continue
        total += s * float(last.n * last.ho * last.wo * last.conv.cout)        # residual read of the block-end normalise
    if net.style == "imagenet":
        u = plan.stem
        e_pool = float(plan.pool_out.numel())
        e_stem = float(u.n * u.ho * u.wo * u.conv.cout)
        # fused bn1+relu+maxpool: the generic unit terms above charged write-a / read-a-side passes the fusion removes
        total += s * e_pool * 2 + e_pool          # pooled tensor written + index byte; pooled gradient read
        total -= s * e_stem                       # the activation a is never written
        if plan.dt == torch.bfloat16 and getattr(plan, "pool_fused", False) and not os.environ.get("IIF_NO_POOL_BWD_FUSED"):
            # round 5: the pool backward is gathered inside the stem's backward normalise (iif_bn_backward_pool_fused): that pass
            # reads the pooled gradient (charged above) and the index bytes instead of a gradient at the stem's resolution
            total += e_pool - s * e_stem
    total += 2 * s * float(plan.final.numel())    # global average pool forward read + backward write
    params = float(sum(p.numel() for p in net.parameters()))
    total += 20.0 * params
    C = net.num_classes
    total += batch * C * 8.0 + 8.0 * batch + 4.0 * C + 4.0
    return total, ("sum over conv+BN units of (2+3)*s*E + E/4 bits, + s*E residual per block, max-pool and average pool passes, "
                   "20 B/parameter SGD, fused IIF loss B*C*8; s = %d" % s)


def profiled_traffic():
    """HBM bytes per launch of the convolution family from the committed rocprofv3 PMC passes
    (profiles/r<round>_*_pmc_hbm_traffic.csv, the newest round: FETCH_SIZE x2 + WRITE_SIZE of the same bench command, collected in
    separate --pmc runs; counters cannot be read from inside the timed run).  None if no summary is there."""
    import csv
    import glob
    import re

    def order(path):       # newest = highest round number, then the letter / text after it ("r10_a" > "r2_z")
        m = re.match(r"r(\d+)_(.*)", os.path.basename(path))
        return (int(m.group(1)), m.group(2)) if m else (-1, os.path.basename(path))
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic.csv")), key=order)
    if not files:
        return None
    mb, launches, step_mb = 0.0, 0.0, 0.0
    with open(files[-1]) as f:
        for row in csv.reader(f):
            if len(row) < 5 or row[0] in ("kernel", "TOTAL"):
                continue
            try:
                step_mb += float(row[4])
            except ValueError:
                continue
            if any(t in row[0] for t in ("conv_igemm", "conv_wgrad", "conv3x3_", "conv4x4_", "gemm1x1_", "wgrad1x1_", "stem4x4")):
                mb += float(row[4]); launches += float(row[1])
    if launches <= 0:
        return None
    return {"MB_per_launch": round(mb / launches, 1), "GB_per_step": round(mb / 1e3, 2),
            "all_kernels_GB_per_step": round(step_mb / 1e3, 2), "source": os.path.relpath(files[-1], ROOT)}


def cpu_baseline(counts, sample_bs, steps, device=None):
    """The CPU oracle (torch-CPU fp32 restatement of the reference training step)
    on a bounded sample: same model, same loss, same SGD, `sample_bs` images."""
    from oracle import iif_oracle as O
    from oracle import resnet_oracle as R
    # threads = this process's CPU share (a 1-GPU box gives 16 cores; os.cpu_count() reports the whole host)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, int(os.environ.get("IIF_CPU_THREADS", "16"))))
    torch.set_num_threads(cores)
    sd = R.init_imagenet("resnet50", len(counts), seed=0)
    # Conditioned initialisation for the loss comparison: the last BN gain of every bottleneck x0.1 (trained networks have
    # small residual gains; torchvision's zero_init_residual sets them to 0).  At the plain random init the REFERENCE's own
    # fp32 run is 7e-3..3e-1 of the loss away from its float64 run after one SGD step (profiles/r2_reference_fp32_noise.txt),
    # so a 1e-4 loss-curve statement is only meaningful on a recipe where fp32 itself is reproducible (here: <= 1e-5,
    # tests/golden/g16_nets_conditioned.npz).  The CPU throughput does not depend on the weight values.
    for k in sd:
        if k.startswith("layer") and k.endswith("bn3.weight"):
            sd[k] = sd[k] * 0.1
    g = torch.Generator().manual_seed(0)
    x = torch.randn(sample_bs, 3, 224, 224, generator=g)
    prior = torch.tensor(counts, dtype=torch.float64)
    y = torch.multinomial(prior / prior.sum(), sample_bs, replacement=True, generator=g)
    table = O.iif_tables(counts)["raw"]
    bufs = {}
    sd0 = {k: v.clone() for k, v in sd.items()}
    losses = [float(R.train_step(sd, bufs, x, y, table, "resnet50", 1e-4)[0])]            # warm-up
    t0 = time.time()
    for it in range(steps):
        losses.append(float(R.train_step(sd, bufs, x, y, table, "resnet50", 1e-4)[0]))
    dt = time.time() - t0
    out = {"value": round(sample_bs * steps / dt, 2), "unit": "images/sec", "cores": cores, "kind": "port",
           "sample": "oracle.resnet_oracle.train_step, ResNet50+IIF(raw) fp32, bs=%d 224x224, %d steps after 1 warm-up "
                     "(%.1f s)" % (sample_bs, steps, dt)}
    # loss delta vs CPU on identical inputs: the same initial weights, batch and SGD steps through the HIP path
    if device is not None:
        from iif_amd import resnet_pytorch
        from iif_amd.custom import IIFLoss
        crit = IIFLoss(_Counts(counts), variant="raw", reduction="mean", device=device)
        for name, dt_ in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
            net = resnet_pytorch.resnet50(num_classes=len(counts), use_norm="None", pretrained="None", device=device, compute_dtype=dt_)
            net.load_state_dict(sd0)
            net.train()
            xd, yd = x.to(device), y.to(device)
            deltas = []
            for it in range(steps + 1):
                loss, _ = net.loss_and_backward(xd, yd, crit)
                net.sgd_step(1e-4, 0.9, 1e-4)
                deltas.append(abs(float(loss) - losses[it]) / max(abs(losses[it]), 1e-12))
            out["loss_rel_delta_first_step_" + name] = float("%.3g" % deltas[0])
            out["loss_max_rel_delta_" + name] = float("%.3g" % max(deltas))
            del net
        out["loss_steps_compared"] = steps + 1
        out["loss_note"] = ("same weights, batch and SGD steps (lr 1e-4) through the HIP path, on the conditioned initialisation (last BN gain of "
                            "every bottleneck x0.1) where fp32 itself is reproducible; at plain random init the reference's own fp32 run is "
                            "7e-3..3e-1 from its float64 run after one step (profiles/r2_reference_fp32_noise.txt); the same curves against "
                            "the reference's stored numbers are asserted to 1e-4 in tests/test_resnet_gpu.py::test_hip_step_against_reference_fixture")
        out["cpu_losses"] = [round(v, 4) for v in losses]
    return out


def _self_launch(args):
    """Parent of a multi-GPU run started without torchrun: checks that the GPUs are there, spawns
    ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py <same arguments>`` as a child process, passes its
    output through and returns its exit code (classification/README.md:32 launches the reference the same way)."""
    import socket
    import subprocess
    n = args.gpus
    on_one = args.backend == "gloo" and args.device_index is not None           # rehearsal: N ranks on one GPU over gloo
    visible = torch.cuda.device_count()
    if not on_one and visible < n:
        print("bench.py: --gpus %d needs %d GPUs, %d visible (one process per GPU over RCCL; nothing was measured)"
              % (n, n, visible), file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] launching %d ranks: %s" % (n, " ".join(cmd[1:9])), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch")
    ap.add_argument("--model", default="resnet50")
    ap.add_argument("--image", type=int, default=224)
    ap.add_argument("--classes", type=int, default=1000)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fp32-step", action="store_true", help="skip the fp32 (parity mode) step timing")
    ap.add_argument("--cpu-batch", type=int, default=32)
    ap.add_argument("--cpu-steps", type=int, default=8)
    ap.add_argument("--no-kernel-events", action="store_true", help="do not bracket conv launches with HIP events")
    ap.add_argument("--per-shape", action="store_true", help="print the per-shape conv table to stderr")
    ap.add_argument("--event-every", type=int, default=20, help="bracket the conv launches of every n-th timed step")
    ap.add_argument("--trace-loss", action="store_true", help="record the loss of every step (one tiny copy per step)")
    ap.add_argument("--force-reducer", action="store_true", help="drive the bucketed all-reduce path even with one rank")
    ap.add_argument("--reducer-cus", type=int, default=None,
                    help="compute units the persistent kernels may take while the reducer is active (iif_set_cu_budget; default: "
                         "240 with more than one rank, no reservation with one)")
    ap.add_argument("--reduce-mode", default="allreduce", choices=["allreduce", "rs_ag"],
                    help="gradient buckets: one all_reduce each, or reduce_scatter + all_gather (iif_amd.ddp)")
    ap.add_argument("--bf16-buckets", action="store_true",
                    help="reduce bf16 copies of the gradient buckets (only if the probe on the first gradients stays in tolerance)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the measured path); gloo only to rehearse N ranks on one GPU")
    ap.add_argument("--device-index", type=int, default=None, help="put every rank on this GPU (rehearsal only)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` as the driver types it: start the N ranks as a CHILD torch.distributed.run and relay rank
        # 0's JSON line.  Always a child process (subprocess), never an exec: counting the devices may already have initialised
        # the HIP runtime in this process (torch.cuda.device_count() falls back to hipGetDeviceCount on builds without amdsmi),
        # and a process that has touched the GPU must not be replaced on this pool.
        sys.exit(_self_launch(args))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0")) if args.device_index is None else args.device_index
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1 or (args.force_reducer and "RANK" in os.environ):
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)   # RCCL over xGMI
        else:
            dist.init_process_group("gloo")
    if args.gpus != world:
        # a scaling run that silently measured another world size would poison SCALE_r*.json
        if args.backend == "nccl" and (args.gpus > 1 or world > 1):
            raise SystemExit("bench.py: --gpus %d but WORLD_SIZE %d under nccl: launch with torch.distributed.run "
                             "--nproc-per-node %d" % (args.gpus, world, args.gpus))
        if rank == 0:
            print("note: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)
    if dist.is_initialized() and dist.get_world_size() != world:
        raise SystemExit("bench.py: process group has %d ranks, WORLD_SIZE says %d" % (dist.get_world_size(), world))

    from iif_amd import ops, resnet_pytorch
    from iif_amd.custom import IIFLoss
    from iif_amd.ddp import broadcast_parameters

    C = args.classes
    if C == 100:
        # config 1: the reference's CIFAR100-LT profile (imbalanced_dataset.py:23-37: exp, img_max 500, imb 0.01 -> 500 ... 5, 10 847 images)
        from iif_amd.imbalanced_dataset import img_num_per_cls
        counts = img_num_per_cls(100, 50000, "exp", 0.01)
    else:
        counts = lt_counts(C, 1280 if C == 1000 else 4980)   # ImageNet-LT / Places-LT shaped profiles
    cdt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    torch.manual_seed(0)
    if hasattr(resnet_pytorch, args.model):
        net = getattr(resnet_pytorch, args.model)(num_classes=C, use_norm="None", pretrained="None", device=dev,
                                                  compute_dtype=cdt)
    else:                                       # CIFAR-style nets (BASELINE config 0: resnet32, 32x32, C=100)
        from iif_amd import resnet_cifar
        net = getattr(resnet_cifar, args.model)(num_classes=C, use_norm="None", device=dev, compute_dtype=cdt)
    net.train()
    broadcast_parameters(net)
    crit = IIFLoss(_Counts(counts), variant="raw", reduction="mean", device=dev)
    reducer = net.make_reducer(mode=args.reduce_mode, cu_budget=args.reducer_cus) if (world > 1 or args.force_reducer) else None
    if reducer is not None and args.force_reducer:
        reducer.force = True
    tail_events = []
    if reducer is not None:
        # exposed tail: how long the compute stream sits in reducer.finish() waiting for the last buckets
        _finish = reducer.finish

        def timed_finish():
            if not measuring[0]:
                return _finish()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); _finish(); e1.record()
            tail_events.append((e0, e1))
        reducer.finish = timed_finish
    measuring = [False]
    g = torch.Generator().manual_seed(1234 + rank)
    B = args.batch
    x = torch.randn(B, 3, args.image, args.image, generator=g).to(dev)
    prior = torch.tensor(counts, dtype=torch.float64)
    y = torch.multinomial(prior / prior.sum(), B, replacement=True, generator=g).to(dev)
    scale = reducer.grad_scale if reducer is not None else 1.0

    trace = []

    def step(it):
        lr = 0.1 * (1e-3 * (1 - min(it, 1000) / 1000.0) + min(it, 1000) / 1000.0)      # train.py:52-56 warm-up
        loss, _ = net.loss_and_backward(x, y, crit, reducer=reducer)
        net.sgd_step(lr, 0.9, 1e-4, grad_scale=scale)
        if args.trace_loss:
            trace.append(loss.clone())
        return loss

    def note(msg):
        if rank == 0:
            print("[bench] " + msg, file=sys.stderr, flush=True)

    note("model built (%d params), starting %d warm-up steps" % (sum(p.numel() for p in net.parameters()), args.warmup))
    bf16_probe = None
    for it in range(args.warmup):
        if it == 0 and reducer is not None and args.bf16_buckets:
            # gradients of the first step stay local: measure what bf16 buckets would do to their average, switch only if
            # that stays within one bf16 rounding, then reduce this step un-overlapped (as iif_amd.train does)
            net.loss_and_backward(x, y, crit, reducer=None)
            bf16_probe = reducer.probe_bf16()
            try:
                reducer.set_bucket_dtype(torch.bfloat16)
            except RuntimeError:
                note("bf16 buckets refused by the probe (%.2e): staying with fp32" % bf16_probe)
            reducer.begin(); reducer.finish()
            net.sgd_step(1e-4, 0.9, 1e-4, grad_scale=scale)
        else:
            step(it)
        if it == 0:
            torch.cuda.synchronize()
            note("first step done")
    timer = None
    if not args.no_kernel_events:
        timer = ConvTimer()
        timer.wrap(ops)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    measuring[0] = True
    t0 = time.perf_counter()
    for it in range(args.steps):
        if timer is not None:
            timer.active = it % max(args.event_every, 1) == 0
            timer.sampled_steps += 1 if timer.active else 0
        loss = step(args.warmup + it)
    t_enq = time.perf_counter() - t0          # the host is done enqueueing (close to dt: the step is bound by the host's launch rate)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if timer is not None:
        timer.unwrap()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    final_loss = float(loss.item())
    note("timed region: %.3f s for %d steps" % (dt, args.steps))
    if trace:
        note("loss trace: " + " ".join("%.4f" % t.item() for t in trace))

    if rank == 0:
        out = {
            "metric": ("images/sec ResNet50+IIF ImageNet-LT bs=256/GPU" if (args.model, B, C, args.image) == ("resnet50", 256, 1000, 224)
                       else "images/sec %s+IIF C=%d %dx%d bs=%d/GPU" % (args.model, C, args.image, args.image, B)),
            "value": round(B * world * args.steps / dt, 2),
            "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1000.0 * dt / args.steps, 3),
            "host_enqueue_ms_per_step": round(1000.0 * t_enq / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if cdt == torch.bfloat16 else "f32", "data": "synthetic",
            "config": {"workload": "%s + IIF(raw) training step, synthetic long-tailed %dx%d, C=%d (counts %d..5), "
                                   "bs=%d/GPU, SGD momentum 0.9 wd 1e-4 with warm-up, random init" % (args.model, args.image, args.image, C, counts[0], B),
                       "global_batch": B * world, "parallelism": "dp%d" % world, "final_loss": round(final_loss, 4)},
        }
        if reducer is not None:
            # what the collective library was asked to do (proof that `world` ranks reduced, and how)
            rd = reducer.describe()
            rd["backend"] = args.backend + (" (RCCL over xGMI)" if args.backend == "nccl" else "")
            rd["exposed_tail_ms_per_step"] = round(sum(a.elapsed_time(b) for a, b in tail_events) / max(len(tail_events), 1), 4)
            if bf16_probe is not None:
                rd["bf16_probe_rel_l2"] = bf16_probe
            out["reducer"] = rd
        nsamp = max(timer.sampled_steps, 1) if timer is not None else 1
        if timer is not None and args.per_shape:
            for sh, (cnt, ms, fl) in sorted(timer.per_shape().items(), key=lambda kv: -kv[1][1]):
                print("[conv] %-70s n=%3d  %8.3f ms/step  %7.1f TFLOP/s" % (sh, cnt // nsamp, ms / nsamp, fl / (ms * 1e-3) / 1e12), file=sys.stderr)
        if timer is not None:
            tot_ms, tot_fl, by = timer.summary()
            nl = len(timer.records)
            headline = (args.model, B, C, args.image, args.dtype) == ("resnet50", 256, 1000, 224, "bf16")
            traffic = profiled_traffic() if headline else None
            step_ms = out["ms_per_step"]
            gflop_step = tot_fl / nsamp / 1e9
            conv_bytes = sum(v[3] for v in by.values()) / nsamp           # every operand / result of a launch exactly once
            pass_bytes, formula = streaming_pass_bytes(net, B, args.image)
            alg_bytes = conv_bytes + pass_bytes
            ach_tf = gflop_step / step_ms                                   # GFLOP / ms = TFLOP/s
            ach_gbs = alg_bytes / (step_ms * 1e-3) / 1e9
            out["roofline"] = {
                # the matrix roof north_star names: algorithmic convolution FLOPs of one step over THIS run's step time
                "bound": "mfma", "kernel": "whole training step (implicit-GEMM convolution family carries the FLOPs)",
                "achieved": round(ach_tf, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach_tf / MFMA_BF16_PEAK_TFLOPS, 4),
                "algorithmic_gflop_per_step": round(gflop_step, 1),
                "t_mfma_ms_per_step": round(gflop_step / MFMA_BF16_PEAK_TFLOPS, 3),
                # measured HBM bytes of one step (committed PMC passes of the same command; null off the profiled config)
                "traffic": round(traffic["all_kernels_GB_per_step"] * 1e9) if traffic else None,
                "traffic_unit": "bytes/step",
                "traffic_source": traffic["source"] if traffic else None,
                # the roof that BINDS this step: algorithmic bytes (formula below) over the same step time
                "hbm": {"bound": "hbm", "achieved": round(ach_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(ach_gbs / HBM_PEAK_GBS, 4),
                        "algorithmic_GB_per_step": round(alg_bytes / 1e9, 2),
                        "conv_operands_once_GB": round(conv_bytes / 1e9, 2),
                        "streaming_passes_GB": round(pass_bytes / 1e9, 2),
                        "t_hbm_ms_per_step": round(alg_bytes / (HBM_PEAK_GBS * 1e9) * 1e3, 3),
                        "formula": "conv: bytes of src + weights + dst (+ residual, + upstream x of a fused BN-backward sum) per launch, "
                                   "each once; passes: " + formula,
                        "measured_over_algorithmic": round(traffic["all_kernels_GB_per_step"] * 1e9 / alg_bytes, 3) if traffic else None},
                "binding": "hbm",
                # HIP-event brackets per launch kind.  The three streams of a step run CONCURRENTLY, so these durations overlap
                # and their sum may exceed ms_per_step: they rank kernels, they are not a share of the step.
                "brackets": {
                    "note": "overlapped, sums across 3 streams; not comparable with ms_per_step",
                    "bracketed_steps": timer.sampled_steps,
                    "launches_per_step": nl // nsamp,
                    "avg_launch_us": round(1000.0 * tot_ms / max(nl, 1), 2),
                    "sum_ms_per_step_overlapped": round(tot_ms / nsamp, 3),
                    "by_kind_ms_per_step_overlapped": {k: round(v[1] / nsamp, 3) for k, v in by.items()},
                    "by_kind_tflops_in_bracket": {k: round(v[2] / (v[1] * 1e-3) / 1e12, 1) for k, v in by.items() if v[1] > 0},
                    "by_kind_ideal_GB_per_step": {k: round(v[3] / nsamp / 1e9, 2) for k, v in by.items()},
                    "conv_family_profiled_GB_per_step": traffic["GB_per_step"] if traffic else None,
                },
            }
        if world == 1 and not args.no_fp32_step and args.dtype == "bf16" and hasattr(resnet_pytorch, args.model):
            # the parity mode (fp32 storage, exact-fp32 MFMA chains: the mode whose loss curve meets north_star's 1e-4) has a
            # number too: same batch, same step, outside the timed region
            del net
            torch.cuda.empty_cache()
            net32 = getattr(resnet_pytorch, args.model)(num_classes=C, use_norm="None", pretrained="None", device=dev,
                                                        compute_dtype=torch.float32)
            net32.train()
            for _ in range(2):
                net32.loss_and_backward(x, y, crit)
                net32.sgd_step(1e-4, 0.9, 1e-4)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            n32 = 3
            for _ in range(n32):
                net32.loss_and_backward(x, y, crit)
                net32.sgd_step(1e-4, 0.9, 1e-4)
            torch.cuda.synchronize()
            out["fp32_ms_per_step"] = round(1000.0 * (time.perf_counter() - t1) / n32, 3)
            out["fp32_note"] = "same step in fp32 storage / exact fp32 MFMA (v_mfma_f32_16x16x4_f32), %d steps after 2 warm-up, not the headline" % n32
            del net32
            torch.cuda.empty_cache()
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(counts, args.cpu_batch, args.cpu_steps, dev if args.model == "resnet50" else None)
        print(json.dumps(out))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
