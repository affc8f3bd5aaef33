"""Isolated timings of the weight-gradient launches of ResNet50 at bs 256 (every distinct shape), each alone on the GPU and
on COLD operands: the inputs rotate over enough copies that no launch finds its operands in the 256 MiB Infinity Cache.
Prints ms, algorithmic GB/s (x + dy once, dW once) and TFLOP/s; `--check` also compares every result with a float64
contraction of the same bf16 operands on a pixel subsample (exact for the sampled rows... the whole tensor at small sizes).

    python scripts/bm_wgrad1x1.py [--only 1x1] [--reps 10]
"""
import argparse
import sys
sys.path.insert(0, '.')
import torch
from iif_amd import ops

SHAPES = [  # count per step, n, h, cin, cout, k, stride
    (1, 256, 56, 64, 64, 1, 1), (3, 256, 56, 64, 64, 3, 1), (4, 256, 56, 64, 256, 1, 1), (2, 256, 56, 256, 64, 1, 1),
    (1, 256, 56, 256, 128, 1, 1), (1, 256, 56, 128, 128, 3, 2), (4, 256, 28, 128, 512, 1, 1), (1, 256, 56, 256, 512, 1, 2),
    (3, 256, 28, 512, 128, 1, 1), (3, 256, 28, 128, 128, 3, 1), (1, 256, 28, 512, 256, 1, 1), (1, 256, 28, 256, 256, 3, 2),
    (6, 256, 14, 256, 1024, 1, 1), (1, 256, 28, 512, 1024, 1, 2), (5, 256, 14, 1024, 256, 1, 1), (5, 256, 14, 256, 256, 3, 1),
    (1, 256, 14, 1024, 512, 1, 1), (1, 256, 14, 512, 512, 3, 2), (3, 256, 7, 512, 2048, 1, 1), (1, 256, 14, 1024, 2048, 1, 2),
    (2, 256, 7, 2048, 512, 1, 1), (2, 256, 7, 512, 512, 3, 1),
    # Gram matrices of the algebraic BN3 route (x = dy = a2)
    (3, 256, 56, 64, 64, 1, 1), (4, 256, 28, 128, 128, 1, 1), (6, 256, 14, 256, 256, 1, 1),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', default='')
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--check', action='store_true')
    ap.add_argument('--batch', type=int, default=256)
    args = ap.parse_args()
    dev = 'cuda:0'
    ws = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    tot = 0.0
    for (cnt, n, h, cin, cout, k, stride) in SHAPES:
        n = args.batch
        if args.only == '1x1' and k != 1:
            continue
        if args.only == '3x3' and k != 3:
            continue
        pad = k // 2
        ho = (h + 2 * pad - k) // stride + 1
        xb, yb = n * h * h * cin * 2, n * ho * ho * cout * 2
        copies = max(2, int((600 << 20) // (xb + yb)) + 1)
        xs = [torch.randn(n, h, h, cin, device=dev).to(torch.bfloat16) for _ in range(copies)]
        ys = [torch.randn(n, ho, ho, cout, device=dev).to(torch.bfloat16) for _ in range(copies)]
        out = torch.zeros(cout, ((k * k * cin + 15) // 16) * 16, dtype=torch.float32, device=dev)
        for i in range(2):
            ops.conv_wgrad(xs[i % copies], ys[i % copies], k, k, stride, pad, workspace=ws, out=out, ldw=out.shape[1])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(args.reps):
            ops.conv_wgrad(xs[i % copies], ys[i % copies], k, k, stride, pad, workspace=ws, out=out, ldw=out.shape[1])
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.reps
        fl = 2.0 * n * ho * ho * cout * k * k * cin
        by = xb + yb + out.numel() * 4
        err = ''
        if args.check:
            i = (args.reps - 1) % copies
            x, dy = xs[i], ys[i]
            if k == 1:
                xx = x[:, ::stride, ::stride, :].reshape(-1, cin).double()
                ref = dy.reshape(-1, cout).double().t() @ xx
                got = out[:, :cin].double()
                err = '  rel err %.2e' % ((got - ref).norm() / ref.norm()).item()
        tot += cnt * ms
        print("x%d wgrad n%d h%d %4d->%4d k%d s%d: %.3f ms  %6.0f GB/s  %6.1f TFLOP/s%s" % (cnt, n, h, cin, cout, k, stride, ms, by / ms / 1e6,
                                                                                        fl / ms / 1e9, err), flush=True)
        del xs, ys
    print("weighted sum over a step: %.3f ms" % tot)


if __name__ == '__main__':
    main()
