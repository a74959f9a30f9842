"""Per-kernel microbenchmark at BASELINE config-2 layer shapes (run on the GPU box):
    python tools/bench_kernels.py [--reps 20] [--only conv|wgrad] [--ck CK] [--cfg 0|1|2]
Prints achieved TFLOP/s (algorithmic FLOPs) for every conv / wgrad launch shape of the training step."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sin_inn_amd                                   # noqa: E402,F401
from sin_inn_amd import _lib, ops                    # noqa: E402


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--only', default='')
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--size', type=int, default=256)
    ap.add_argument('--ck', type=int, default=0)
    ap.add_argument('--cfg', type=int, default=0)
    ap.add_argument('--wgrad16', type=int, default=0, help='bit0: 16-wide tiles, bit1: disable the Winograd wgrad')
    ap.add_argument('--wino', type=int, default=0, help='1: run 3x3 convs with 32-multiple columns through the Winograd kernel')
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    _lib.lib().sininn_conv_test_hooks(a.cfg, a.ck)
    _lib.lib().sininn_wgrad_test_hooks(a.wgrad16)
    b = a.batch
    rows = []
    for level, (c, hw) in enumerate(((48, a.size // 4), (192, a.size // 8))):
        co = c // 2
        m = b * hw * hw
        for k in (3, 1):
            taps = k * k
            shapes = [('fwd conv1+relu', co, 256, _lib.CONV_RELU), ('fwd conv2+couple', 256, 2 * co, _lib.CONV_COUPLE_FWD),
                      ('dgrad conv2+mask', 2 * co, 256, _lib.CONV_MASK), ('dgrad conv1+add', 256, co, _lib.CONV_ADD)]
            for name, cin, n, mode in shapes:
                if a.only and a.only != 'conv':
                    continue
                npk = ops.pad16(n)
                wino = a.wino and k == 3
                if wino and mode != _lib.CONV_COUPLE_FWD:
                    npk = ops.pad32(n)
                x = torch.randn(m, cin, device=dev)
                w = torch.randn((16 if wino else taps) * npk * cin, device=dev) * 0.05
                bias = torch.randn(npk, device=dev) * 0.1
                ostr = max(n, 256)
                out = torch.empty(m, ostr, device=dev)
                v = torch.randn(m, c, device=dev)
                kw = dict(in_=ops.ptr(x), in_stride=cin, Cin=cin, w=ops.ptr(w), bias=ops.ptr(bias), Np=npk, B=b, H=hw, W=hw,
                          ksize=k, mode=mode, out=ops.ptr(out), out_stride=ostr, N=n, winograd=1 if wino else 0)
                keep = [x, w, bias, out, v]
                if mode == _lib.CONV_COUPLE_FWD:
                    sb = torch.empty(m, co, device=dev); ld = torch.zeros(b, device=dev)
                    keep += [sb, ld]
                    kw.update(out_stride=c, v=ops.ptr(v), v_stride=c, sbuf=ops.ptr(sb), logdet=ops.ptr(ld), Co=co, clamp=1.2, col_tile=ops.coupling_tile(co))
                if mode == _lib.CONV_MASK:
                    mk = torch.randn(m, 256, device=dev); keep.append(mk)
                    kw.update(mask=ops.ptr(mk), mask_stride=256)
                if mode == _lib.CONV_ADD:
                    ad = torch.randn(m, c, device=dev); keep.append(ad)
                    kw.update(addend=ops.ptr(ad), addend_stride=c)
                ms = timeit(lambda: ops.conv(**kw), a.reps)
                rows.append((f'L{level} {k}x{k} {name} {cin}->{n}', ms, 2.0 * m * taps * cin * n))
            for name, cin, n in (('wgrad conv1', co, 256), ('wgrad conv2', 256, 2 * co)):
                if a.only and a.only != 'wgrad':
                    continue
                x = torch.randn(m, cin, device=dev); g = torch.randn(m, n, device=dev)
                gw = torch.zeros(n, cin, k, k, device=dev); gb = torch.zeros(n, device=dev)
                ms = timeit(lambda: ops.wgrad(x, 0, cin, cin, g, n, n, b, hw, hw, k, gw, gb), a.reps)
                rows.append((f'L{level} {k}x{k} {name} {cin}x{n}', ms, 2.0 * m * taps * cin * n))
    tot_ms = 0
    for name, ms, fl in rows:
        print(f'{name:42s} {ms * 1e3:9.1f} us  {fl / ms / 1e9:8.1f} TFLOP/s')
        tot_ms += ms
    print(f'sum of one launch each: {tot_ms:.3f} ms')


if __name__ == '__main__':
    main()
