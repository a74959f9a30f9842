"""Microbenchmark of the fused 1x1 conv pair (sininn_conv_pair_k1) against the two-launch path at the layer shapes of
BASELINE configs[1] (random packed weights: timing only).  `python tools/bench_pair.py [--reps 30] [--size 256]`."""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sin_inn_amd                                   # noqa: E402,F401
from sin_inn_amd import _lib, ops                    # noqa: E402
from bench_kernels import timeit                     # noqa: E402


def args_of(**kw):
    a = _lib.ConvArgs()
    for k, v in kw.items():
        setattr(a, 'inp' if k == 'in_' else k, v)
    return a


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=30)
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--size', type=int, default=256)
    ap.add_argument('--ablate', type=int, default=0, help='diagnostic: 1 weights read from one address (all L1 hits), 2 no hidden store')
    ap.add_argument('--phases', action='store_true', help='in-kernel phase stamps (shader clock) of the pair kernel')
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    lib = _lib.lib()
    lib.sininn_conv_test_hooks(a.ablate * 1000, 0)
    st = ops._stream
    b = a.batch
    for level, (c, hw) in enumerate(((48, a.size // 4), (192, a.size // 8))):
        co = c // 2
        m = b * hw * hw
        x = torch.randn(m, c, device=dev)
        hid = torch.empty(m, 256, device=dev)
        w1 = torch.randn(256 * co, device=dev) * 0.05; b1 = torch.randn(256, device=dev) * 0.1
        w2 = torch.randn(2 * co * 256, device=dev) * 0.02; b2 = torch.randn(2 * co, device=dev) * 0.1
        out = torch.empty(m, c, device=dev); sb = torch.empty(m, co, device=dev); ld = torch.zeros(b, device=dev)
        common = dict(B=b, H=hw, W=hw, ksize=1)
        f = args_of(in_=ops.ptr(x, co), in_stride=c, Cin=co, w=ops.ptr(w1), bias=ops.ptr(b1), Np=256, mode=_lib.CONV_RELU,
                    out=ops.ptr(hid), out_stride=256, N=256, **common)
        s = args_of(in_=ops.ptr(hid), in_stride=256, Cin=256, w=ops.ptr(w2), bias=ops.ptr(b2), Np=2 * co, mode=_lib.CONV_COUPLE_FWD,
                    out=ops.ptr(out), out_stride=c, v=ops.ptr(x), v_stride=c, sbuf=ops.ptr(sb), logdet=ops.ptr(ld), Co=co,
                    clamp=1.2, col_tile=ops.coupling_tile(co), **common)
        assert lib.sininn_conv_pair_k1_supported(C.byref(f), C.byref(s))
        fl = 2.0 * m * 256 * (co + 2 * co)
        t2 = timeit(lambda: (_lib.check(lib.sininn_conv(C.byref(f), st())), _lib.check(lib.sininn_conv(C.byref(s), st()))), a.reps)
        t1 = timeit(lambda: _lib.check(lib.sininn_conv_pair_k1(C.byref(f), C.byref(s), st())), a.reps)
        f0 = args_of(in_=ops.ptr(x, co), in_stride=c, Cin=co, w=ops.ptr(w1), bias=ops.ptr(b1), Np=256, mode=_lib.CONV_RELU,
                     out_stride=256, N=256, **common)
        t0 = timeit(lambda: _lib.check(lib.sininn_conv_pair_k1(C.byref(f0), C.byref(s), st())), a.reps)
        if a.phases:
            stamp = torch.zeros(8, dtype=torch.int64, device=dev)
            s.stamp = stamp.data_ptr()
            _lib.check(lib.sininn_conv_pair_k1(C.byref(f), C.byref(s), st()))
            torch.cuda.synchronize()
            v = stamp.cpu().tolist()
            names = ['input staging', 'GEMM 1', 'activation pass + HBM store', 'GEMM 2', 'T + epilogue']
            print(f'  L{level} forward phases per block (shader clocks, {v[7]} blocks, total {v[6] / v[7]:.0f}): ' +
                  ', '.join(f'{n} {v[i] / v[7]:.0f}' for i, n in enumerate(names)))
            s.stamp = None
        print(f'L{level} forward  {co}->256->{2 * co}: two launches {t2 * 1e3:7.1f} us | pair {t1 * 1e3:7.1f} us ({fl / t1 / 1e9:6.1f} TF/s) | '
              f'pair, hidden not stored {t0 * 1e3:7.1f} us')
        # backward pair: dh = (dr W2) . mask -> dx = dh W1 (+ addend)
        dr = torch.randn(m, 2 * co, device=dev); dh = torch.empty(m, 256, device=dev)
        w2d = torch.randn(256 * 2 * co, device=dev) * 0.02
        npd = ops.pad16(co)
        w1d = torch.randn(npd * 256, device=dev) * 0.05
        dx = torch.zeros(m, c, device=dev); ad = torch.randn(m, c, device=dev)
        f = args_of(in_=ops.ptr(dr), in_stride=2 * co, Cin=2 * co, w=ops.ptr(w2d), Np=256, mode=_lib.CONV_MASK, out=ops.ptr(dh),
                    out_stride=256, N=256, mask=ops.ptr(hid), mask_stride=256, **common)
        s = args_of(in_=ops.ptr(dh), in_stride=256, Cin=256, w=ops.ptr(w1d), Np=npd, mode=_lib.CONV_ADD, out=ops.ptr(dx),
                    out_stride=c, N=co, addend=ops.ptr(ad), addend_stride=c, **common)
        assert lib.sininn_conv_pair_k1_supported(C.byref(f), C.byref(s))
        t2 = timeit(lambda: (_lib.check(lib.sininn_conv(C.byref(f), st())), _lib.check(lib.sininn_conv(C.byref(s), st()))), a.reps)
        t1 = timeit(lambda: _lib.check(lib.sininn_conv_pair_k1(C.byref(f), C.byref(s), st())), a.reps)
        print(f'L{level} backward {2 * co}->256->{co}: two launches {t2 * 1e3:7.1f} us | pair {t1 * 1e3:7.1f} us ({fl / t1 / 1e9:6.1f} TF/s)')


if __name__ == '__main__':
    main()
