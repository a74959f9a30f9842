"""Per-launch time and phase split of the fused 1x1 subnet backward (sininn_conv_sub1_bwd) at a level-0 shape, beside the
launches it replaces (data-gradient pair + the two weight gradients).   python tools/bench_sub1.py [--b 16 --hw 64]"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--b', type=int, default=16)
    ap.add_argument('--hw', type=int, default=64)
    ap.add_argument('--co', type=int, default=24)
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--no-dx', action='store_true')
    ap.add_argument('--bf16', action='store_true', help='the mixed-precision twins (conv_sub1_bf16.hip) beside the bf16 pair + grouped bf16 weight gradients')
    a = ap.parse_args()
    if a.bf16:
        return main_bf16(a)
    import sin_inn_amd
    from sin_inn_amd import _lib, ops
    lib = _lib.lib()
    dev = torch.device('cuda')
    co, b, h, w = a.co, a.b, a.hw, a.hw
    k1, k2, m = co, 2 * co, b * h * w
    torch.manual_seed(0)
    x = torch.randn(m, 2 * co, device=dev)
    conv1 = torch.nn.Conv2d(k1, 256, 1).to(dev)
    conv2 = torch.nn.Conv2d(256, k2, 1).to(dev)
    pk1 = ops.pack_conv(conv1.weight.detach(), conv1.bias.detach(), None, True)
    pk2 = ops.pack_conv(conv2.weight.detach(), conv2.bias.detach(), ops.coupling_colmap(co, dev), True)
    dr = torch.randn(m, k2, device=dev)
    addend = torch.randn(m, k1, device=dev)
    hid = torch.randn(m, 256, device=dev)
    dh = torch.empty(m, 256, device=dev)
    dx = torch.empty(m, k1, device=dev)
    gw2, gb2 = torch.zeros(k2, 256, 1, 1, device=dev), torch.zeros(k2, device=dev)
    gw1, gb1 = torch.zeros(256, k1, 1, 1, device=dev), torch.zeros(256, device=dev)
    stamps = torch.zeros(32, dtype=torch.int64, device=dev)

    def args(**kw):
        q = _lib.ConvArgs()
        for k, v in kw.items():
            setattr(q, 'inp' if k == 'in_' else k, v)
        return q
    common = dict(B=b, H=h, W=w, ksize=1)
    rc = args(in_=ops.ptr(x, co), in_stride=2 * co, Cin=k1, w=ops.ptr(pk1[0]), bias=ops.ptr(pk1[1]), Np=256, **common)
    d2 = args(in_=ops.ptr(dr), in_stride=k2, Cin=k2, w=ops.ptr(pk2[2]), Np=256, mode=_lib.CONV_MASK, out=ops.ptr(dh), out_stride=256,
              N=256, mask=ops.ptr(hid), mask_stride=256, **common)
    d1 = args(in_=ops.ptr(dh), in_stride=256, Cin=256, w=ops.ptr(pk1[2]), Np=ops.pad16(k1), mode=_lib.CONV_ADD, out=ops.ptr(dx),
              out_stride=k1, N=k1, addend=ops.ptr(addend), addend_stride=k1, **common)
    nbytes = lib.sininn_conv_sub1_bwd_workspace_bytes(k1, co)
    ws = torch.empty(nbytes // 4, device=dev)

    def fused(stamp=False):
        d1.stamp = stamps.data_ptr() if stamp else None
        _lib.check(lib.sininn_conv_sub1_bwd(C.byref(rc), C.byref(d2), C.byref(d1), int(a.no_dx), ops.ptr(gw2), ops.ptr(gb2), ops.ptr(gw1),
                                            ops.ptr(gb1), ops.ptr(ws), nbytes, ops._stream()))
        d1.stamp = None

    def old():
        _lib.check(lib.sininn_conv_pair_k1(C.byref(d2), C.byref(d1), ops._stream()))
        ops.wgrad_group([(hid, 0, 256, 256, dr, 0, k2, k2, gw2, gb2), (x, co, 2 * co, k1, dh, 0, 256, 256, gw1, gb1)], b, h, w, 1)

    def timeit(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.reps * 1e3
    # forward twin: conv1 -> conv2 -> coupling + log-det (sininn_conv_sub1_fwd), with and without its epilogue (ablation hook 8)
    out = torch.empty(m, 2 * co, device=dev); sb = torch.empty(m, co, device=dev); ld = torch.zeros(b, device=dev); y2 = torch.empty(m, co, device=dev)
    f1 = args(in_=ops.ptr(x, co), in_stride=2 * co, Cin=k1, w=ops.ptr(pk1[0]), bias=ops.ptr(pk1[1]), Np=256, mode=_lib.CONV_RELU, out_stride=256,
              N=256, **common)
    f2 = args(in_=ops.ptr(hid), in_stride=256, Cin=256, w=ops.ptr(pk2[0]), bias=ops.ptr(pk2[1]), Np=k2, mode=_lib.CONV_COUPLE_FWD, out=ops.ptr(out),
              out_stride=2 * co, v=ops.ptr(x), v_stride=2 * co, sbuf=ops.ptr(sb), logdet=ops.ptr(ld), Co=co, clamp=1.2, out2=ops.ptr(y2),
              out2_stride=co, col_tile=ops.coupling_tile(co), **common)

    def fwd():
        _lib.check(lib.sininn_conv_sub1_fwd(C.byref(f1), C.byref(f2), ops._stream()))

    def fwd_pair():
        _lib.check(lib.sininn_conv_pair_k1(C.byref(f1), C.byref(f2), ops._stream()))
    print(f'shape: batch {b}, {h}x{w}, Cin {k1}, 2Co {k2}: M = {m}')
    print(f'persistent forward      : {timeit(fwd):8.1f} us')
    print(f'pair forward (no store) : {timeit(fwd_pair):8.1f} us')
    lib.sininn_conv_test_hooks(8000, 0)
    print(f'persistent forward, epilogue skipped : {timeit(fwd):8.1f} us')
    lib.sininn_conv_test_hooks(0, 0)
    print(f'fused backward + reduce : {timeit(fused):8.1f} us')
    print(f'pair + grouped wgrad    : {timeit(old):8.1f} us')
    stamps.zero_()
    fused(True)
    torch.cuda.synchronize()
    st = stamps.cpu().tolist()
    names = ['staging (+ barrier A)', 'stage R', 'stage W2 + db2', 'stage 2 (+ mask RMW)', 'barrier C + stage 3', 'stage W1',
             'barrier D + T + epilogue', 'slab write', 'total']
    tiles = max(st[9], 1)
    names += ['(tiles)', 'store_tile (vmcnt wait + LDS writes)', 'issue next + side loads', 'barrier C', 'T write', 'barrier E']
    print(f'phase clocks per tile ({tiles} tiles):                      wave 0      wave 7')
    for i, n in enumerate(names):
        if i != 9:
            print(f'  {i:2d} {n:40s} {st[i] / tiles:10.0f}  {st[16 + i] / tiles:10.0f}')


def main_bf16(a):
    import sin_inn_amd
    from sin_inn_amd import _lib, ops
    lib = _lib.lib()
    dev = torch.device('cuda')
    co, b, h, w = a.co, a.b, a.hw, a.hw
    k1, k2, m = co, 2 * co, b * h * w
    torch.manual_seed(0)
    bf = torch.bfloat16
    x = torch.randn(m, 2 * co, device=dev)
    conv1 = torch.nn.Conv2d(k1, 256, 1).to(dev)
    conv2 = torch.nn.Conv2d(256, k2, 1).to(dev)
    pk1 = ops.pack_conv_bf16(conv1.weight.detach(), conv1.bias.detach(), None, True)
    pk2 = ops.pack_conv_bf16(conv2.weight.detach(), conv2.bias.detach(), ops.coupling_colmap(co, dev), True)
    dr = torch.randn(m, k2, device=dev)
    addend = torch.randn(m, k1, device=dev)
    hid = torch.randn(m, 256, device=dev).to(bf)
    dh = torch.empty(m, 256, device=dev, dtype=bf)
    dx = torch.empty(m, k1, device=dev)
    gw2, gb2 = torch.zeros(k2, 256, 1, 1, device=dev), torch.zeros(k2, device=dev)
    gw1, gb1 = torch.zeros(256, k1, 1, 1, device=dev), torch.zeros(256, device=dev)

    def args(**kw):
        q = _lib.ConvArgs()
        for k, v in kw.items():
            setattr(q, 'inp' if k == 'in_' else k, v)
        return q
    common = dict(B=b, H=h, W=w, ksize=1, w_bf16=1)
    pb = lambda t: ops.ptr(t, dtype=bf)
    rc = args(in_=ops.ptr(x, co), in_stride=2 * co, Cin=k1, w=pb(pk1[0]), bias=ops.ptr(pk1[1]), Np=256, **common)
    d2 = args(in_=ops.ptr(dr), in_stride=k2, Cin=k2, w=pb(pk2[2]), Np=256, mode=_lib.CONV_MASK, out=pb(dh), out_stride=256, N=256,
              mask=pb(hid), mask_stride=256, out_bf16=1, mask_bf16=1, **common)
    d1 = args(in_=pb(dh), in_stride=256, Cin=256, w=pb(pk1[2]), Np=ops.pad16(k1), mode=_lib.CONV_ADD, out=ops.ptr(dx), out_stride=k1, N=k1,
              addend=ops.ptr(addend), addend_stride=k1, in_bf16=1, **common)
    nbytes = lib.sininn_conv_sub1_bwd_workspace_bytes(k1, co)
    ws = torch.empty(nbytes // 4, device=dev)

    def fused():
        _lib.check(lib.sininn_conv_sub1_bwd(C.byref(rc), C.byref(d2), C.byref(d1), int(a.no_dx), ops.ptr(gw2), ops.ptr(gb2), ops.ptr(gw1),
                                            ops.ptr(gb1), ops.ptr(ws), nbytes, ops._stream()))

    def old():
        _lib.check(lib.sininn_conv_pair_k1(C.byref(d2), C.byref(d1), ops._stream()))
        ops.wgrad_group([(hid, 0, 256, 256, dr, 0, k2, k2, gw2, gb2, True, False), (x, co, 2 * co, k1, dh, 0, 256, 256, gw1, gb1, False, True)], b, h, w, 1)

    def timeit(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.reps * 1e3
    out = torch.empty(m, 2 * co, device=dev); sb = torch.empty(m, co, device=dev); ld = torch.zeros(b, device=dev); y2 = torch.empty(m, co, device=dev)
    hsave = torch.empty(m, 256, device=dev, dtype=bf)
    f1 = args(in_=ops.ptr(x, co), in_stride=2 * co, Cin=k1, w=pb(pk1[0]), bias=ops.ptr(pk1[1]), Np=256, mode=_lib.CONV_RELU, out_stride=256,
              N=256, out_bf16=1, **common)
    f2 = args(in_=pb(hsave), in_stride=256, Cin=256, w=pb(pk2[0]), bias=ops.ptr(pk2[1]), Np=k2, mode=_lib.CONV_COUPLE_FWD, out=ops.ptr(out),
              out_stride=2 * co, v=ops.ptr(x), v_stride=2 * co, sbuf=ops.ptr(sb), logdet=ops.ptr(ld), Co=co, clamp=1.2, out2=ops.ptr(y2),
              out2_stride=co, col_tile=ops.coupling_tile(co), in_bf16=1, **common)

    def fwd():
        _lib.check(lib.sininn_conv_sub1_fwd(C.byref(f1), C.byref(f2), ops._stream()))

    def fwd_pair_store():
        f1.out = pb(hsave)
        _lib.check(lib.sininn_conv_pair_k1(C.byref(f1), C.byref(f2), ops._stream()))
        f1.out = None
    alg_f = 4 * m * (k1 + 3 * co + co) / 1e6
    alg_b = 4 * m * (k1 + k2 + 2 * k1) / 1e6
    print(f'bf16 shape: batch {b}, {h}x{w}, Cin {k1}, 2Co {k2}: M = {m}; algorithmic MB forward {alg_f:.0f}, backward {alg_b:.0f} (ADD epilogue)')
    t = timeit(fwd); print(f'persistent forward          : {t:8.1f} us  ({alg_f / t * 1e3:.0f} GB/s algorithmic)')
    print(f'pair forward (h stored)     : {timeit(fwd_pair_store):8.1f} us')
    t = timeit(fused); print(f'fused backward + reduce     : {t:8.1f} us  ({alg_b / t * 1e3:.0f} GB/s algorithmic)')
    print(f'pair + grouped bf16 wgrad   : {timeit(old):8.1f} us')


if __name__ == '__main__':
    main()
