"""Microbenchmark of the flow-loss operators (csrc/flowloss.hip) at BASELINE config-3 size (512x512) on the GPU box:
    python tools/bench_flowloss.py [--batch 4] [--size 512] [--reps 20]
Prints time and achieved GB/s of ALGORITHMIC bytes (every operand read / written once) against the 8 TB/s HBM roof."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sin_inn_amd                                   # noqa: E402,F401
from sin_inn_amd import flowloss as FL               # noqa: E402


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
    ap.add_argument('--batch', type=int, default=4)
    ap.add_argument('--size', type=int, default=512)
    ap.add_argument('--reps', type=int, default=20)
    a = ap.parse_args()
    b, h, w = a.batch, a.size, a.size
    dev = torch.device('cuda', 0)
    px = b * h * w
    img = torch.rand(b, 3, h, w, device=dev)
    img2 = (img + 0.05 * torch.randn_like(img)).clamp(0, 1)
    flow = torch.randn(b, 2, h, w, device=dev) * 2
    metric = torch.rand(b, 1, h, w, device=dev)
    mask = (torch.rand(b, 1, h, w, device=dev) > 0.2).float()
    rows = []
    x4 = torch.cat([img * metric.exp(), metric.exp()], 1).contiguous()
    out = torch.zeros_like(x4)
    lib, ptr, st = FL._lib.lib(), FL.ptr, FL._stream
    rows.append(('softsplat fwd (4 ch, softmax payload)', timeit(lambda: lib.sininn_softsplat(ptr(x4), ptr(flow), b, 4, h, w, ptr(out), st()), a.reps),
                 px * 4 * (4 + 2 + 4)))
    g = torch.randn_like(x4); gi = torch.empty_like(x4); gf = torch.empty_like(flow)
    rows.append(('softsplat bwd (d in + d flow)', timeit(lambda: lib.sininn_softsplat_bwd(ptr(x4), ptr(flow), ptr(g), b, 4, h, w, ptr(gi), ptr(gf), st()), a.reps),
                 px * 4 * (4 + 2 + 4 + 4 + 2)))
    corr = torch.zeros(b, 1, h, w, device=dev); m = torch.empty_like(corr)
    rows.append(('occlusion_wang (map + mask)', timeit(lambda: lib.sininn_occlusion_wang(ptr(flow), b, h, w, 0.7, ptr(corr), ptr(m), st()), a.reps),
                 px * 4 * (2 + 1 + 1 + 1)))
    acc = torch.zeros(130, device=dev); o = torch.empty(1, device=dev)
    for md in (2, 3):
        rows.append((f'census fwd (max_distance {md})', timeit(lambda: lib.sininn_census(ptr(img), ptr(img2), ptr(mask), 1, b, h, w, md, 0.1, ptr(acc), ptr(o), st()), a.reps),
                     px * 4 * (3 + 3 + 1)))
        g1 = torch.empty_like(img); g2 = torch.empty_like(img)
        rows.append((f'census bwd (max_distance {md})', timeit(lambda: lib.sininn_census_bwd(ptr(img), ptr(img2), ptr(mask), 1, b, h, w, md, 0.1, ptr(acc), None, ptr(g1), ptr(g2), st()), a.reps),
                     px * 4 * (3 + 3 + 1 + 3 + 3)))
    g1 = torch.empty_like(img); g2 = torch.empty_like(img)
    rows.append(('masked L1 fwd', timeit(lambda: lib.sininn_masked_l1(ptr(img), ptr(img2), ptr(mask), 1, b, 3, h, w, 1.0, ptr(acc), ptr(o), st()), a.reps), px * 4 * 7))
    rows.append(('masked L1 bwd', timeit(lambda: lib.sininn_masked_l1_bwd(ptr(img), ptr(img2), ptr(mask), 1, b, 3, h, w, 1.0, ptr(acc), None, ptr(g1), ptr(g2), st()), a.reps), px * 4 * 13))
    gf = torch.empty_like(flow)
    for order in (1, 2):
        rows.append((f'bilateral smooth fwd (order {order})', timeit(lambda: lib.sininn_bilateral_smooth(ptr(img), ptr(flow), b, 3, h, w, order, 1, 150.0, 0.1, ptr(acc), ptr(o), st()), a.reps), px * 4 * 5))
        rows.append((f'bilateral smooth bwd (order {order})', timeit(lambda: lib.sininn_bilateral_smooth_bwd(ptr(img), ptr(flow), b, 3, h, w, order, 1, 150.0, 0.1, None, ptr(gf), st()), a.reps), px * 4 * 7))
    # flow warp + photometric L1 (Resample2d + trainer.py:61-62; north_star's second fused kernel) at config-3 size, batch 16
    fb = max(b, 16)
    fpx = fb * h * w
    fimg = torch.rand(fb, 3, h, w, device=dev); ftgt = torch.rand(fb, 3, h, w, device=dev)
    fflow = torch.randn(fb, 2, h, w, device=dev) * 2
    fwarp = torch.empty_like(fimg); fmet = torch.empty(fb, 1, h, w, device=dev)
    rows.append((f'flow_warp_l1 fwd (batch {fb}: warp + metric)',
                 timeit(lambda: lib.sininn_flow_warp_l1(ptr(fimg), ptr(fflow), ptr(ftgt), fb, 3, h, w, ptr(fwarp), ptr(fmet), st()), a.reps),
                 fpx * 4 * (3 + 2 + 3 + 3 + 1)))
    gwp = torch.randn_like(fimg); gmt = torch.rand_like(fmet); gim = torch.zeros_like(fimg); gfl = torch.empty_like(fflow)

    def fw_bwd():
        gim.zero_()
        lib.sininn_flow_warp_l1_bwd(ptr(fimg), ptr(fflow), ptr(ftgt), ptr(fwarp), ptr(gwp), ptr(gmt), fb, 3, h, w, ptr(gim), ptr(gfl), st())
    rows.append((f'flow_warp_l1 bwd (batch {fb}: d img + d flow, incl. zero fill)', timeit(fw_bwd, a.reps),
                 fpx * 4 * (3 + 2 + 3 + 3 + 3 + 1 + 3 + 3 + 2)))
    rows.append((f'flow_warp_l1 bwd (batch {fb}: d flow only)',
                 timeit(lambda: lib.sininn_flow_warp_l1_bwd(ptr(fimg), ptr(fflow), ptr(ftgt), ptr(fwarp), ptr(gwp), ptr(gmt), fb, 3, h, w, None, ptr(gfl), st()), a.reps),
                 fpx * 4 * (3 + 2 + 3 + 3 + 3 + 1 + 2)))
    # the same three calls on a SMOOTH flow (what a flow network emits; bench.py --with-flow uses the same field): neighbouring
    # pixels sample neighbouring source pixels, so the forward shares taps by shuffle and the image-gradient backward merges
    # coinciding taps across lanes before its LDS atomics.  (The white-noise flow above is the worst case for both.)
    yy, xx = torch.meshgrid(torch.arange(h, device=dev, dtype=torch.float32), torch.arange(w, device=dev, dtype=torch.float32), indexing='ij')
    sflow = torch.stack((3.0 * torch.sin(yy / 37.0) + 1.5 * torch.cos(xx / 23.0), 2.5 * torch.cos(yy / 29.0 + xx / 41.0)))
    sflow = sflow.unsqueeze(0).repeat(fb, 1, 1, 1).contiguous()
    rows.append((f'flow_warp_l1 fwd, smooth flow',
                 timeit(lambda: lib.sininn_flow_warp_l1(ptr(fimg), ptr(sflow), ptr(ftgt), fb, 3, h, w, ptr(fwarp), ptr(fmet), st()), a.reps),
                 fpx * 4 * (3 + 2 + 3 + 3 + 1)))

    def fw_bwd_s():
        gim.zero_()
        lib.sininn_flow_warp_l1_bwd(ptr(fimg), ptr(sflow), ptr(ftgt), ptr(fwarp), ptr(gwp), ptr(gmt), fb, 3, h, w, ptr(gim), ptr(gfl), st())
    rows.append((f'flow_warp_l1 bwd, smooth flow (d img + d flow, incl. zero fill)', timeit(fw_bwd_s, a.reps),
                 fpx * 4 * (3 + 2 + 3 + 3 + 3 + 1 + 3 + 3 + 2)))
    rows.append((f'flow_warp_l1 bwd, smooth flow (d flow only)',
                 timeit(lambda: lib.sininn_flow_warp_l1_bwd(ptr(fimg), ptr(sflow), ptr(ftgt), ptr(fwarp), ptr(gwp), ptr(gmt), fb, 3, h, w, None, ptr(gfl), st()), a.reps),
                 fpx * 4 * (3 + 2 + 3 + 3 + 3 + 1 + 2)))
    s4 = sflow[:b].contiguous()
    rows.append(('softsplat fwd, smooth flow', timeit(lambda: lib.sininn_softsplat(ptr(x4), ptr(s4), b, 4, h, w, ptr(out), st()), a.reps), px * 4 * (4 + 2 + 4)))
    rows.append(('occlusion_wang, smooth flow', timeit(lambda: lib.sininn_occlusion_wang(ptr(s4), b, h, w, 0.7, ptr(corr), ptr(m), st()), a.reps), px * 4 * (2 + 1 + 1 + 1)))
    for name, ms, nbytes in rows:
        print(f'{name:66s} {ms * 1e3:9.1f} us  {nbytes / ms / 1e6:8.1f} GB/s algorithmic = {nbytes / ms / 1e6 / 8000 * 100:5.1f} % of the HBM roof')


if __name__ == '__main__':
    main()
