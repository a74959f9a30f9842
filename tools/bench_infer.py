"""Inference throughput: the inverse-only pass (LR window + z -> HR frame) under torch.no_grad at the
reference's validation batch size 40 (data.py:137), 256x256 output, -c 4, fp32.  `python tools/bench_infer.py [--arch SRF|IRN]`"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sin_inn_amd                                   # noqa: E402,F401
import lit_wrapper                                   # noqa: E402
from bench import make_opt                           # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--arch', default='SRF')
    ap.add_argument('--batch', type=int, default=40)
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--precision', choices=['fp32', 'bf16'], default='fp32')
    ap.add_argument('--size', type=int, default=256)
    ap.add_argument('--hook', type=int, default=1, help='sininn_pair_k1_test_hook: 1 default, 5 = fused 3x3 subnet forced on (bf16 no-grad), 3 = forced off, '
                    '0 = no fused 1x1 pair')
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    opt = make_opt(4, 10)
    opt.architecture = a.arch
    opt.precision = a.precision
    from sin_inn_amd import _lib
    _lib.lib().sininn_pair_k1_test_hook(a.hook)
    torch.manual_seed(0)
    model = lit_wrapper.SingleVideoINN(3, a.size, a.size, opt).to(dev).eval()
    lr = torch.rand(a.batch, a.size // 8, a.size // 8, opt.lr_dims, device=dev).permute(0, 3, 1, 2)
    with torch.no_grad():
        def run():
            z = lit_wrapper._latent(a.batch, opt.z_dims, a.size // 8, a.size // 8, dev, opt.temp)
            return model.inn(lit_wrapper._cat_channels(lr, z), rev=True)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            out = run()
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.reps
    assert out.shape == (a.batch, 3, a.size, a.size)
    print(f'{a.arch} inverse pass, {a.precision}, hook {a.hook}, batch {a.batch}, {a.size}x{a.size}: {dt * 1e3:.2f} ms = {a.batch / dt:.0f} frames/s')


if __name__ == '__main__':
    main()
