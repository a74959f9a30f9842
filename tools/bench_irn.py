"""Throughput of the IRN architecture (`-a IRN`) on the same workload as bench.py (256x256x3, batch 16,
lr_window 10, -c 4): full training step, fp32.  `python tools/bench_irn.py [--steps 10]` on the GPU box."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sin_inn_amd                                   # noqa: E402,F401
import lit_wrapper                                   # noqa: E402
from bench import make_opt                           # noqa: E402
from data import FrameStore                          # noqa: E402
from sin_inn_amd.functional import sample_windows    # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--batch', type=int, default=16)
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    opt = make_opt(4, 10)
    opt.architecture = 'IRN'
    torch.manual_seed(0)
    model = lit_wrapper.SingleVideoINN(3, 256, 256, opt).to(dev)
    model.attach_optimizer()
    store = FrameStore.synthetic(64, 256, 256).to(dev)
    g = torch.Generator().manual_seed(1)

    def step():
        idx = torch.randint(10, 54, (a.batch,), generator=g).to(dev)
        hr, lr = sample_windows(store.hr, store.lr, idx, 10)
        model.training_step([{'hr': hr, 'lr': lr}] * 2, 0)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    n = sum(p.numel() for p in model.parameters())
    print(f'IRN -c 4 ({n / 1e6:.2f} M parameters): {dt * 1e3:.1f} ms / step = {a.batch / dt:.0f} training frames/s (fp32, batch {a.batch}, 256x256)')


if __name__ == '__main__':
    main()
