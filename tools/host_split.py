"""Diagnostic: per training step at the benchmark's shape, how long the host spends (a) waiting in the run-ahead throttle and
(b) enqueueing work; plus the GPU's step time.  `python tools/host_split.py [--steps 40]`"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.getcwd())
import sin_inn_amd                                   # noqa: E402,F401
import lit_wrapper                                   # noqa: E402
from bench import make_opt                           # noqa: E402
from data import FrameStore                          # noqa: E402
from sin_inn_amd.functional import sample_windows    # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=40)
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    opt = make_opt(4, 10)
    torch.manual_seed(0)
    model = lit_wrapper.SingleVideoINN(3, 256, 256, opt).to(dev)
    model.attach_optimizer()
    store = FrameStore.synthetic(64, 256, 256).to(dev)
    idx = torch.randint(10, 54, (a.steps + 10, 16)).to(device=dev, dtype=torch.int32)
    waits = [0.0]
    orig = model._throttle

    def timed_throttle(hr):
        t = time.perf_counter()
        r = orig(hr)
        waits[0] += time.perf_counter() - t
        return r
    model._throttle = timed_throttle

    def step(i):
        hr, lr = sample_windows(store.hr, store.lr, idx[i], 10)
        model.training_step([{'hr': hr, 'lr': lr}, {'hr': hr, 'lr': lr}], 0)

    for i in range(8):
        step(i)
    torch.cuda.synchronize()
    waits[0] = 0.0
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(8 + i)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f'step {t_all / a.steps * 1e3:.2f} ms | host loop {t_host / a.steps * 1e3:.2f} ms of which throttle wait '
          f'{waits[0] / a.steps * 1e3:.2f} ms, enqueue {(t_host - waits[0]) / a.steps * 1e3:.2f} ms')


if __name__ == '__main__':
    main()
