"""Where the host time of a training step goes: cProfile over steps whose GPU work is tiny (64x64, batch 1), so the wall time is
the host's.  `python tools/host_profile.py [--arch SRF|IRN] [--precision fp32|bf16] [--steps 60]`"""
import argparse
import cProfile
import os
import pstats
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
    ap.add_argument('--arch', default='SRF')
    ap.add_argument('--precision', default='fp32')
    ap.add_argument('--steps', type=int, default=60)
    ap.add_argument('--size', type=int, default=64)
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    opt = make_opt(4, 10)
    opt.architecture, opt.precision = a.arch, a.precision
    torch.manual_seed(0)
    model = lit_wrapper.SingleVideoINN(3, a.size, a.size, opt).to(dev)
    model.attach_optimizer()
    store = FrameStore.synthetic(32, a.size, a.size).to(dev)
    idx = torch.full((1,), 12, dtype=torch.int32, device=dev)

    def step():
        hr, lr = sample_windows(store.hr, store.lr, idx, 10)
        model.training_step([{'hr': hr, 'lr': lr}, {'hr': hr, 'lr': lr}], 0)

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    print(f'{a.arch} {a.precision}: {(time.perf_counter() - t0) / a.steps * 1e3:.2f} ms per step (host-bound by construction)')
    torch.autograd.set_multithreading_enabled(False)       # backward in THIS thread, so that the profile sees it
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats('tottime').print_stats(40)


if __name__ == '__main__':
    main()
