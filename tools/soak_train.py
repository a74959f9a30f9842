"""Soak run (diagnostic): N training steps of the headline configuration on a synthetic clip, loss printed every `--every`
steps; fails if the loss is not finite or did not fall.  `python tools/soak_train.py [--steps 300] [--arch SRF|IRN] [--precision fp32|bf16]`"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sin_inn_amd                                   # noqa: E402,F401
import lit_wrapper                                   # noqa: E402
from bench import make_opt                           # noqa: E402
from data import FrameStore                          # noqa: E402
from sin_inn_amd import functional as F              # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=300)
    ap.add_argument('--every', type=int, default=50)
    ap.add_argument('--arch', default='SRF')
    ap.add_argument('--precision', default='fp32')
    ap.add_argument('--size', type=int, default=256)
    ap.add_argument('--batch', type=int, default=16)
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    opt = make_opt(4, 10)
    opt.architecture, opt.precision = a.arch, a.precision
    torch.manual_seed(0)
    model = lit_wrapper.SingleVideoINN(3, a.size, a.size, opt).to(dev)
    model.attach_optimizer()
    store = FrameStore.synthetic(64, a.size, a.size).to(dev)
    gen = torch.Generator().manual_seed(7)
    idx_all = torch.randint(10, 54, (a.steps, a.batch), generator=gen).to(device=dev, dtype=torch.int32)
    losses = []
    for s in range(a.steps):
        hr, lr = F.sample_windows(store.hr, store.lr, idx_all[s], opt.lr_window)
        model.training_step([{'hr': hr, 'lr': lr}], s)
        if s % a.every == 0 or s == a.steps - 1:
            v = float(model._logged['train'])
            losses.append(v)
            print(f'step {s:5d}  loss {v:.6f}', flush=True)
            assert v == v and abs(v) < 1e9, 'loss is not finite'
    assert losses[-1] < losses[0], f'the loss did not fall: {losses[0]} -> {losses[-1]}'
    print('ok')


if __name__ == '__main__':
    main()
