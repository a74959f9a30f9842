"""Per-launch time of the grouped bf16 3x3 (or 1x1) weight gradient of a GLOW block at a level's shape (two halves: conv2's
gradient from the bf16 hidden tensor and the fp32 tail gradient, conv1's from the fp32 input and the bf16 hidden gradient).
`SININN_LIB=build/variants/libsininn_wgb<N>.so python tools/bench_wgrad_bf16.py` times an ablation build (tools/build_variant.sh
wgb<N> wgrad_mfma.hip "-DWGB_ABL=<N>": 1 no MFMA loop, 2 no transposing LDS stores, 4 no global loads, 8 no operand shifts, 16 no barriers).
    python tools/bench_wgrad_bf16.py [--b 16 --hw 128 --c 48 --ksize 3]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--b', type=int, default=16)
    ap.add_argument('--hw', type=int, default=128)
    ap.add_argument('--c', type=int, default=48, help='channels of the level (48: level 0, 192: level 1)')
    ap.add_argument('--ksize', type=int, default=3)
    ap.add_argument('--reps', type=int, default=10)
    a = ap.parse_args()
    import sin_inn_amd
    from sin_inn_amd import ops
    dev = torch.device('cuda')
    b, h, w, co, k = a.b, a.hw, a.hw, a.c // 2, a.ksize
    m = b * h * w
    bf = torch.bfloat16
    probs = []
    for _ in range(2):
        hid = torch.randn(m, 256, device=dev).to(bf)
        dr = torch.randn(m, 2 * co, device=dev)
        x = torch.randn(m, co, device=dev)
        dh = torch.randn(m, 256, device=dev).to(bf)
        gw2, gb2 = torch.zeros(2 * co, 256, k, k, device=dev), torch.zeros(2 * co, device=dev)
        gw1, gb1 = torch.zeros(256, co, k, k, device=dev), torch.zeros(256, device=dev)
        probs += [(hid, 0, 256, 256, dr, 0, 2 * co, 2 * co, gw2, gb2, True, False), (x, 0, co, co, dh, 0, 256, 256, gw1, gb1, False, True)]

    def run():
        ops.wgrad_group(probs, b, h, w, k)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / a.reps * 1e3
    flops = 2 * 2.0 * m * k * k * 256 * 3 * co
    print(f'{os.environ.get("SININN_LIB", "default build")}: batch {b}, {h}x{w}, C {a.c}, k {k}: {us:8.1f} us per grouped launch + reduce '
          f'({flops / us / 1e6:.0f} TF/s algorithmic)')


if __name__ == '__main__':
    main()
