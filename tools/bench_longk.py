"""Does the conv kernel approach the MFMA roof when K is long (fixed per-block costs amortised)?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sin_inn_amd
from sin_inn_amd import _lib, ops
dev = torch.device('cuda', 0)
def run(b, hw, cin, n, k=3, reps=10):
    m = b * hw * hw; taps = k * k; npk = ops.pad16(n)
    x = torch.randn(m, cin, device=dev); w = torch.randn(taps * npk * cin, device=dev) * .02
    bias = torch.randn(npk, device=dev); out = torch.empty(m, n, device=dev)
    kw = dict(in_=ops.ptr(x), in_stride=cin, Cin=cin, w=ops.ptr(w), bias=ops.ptr(bias), Np=npk, B=b, H=hw, W=hw, ksize=k,
              mode=_lib.CONV_RELU, out=ops.ptr(out), out_stride=n, N=n)
    for _ in range(2): ops.conv(**kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): ops.conv(**kw)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f'B{b} {hw}x{hw} {cin}->{n} k{k}: {ms*1e3:8.1f} us  {2.0*m*taps*cin*n/ms/1e9:6.1f} TFLOP/s  (iterations/block {taps*cin//32})')
for cin in (32, 64, 128, 256, 512, 1024, 2048):
    run(16, 32, cin, 256)
for cin in (256, 1024):
    run(16, 64, cin, 256)
    run(64, 32, cin, 256)
