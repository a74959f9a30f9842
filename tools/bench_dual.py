"""Does running independent conv kernels on two HIP streams hide their prologue/epilogue/tail bubbles?
Times N back-to-back launches of {conv1+relu, dgrad conv2} on one stream vs alternating over two streams."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sin_inn_amd
from sin_inn_amd import _lib, ops

dev = torch.device('cuda', 0)
b, hw, c = 16, 64, 48
m = b * hw * hw
def mk(cin, n, mode, k=3):
    npk = ops.pad16(n); taps = k * k
    x = torch.randn(m, cin, device=dev); w = torch.randn(taps * npk * cin, device=dev) * .05
    bias = torch.randn(npk, device=dev); out = torch.empty(m, 256, device=dev); mk_ = torch.randn(m, 256, device=dev)
    kw = dict(in_=ops.ptr(x), in_stride=cin, Cin=cin, w=ops.ptr(w), bias=ops.ptr(bias), Np=npk, B=b, H=hw, W=hw, ksize=k,
              mode=mode, out=ops.ptr(out), out_stride=256, N=n, mask=ops.ptr(mk_), mask_stride=256)
    return kw, (x, w, bias, out, mk_)
jobs = [mk(24, 256, _lib.CONV_RELU), mk(48, 256, _lib.CONV_MASK), mk(24, 256, _lib.CONV_RELU), mk(48, 256, _lib.CONV_MASK)]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run(two, reps=20):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
    for r in range(reps):
        for i, (kw, _) in enumerate(jobs):
            with torch.cuda.stream(s2 if (two and i % 2) else s1):
                ops.conv(**kw)
    torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for _ in range(2):
    print('one stream : %.1f us per group of 4' % (run(False) * 1e3))
    print('two streams: %.1f us per group of 4' % (run(True) * 1e3))
