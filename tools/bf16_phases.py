"""Diagnostic: phase timing of the bf16 3x3 conv kernel (in-kernel s_memtime stamps) on the level-0 coupling-conv shape of
BASELINE configs[3] (batch 16, 128x128 pixels, 256 -> 64 columns)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sin_inn_amd
from sin_inn_amd import ops, _lib

import argparse
ap = argparse.ArgumentParser()
ap.add_argument('--cin', type=int, default=256)
ap.add_argument('--n', type=int, default=64)
ap.add_argument('--hidden-out', action='store_true', help='the conv1 / dgrad2 shape: fp32 input, ReLU, bf16 output (the hidden tensor)')
args = ap.parse_args()
dev = torch.device('cuda')
b, h, w, cin, n = 16, 128, 128, args.cin, args.n
torch.manual_seed(0)
conv = torch.nn.Conv2d(cin, n, 3, padding=1).cuda()
wf, bfw, wd = ops.pack_conv_bf16(conv.weight.detach().contiguous(), conv.bias.detach().contiguous(), None, False)
x = torch.randn(b, h, w, cin, device=dev)
if not args.hidden_out:
    x = x.to(torch.bfloat16)
out = torch.empty((b, h, w, n), device=dev, dtype=torch.bfloat16 if args.hidden_out else torch.float32)
names = ['barrier A', 'staging (store_chunk)', 'barrier B', 'global load issue', 'MFMA loop', 'epilogue']
for rep in range(3):
    st = torch.zeros(8, dtype=torch.int64, device=dev)
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    if args.hidden_out:
        ops.conv(in_=ops.ptr(x), in_stride=cin, Cin=cin, w=ops.ptr(wf, dtype=torch.bfloat16), bias=ops.ptr(bfw),
                 Np=n, B=b, H=h, W=w, ksize=3, mode=_lib.CONV_RELU, out=ops.ptr(out, dtype=torch.bfloat16), out_stride=n, N=n, w_bf16=1,
                 in_bf16=0, out_bf16=1, stamp=st.data_ptr() if rep == 2 else None)
    else:
        ops.conv(in_=ops.ptr(x, dtype=torch.bfloat16), in_stride=cin, Cin=cin, w=ops.ptr(wf, dtype=torch.bfloat16), bias=ops.ptr(bfw),
                 Np=n, B=b, H=h, W=w, ksize=3, mode=_lib.CONV_LINEAR, out=ops.ptr(out), out_stride=n, N=n, w_bf16=1, in_bf16=1,
                 stamp=st.data_ptr() if rep == 2 else None)
    t1.record(); torch.cuda.synchronize()
    print(f'rep {rep}: {t0.elapsed_time(t1) * 1e3:.1f} us', 2.0 * b * h * w * 9 * cin * n / (t0.elapsed_time(t1) * 1e-3) / 1e12, 'TF/s')
s = st.cpu().tolist()
blocks = s[7]
print('blocks', blocks, 'cycles per block', s[6] / blocks)
for k, nm in enumerate(names):
    print(f'  {nm:24s} {s[k] / blocks:10.0f} cycles per block  {100.0 * s[k] / s[6]:5.1f} %')
