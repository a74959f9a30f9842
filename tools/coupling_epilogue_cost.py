"""Diagnostic: what the epilogue options of the fused fp32 coupling conv cost at the level-0 shape of BASELINE configs[1]
(batch 16, 64x64, 256 -> 2*24): channel permutation folded into the store (out_map: 4-byte scattered stores), the compact
copy of y (out2) and the saved s (sbuf)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sin_inn_amd
from sin_inn_amd import ops, _lib

dev = torch.device('cuda')
b, h, w, cin, co = 16, 64, 64, 256, 24
torch.manual_seed(0)
conv = torch.nn.Conv2d(cin, 2 * co, 3, padding=1).cuda()
cmap = ops.coupling_colmap(co, dev)
wf, bfw, _ = ops.pack_conv(conv.weight.detach().contiguous(), conv.bias.detach().contiguous(), cmap, False, wino_fwd=True)
hid = torch.randn(b, h, w, cin, device=dev)
x = torch.randn(b, h, w, 48, device=dev)
perm = torch.randperm(48, device=dev).to(torch.int32)
for name, use_map, use_out2, use_s in (('plain stores', 0, 0, 0), ('+ out_map (folded permutation)', 1, 0, 0), ('+ out2 + sbuf (training)', 1, 1, 1),
                                         ('out2 + sbuf, no map', 0, 1, 1)):
    out = torch.empty_like(x); out2 = torch.empty(b, h, w, co, device=dev); sb = torch.empty(b, h, w, co, device=dev)
    ld = torch.zeros(b, device=dev)
    kw = dict(in_=ops.ptr(hid), in_stride=cin, Cin=cin, w=ops.ptr(wf), bias=ops.ptr(bfw), Np=2 * co, winograd=1, B=b, H=h, W=w, ksize=3,
              mode=_lib.CONV_COUPLE_FWD, out=ops.ptr(out), out_stride=48, v=ops.ptr(x), v_stride=48, Co=co, clamp=1.2,
              col_tile=ops.coupling_tile(co), logdet=ops.ptr(ld))
    if use_map: kw['out_map'] = ops.ptr(perm, dtype=torch.int32)
    if use_out2: kw.update(out2=ops.ptr(out2), out2_stride=co)
    if use_s: kw['sbuf'] = ops.ptr(sb)
    for _ in range(3): ops.conv(**kw)
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(20): ops.conv(**kw)
    t1.record(); torch.cuda.synchronize()
    print(f'{name:36s} {t0.elapsed_time(t1) / 20 * 1e3:7.1f} us')

# channel-group-major hidden tensor [C/8][pixel][8]: every halo row of a chunk is one contiguous run
m = b * h * w
hid_g = hid.reshape(m, cin // 8, 8).permute(1, 0, 2).contiguous()
out_a, out_b = torch.empty_like(x), torch.empty_like(x)
base = dict(w=ops.ptr(wf), bias=ops.ptr(bfw), Np=2 * co, winograd=1, B=b, H=h, W=w, ksize=3, mode=_lib.CONV_COUPLE_FWD, out_stride=48,
            v=ops.ptr(x), v_stride=48, Co=co, clamp=1.2, col_tile=ops.coupling_tile(co), Cin=cin)
for name, kw in (('pixel-major input', dict(in_=ops.ptr(hid), in_stride=cin, out=ops.ptr(out_a))),
                 ('group-major input', dict(in_=ops.ptr(hid_g), in_stride=8, in_group_stride=m * 8, out=ops.ptr(out_b)))):
    kw = dict(base, **kw)
    for _ in range(3): ops.conv(**kw)
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(20): ops.conv(**kw)
    t1.record(); torch.cuda.synchronize()
    print(f'{name:36s} {t0.elapsed_time(t1) / 20 * 1e3:7.1f} us')
print('same result:', bool(torch.equal(out_a[..., :co], out_b[..., :co])), 'max abs diff', float((out_a[..., :co] - out_b[..., :co]).abs().max()))
