import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import archs
import sin_inn_amd as S
from sin_inn_amd import _lib
channels, (h, w), rev = 48, (13, 21), False
torch.manual_seed(channels + h)
blk = S.GLOWCouplingBlock([(channels, h, w)], subnet_constructor=archs.subnet_conv_1x1, clamp=1.2)
for p in blk.parameters():
    p.data.mul_(3.0)
blk.cuda(); blk.precision = 'bf16'
x = torch.randn(2, channels, h, w)
wgt, ld_w = torch.randn_like(x), torch.randn(2)
res = []
for fused in (1, 0):
    _lib.lib().sininn_sub1_bwd_test_hook(fused)
    blk.zero_grad()
    xg = x.cuda().requires_grad_(True)
    y = blk([xg], rev=rev)[0]
    ((y * wgt.cuda()).sum() + (blk.last_jac * ld_w.cuda()).sum()).backward()
    S.modules.join_side_streams()
    res.append([y.detach(), blk.last_jac.detach().clone(), xg.grad] + [p.grad.clone() for p in blk.parameters()])
_lib.lib().sininn_sub1_bwd_test_hook(1)
names = ['y', 'logdet', 'dx'] + [n for n, _ in blk.named_parameters()]
for n, a, b in zip(names, *res):
    d = (a - b).abs().float()
    scale = float(b.abs().max())
    bad = (d > 1e-4 * scale)
    print(n, tuple(a.shape), 'max rel %.3e' % (float(d.max()) / scale), 'n_bad', int(bad.sum()), 'of', d.numel())
    if n == 'y':
        idx = bad.nonzero()
        print(' bad channels', sorted(set(idx[:, 1].tolist()))[:50])
        print(' bad idx sample', idx[:12].tolist())
