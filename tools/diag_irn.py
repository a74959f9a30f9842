"""Diagnostic: IRN input / parameter gradients against the oracle, per block and for the full net."""
import sys, os, types, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import archs
import sin_inn_amd
from sin_inn_amd import modules as M, irn as I
from oracle import sininn_oracle as O
def relerr(a, b): return float((a.detach().cpu() - b.detach().cpu()).abs().max() / b.detach().abs().max().clamp_min(1e-30))
LR_DIMS, SIZE = int(os.environ.get('LR_DIMS', 12)), int(os.environ.get('SIZE', 64))
opt = types.SimpleNamespace(scale=4, num_coupling=int(os.environ.get('NC', 4)), lr_dims=LR_DIMS)
M.USE_WINOGRAD[0] = os.environ.get('WINO', '1') != '0'           # 0: direct implicit-GEMM convs (smaller forward rounding error)
torch.manual_seed(5)
from sin_inn_amd import _lib
_lib.lib().sininn_wgrad_test_hooks(int(os.environ.get('WG', 0)))      # bit 1: no Winograd weight gradient, bit 4: per-conv launches
net = archs.InvRescaleNet(3, SIZE, SIZE, opt)
g5 = torch.Generator().manual_seed(55)
for m in net.modules():
    if isinstance(m, archs.DenseBlock):
        m.conv5.weight.data = torch.randn(m.conv5.weight.shape, generator=g5) * 0.02
ref = O.IRNOracle(3, LR_DIMS, scale=4, num_coupling=opt.num_coupling)
O.load_reference_irn_state(ref, {k: v.detach().clone() for k, v in net.state_dict().items()})
net.cuda()
x = torch.rand(2, 3, SIZE, SIZE)
wgt = torch.randn(2, 192, SIZE // 8, SIZE // 8)
xc = x.clone().requires_grad_(True)
(ref(xc) * wgt).sum().backward()
for name, side, sync in [('default', True, False), ('no side stream', False, False), ('sync each', True, True), ('default again', True, False)]:
    M.USE_SIDE_STREAM[0] = side
    I.DEBUG_SYNC[0] = sync
    for p in net.parameters(): p.grad = None
    xg = x.cuda().requires_grad_(True)
    (net(xg) * wgt.cuda()).sum().backward()
    M.join_side_streams(); torch.cuda.synchronize()
    named = dict(net.named_parameters())
    worst = (0, '')
    for (n, pc) in ref.named_parameters():
        parts = n.split('.')
        op_ids = sorted({int(k.split('.')[1]) for k in named if '.conv' in k})
        key = f'operations.{op_ids[int(parts[1])]}.{parts[2]}.conv{int(parts[4]) + 1}.{parts[5]}'
        worst = max(worst, (relerr(named[key].grad, pc.grad), key))
    l2 = float((xg.grad.detach().double().cpu() - xc.grad.double()).norm() / xc.grad.double().norm())
    errs = sorted(((float((named[f'operations.{op_ids[int(n.split(".")[1])]}.{n.split(".")[2]}.conv{int(n.split(".")[4]) + 1}.{n.split(".")[5]}'].grad.detach().double().cpu() - pc.grad.double()).norm() / pc.grad.double().norm()), n) for n, pc in ref.named_parameters()), reverse=True)
    print('  worst L2:', [(round(e, 6), n) for e, n in errs[:6]])
    print(name, 'dx max-norm', relerr(xg.grad, xc.grad), 'L2', l2, 'param', worst, flush=True)
