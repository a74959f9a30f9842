"""GPU parity of the IRN architecture (reference archs.py:74-233) -- the architecture whose oracle is pinned to the
reference's own code: Haar, DenseBlock, InvBlockExp (golden G2/G4 fixtures) and the full InvRescaleNet (golden G5)."""
import types

import pytest
import torch

import sin_inn_amd.modules

pytestmark = pytest.mark.gpu
RTOL = 1e-4
T = torch.from_numpy


def relerr(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


def test_haar_golden_and_grad(golden):
    import archs
    from oracle import sininn_oracle as O
    x = T(golden['g2_x'])
    op = archs.HaarDownsampling(3).cuda()
    y = op(x.cuda())
    assert relerr(y, T(golden['g2_fwd'])) < 1e-6
    assert relerr(op(y, rev=True), T(golden['g2_rev'])) < 1e-6
    assert abs(op.last_jac - float(golden['g2_jac_rev'])) < 1e-6
    xg = x.cuda().requires_grad_(True); xc = x.clone().requires_grad_(True)
    wgt = torch.randn(2, 12, 8, 8)
    (op(xg) * wgt.cuda()).sum().backward(); (O.haar_fwd(xc) * wgt).sum().backward()
    assert relerr(xg.grad, xc.grad) < 1e-6
    yg = T(golden['g2_fwd']).cuda().requires_grad_(True); yc = T(golden['g2_fwd']).clone().requires_grad_(True)
    w2 = torch.randn(2, 3, 16, 16)
    (op(yg, rev=True) * w2.cuda()).sum().backward(); (O.haar_inv(yc) * w2).sum().backward()
    assert relerr(yg.grad, yc.grad) < 1e-6


def _load_g4(blk, golden):
    sd = {k[len('g4_sd_'):]: T(golden[k]) for k in golden.files if k.startswith('g4_sd_')}
    blk.load_state_dict(sd)


@pytest.mark.parametrize('rev', [False, True])
def test_invblockexp_golden_and_gradients(golden, rev):
    import archs
    from oracle import sininn_oracle as O
    blk = archs.InvBlockExp(8, 4)
    _load_g4(blk, golden)
    ref = O.InvBlockExpOracle(8, 4)
    ref.load_state_dict({f'{n.split(".")[0]}.convs.{int(n.split(".")[1][4:]) - 1}.{n.split(".")[2]}': v.clone()
                         for n, v in blk.state_dict().items()})
    blk.cuda()
    x = T(golden['g4_x'])
    with torch.no_grad():
        y = blk(x.cuda())
        assert relerr(y, T(golden['g4_fwd'])) < RTOL                       # the reference's own output
        assert relerr(blk(y, rev=True), T(golden['g4_rev'])) < RTOL
        assert relerr(blk.F(x[:, 4:].cuda()), T(golden['g4_dense_F'])) < RTOL
    xin = T(golden['g4_fwd']) if rev else x
    xg = xin.cuda().requires_grad_(True); xc = xin.clone().requires_grad_(True)
    wgt = torch.randn(2, 8, 8, 8)
    (blk(xg, rev=rev) * wgt.cuda()).sum().backward(); (ref(xc, rev=rev) * wgt).sum().backward()
    sin_inn_amd.modules.join_side_streams()                # weight gradients run on the side stream
    assert relerr(xg.grad, xc.grad) < RTOL
    for (n, pg), (_, pc) in zip(blk.named_parameters(), ref.named_parameters()):
        assert relerr(pg.grad, pc.grad) < 3e-4, n


def test_full_irn_matches_reference_fixture_and_oracle_gradients(golden):
    import archs
    from oracle import sininn_oracle as O
    opt = types.SimpleNamespace(scale=4, num_coupling=4, lr_dims=12)
    torch.manual_seed(5)                                   # same seed / constructor order as make_golden.py
    net = archs.InvRescaleNet(3, 64, 64, opt)
    assert sum(p.numel() for p in net.parameters()) == int(golden['g5_nparams'])
    g5 = torch.Generator().manual_seed(55)
    for m in net.modules():
        if isinstance(m, archs.DenseBlock):
            m.conv5.weight.data = torch.randn(m.conv5.weight.shape, generator=g5) * 0.02
    ref = O.IRNOracle(3, 12, scale=4, num_coupling=4)
    O.load_reference_irn_state(ref, {k: v.detach().clone() for k, v in net.state_dict().items()})
    net.cuda()
    x = T(golden['g5_x'])
    with torch.no_grad():
        y = net(x.cuda())
        assert relerr(y[:, ::16, ::2, ::2], T(golden['g5_out_slice'])) < RTOL     # reference's own output
        assert abs(float(y.norm()) / float(golden['g5_out_norm']) - 1) < 1e-5
        assert relerr(net(y, rev=True), x) < RTOL
    xg = x.cuda().requires_grad_(True); xc = x.clone().requires_grad_(True)
    wgt = torch.randn(2, 192, 8, 8)
    (net(xg) * wgt.cuda()).sum().backward(); (ref(xc) * wgt).sum().backward()
    sin_inn_amd.modules.join_side_streams()
    assert relerr(xg.grad, xc.grad) < RTOL
    named = dict(net.named_parameters())
    for (n, pc) in ref.named_parameters():
        parts = n.split('.')                             # blocks.M.F.convs.K.weight
        op_ids = sorted({int(k.split('.')[1]) for k in named if '.conv' in k})
        key = f'operations.{op_ids[int(parts[1])]}.{parts[2]}.conv{int(parts[4]) + 1}.{parts[5]}'
        assert relerr(named[key].grad, pc.grad) < 5e-4, key


def test_irn_training_step_runs():
    import lit_wrapper
    from data import FrameStore
    from sin_inn_amd.functional import sample_windows
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_gpu_model import make_opt
    opt = make_opt(num_coupling=1, architecture='IRN')
    torch.manual_seed(0)
    model = lit_wrapper.SingleVideoINN(3, 32, 32, opt).cuda()
    optim = model.attach_optimizer()
    store = FrameStore.synthetic(8, 32, 32)
    hr, lr = sample_windows(store.hr.cuda(), store.lr.cuda(), torch.tensor([2, 3]).cuda(), 1)
    before = optim.flat_params()[0].clone()
    model.training_step([{'hr': hr, 'lr': lr}, {'hr': hr, 'lr': lr}], 0)
    assert torch.isfinite(model._logged['train'])
    assert not torch.equal(before, optim.flat_params()[0])


def test_irn_at_baseline_config_shape_matches_oracle():
    """IRN (-a IRN) at BASELINE configs[1]'s own shape: 256x256, -c 4, lr_window 10 (splits 24|24 and 84|108), batch 2 --
    the DenseBlock executor with the kernels `bench.py --arch IRN` dispatches (level-1 convs on 64-block grids, grouped
    weight gradients with the pad-channel gap at cin = 84 / 108), against the oracle that is pinned to the reference's own code:
    output, inverse, input gradient and every parameter gradient."""
    import archs
    from oracle import sininn_oracle as O

    def rel_l2(a, b):
        a, b = a.detach().double().cpu(), b.detach().double().cpu()
        return float((a - b).norm() / b.norm().clamp_min(1e-30))

    opt = types.SimpleNamespace(scale=4, num_coupling=4, lr_dims=84)
    torch.manual_seed(11)
    torch.set_num_threads(min(16, len(__import__('os').sched_getaffinity(0))))
    net = archs.InvRescaleNet(3, 256, 256, opt)
    g5 = torch.Generator().manual_seed(12)
    for m in net.modules():
        if isinstance(m, archs.DenseBlock):                # the reference initialises conv5 to zero (identity blocks)
            m.conv5.weight.data = torch.randn(m.conv5.weight.shape, generator=g5) * 0.02
    ref = O.IRNOracle(3, 84, scale=4, num_coupling=4)
    O.load_reference_irn_state(ref, {k: v.detach().clone() for k, v in net.state_dict().items()})
    net.cuda()
    x = torch.rand(2, 3, 256, 256)
    xg = x.cuda().requires_grad_(True); xc = x.clone().requires_grad_(True)
    yg, yc = net(xg), ref(xc)
    assert yg.shape == (2, 192, 32, 32) and relerr(yg, yc) < RTOL
    with torch.no_grad():
        assert relerr(net(yg.detach(), rev=True), x) < RTOL
    wgt = torch.randn(2, 192, 32, 32)
    (yg * wgt.cuda()).sum().backward(); (yc * wgt).sum().backward()
    sin_inn_amd.modules.join_side_streams()
    # bounds from the kink-flip band measured by tools/irn_kink_noise.py (dx L2 5e-5 .. 1.7e-4), see the parameter loop below
    assert rel_l2(xg.grad, xc.grad) < 3e-4, rel_l2(xg.grad, xc.grad)
    assert relerr(xg.grad, xc.grad) < 2e-3, relerr(xg.grad, xc.grad)
    named = dict(net.named_parameters())
    op_ids = sorted({int(k.split('.')[1]) for k in named if '.conv' in k})
    got, want, per = [], [], []
    for (n, pc) in ref.named_parameters():
        parts = n.split('.')                             # blocks.M.F.convs.K.weight
        key = f'operations.{op_ids[int(parts[1])]}.{parts[2]}.conv{int(parts[4]) + 1}.{parts[5]}'
        got.append(named[key].grad.detach().reshape(-1).cpu()); want.append(pc.grad.reshape(-1))
        per.append(rel_l2(named[key].grad, pc.grad))
        # A LeakyReLU unit whose pre-activation is within rounding distance of 0 takes slope 1 in one correct fp32 evaluation
        # and 0.2 in the other; the flip moves one pixel's whole term in its conv's gradient sums and perturbs everything
        # upstream of it.  tools/irn_kink_noise.py quantifies it on the CPU (float64 oracle + 1e-7 .. 1e-6 relative noise on
        # the conv outputs, i.e. an fp32 conv with another summation order): per-tensor median 5e-5 .. 1.7e-4, max 2.5e-3 ..
        # 4.9e-3, flat gradient 2e-4 .. 7e-4.  The bounds below are that band; tools/diag_irn.py additionally shows that the
        # affected convs are identical to three digits across the three weight-gradient algorithms and change with the
        # forward conv algorithm -- the difference is in the gates, not in the arithmetic.
        assert per[-1] < 2e-2, (key, per[-1])
        assert relerr(named[key].grad, pc.grad) < 5e-2, (key, relerr(named[key].grad, pc.grad))
    per.sort()
    assert per[len(per) // 2] < 3e-4, per[len(per) // 2]
    assert per[len(per) * 9 // 10] < 5e-3, per[len(per) * 9 // 10]
    assert rel_l2(torch.cat(got), torch.cat(want)) < 1.5e-3, rel_l2(torch.cat(got), torch.cat(want))
