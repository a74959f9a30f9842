"""GPU: the mixed-precision path (BASELINE configs[3] / [4]): conv subnets on v_mfma_f32_32x32x16_bf16 with fp32
accumulation, hidden tensors stored as bf16, fp32 flow / coupling / log-det / parameter gradients.

Tolerances (stated, separate from the fp32 path's 1e-4):
  * a kernel whose operands are exactly representable in bf16 on both sides differs from torch only by fp32 accumulation
    order: 1e-4 of the max-norm (fp32 outputs), one bf16 ulp = 2^-8 relative (bf16 outputs);
  * the composed network against the oracle's bf16 EMULATION (operands rounded to bf16, fp32 accumulation, straight-through
    gradients): outputs 2e-2 max-norm / 3e-3 L2 (a hidden value that sits on a bf16 rounding boundary lands on the other
    side), input gradients 5e-2 max-norm / 2e-2 L2, per-tensor parameter gradients 5e-2 L2 (the HIP path also rounds the
    coupling-tail gradient dr and the hidden gradient dh to bf16, which the emulation's straight-through gradients do not);
  * against the fp32 oracle: 5e-2 L2 on outputs -- the price of bf16 operands, stated so it is not mistaken for parity."""
import os
import sys
import types

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def relerr(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def bf(t):
    return t.to(torch.bfloat16).float()


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().cuda()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous().cpu()


@pytest.mark.parametrize('ksize', [3, 1])
@pytest.mark.parametrize('cin,n,hw', [(24, 256, (16, 32)), (256, 48, (13, 21)), (96, 256, (9, 17)), (256, 192, (12, 20)),
                                      (48, 256, (8, 16)), (192, 256, (5, 7)),
                                      # Cin <= 32 -> 256 with ReLU: the persistent small-K kernel (conv3_smallk_bf16.hip); tiles cut in x and y
                                      (24, 256, (21, 37)), (8, 256, (17, 16)), (32, 256, (5, 40)), (16, 256, (33, 18)),
                                      # ... and its masked data-gradient twin (2 Co <= 48 channels -> 256, [h > 0] mask): the last section below
                                      (256, 32, (18, 35)), (256, 16, (7, 9)), (256, 48, (40, 33))])
def test_conv_bf16_kernel(ksize, cin, n, hw):
    """every operand flavour of the bf16 conv engine against torch on the same bf16-rounded values."""
    import sin_inn_amd
    from sin_inn_amd import ops, _lib
    torch.manual_seed(cin + n + ksize)
    dev = torch.device('cuda')
    b, (h, w) = 2, hw
    conv = torch.nn.Conv2d(cin, n, ksize, padding=ksize // 2)
    wq, bias = bf(conv.weight.detach()), conv.bias.detach()
    x = torch.randn(b, cin, h, w)
    wf, bfw, wd = ops.pack_conv_bf16(conv.weight.detach().cuda().contiguous(), bias.cuda().contiguous(), None, True)
    npk = ops.pad16(n)
    want = F.conv2d(bf(x), wq, bias, padding=ksize // 2)
    # fp32 input (converted while staged) -> bf16 output, ReLU epilogue (conv1 of a subnet: cond -> h)
    if n % 8 == 0:
        out = torch.full((b, h, w, n), float('nan'), device=dev, dtype=torch.bfloat16)
        ops.conv(in_=ops.ptr(nhwc(x)), in_stride=cin, Cin=cin, w=ops.ptr(wf, dtype=torch.bfloat16), bias=ops.ptr(bfw), Np=npk,
                 B=b, H=h, W=w, ksize=ksize, mode=_lib.CONV_RELU, out=ops.ptr(out, dtype=torch.bfloat16), out_stride=n, N=n,
                 w_bf16=1, in_bf16=0, out_bf16=1)
        got = nchw(out.float())
        ref = F.relu(want)
        assert float((got - ref).abs().max()) <= 2.0 ** -8 * float(ref.abs().max()) + 1e-6
    # bf16 input -> fp32 output, LINEAR (accumulation order is the only difference)
    xb = nhwc(x).to(torch.bfloat16)
    out = torch.full((b, h, w, n), float('nan'), device=dev)
    ops.conv(in_=ops.ptr(xb, dtype=torch.bfloat16), in_stride=cin, Cin=cin, w=ops.ptr(wf, dtype=torch.bfloat16), bias=ops.ptr(bfw),
             Np=npk, B=b, H=h, W=w, ksize=ksize, mode=_lib.CONV_LINEAR, out=ops.ptr(out), out_stride=n, N=n, w_bf16=1, in_bf16=1)
    assert relerr(nchw(out), want) < 1e-4
    # data gradient: bf16 gradient in -> fp32 out + addend (conv1's dgrad: dh -> dcond)
    g = torch.randn(b, n, h, w)
    xin = bf(x).clone().requires_grad_(True)
    F.conv2d(xin, wq, None, padding=ksize // 2).backward(bf(g))
    add = torch.randn(b, h, w, cin, device=dev)
    dx = torch.full((b, h, w, cin), float('nan'), device=dev)
    gb16 = nhwc(g).to(torch.bfloat16)
    ops.conv(in_=ops.ptr(gb16, dtype=torch.bfloat16), in_stride=n, Cin=n, w=ops.ptr(wd, dtype=torch.bfloat16), Np=ops.pad16(cin),
             B=b, H=h, W=w, ksize=ksize, mode=_lib.CONV_ADD, out=ops.ptr(dx), out_stride=cin, N=cin, addend=ops.ptr(add),
             addend_stride=cin, w_bf16=1, in_bf16=1)
    assert relerr(nchw(dx), xin.grad + nchw(add)) < 1e-4
    # data gradient through the ReLU: fp32 gradient in -> bf16 out, masked by a bf16 hidden tensor (conv2's dgrad: dr -> dh)
    if cin % 8 == 0:
        hid = torch.randn(b, h, w, cin, device=dev).to(torch.bfloat16)
        dh = torch.full((b, h, w, cin), float('nan'), device=dev, dtype=torch.bfloat16)
        ops.conv(in_=ops.ptr(nhwc(g)), in_stride=n, Cin=n, w=ops.ptr(wd, dtype=torch.bfloat16), Np=ops.pad16(cin), B=b, H=h, W=w,
                 ksize=ksize, mode=_lib.CONV_MASK, out=ops.ptr(dh, dtype=torch.bfloat16), out_stride=cin, N=cin,
                 mask=ops.ptr(hid, dtype=torch.bfloat16), mask_stride=cin, w_bf16=1, in_bf16=0, out_bf16=1, mask_bf16=1)
        ref = xin.grad * (nchw(hid.float()) > 0)
        got = nchw(dh.float())
        assert float((got - ref).abs().max()) <= 2.0 ** -8 * float(ref.abs().max()) + 1e-6


@pytest.mark.parametrize('ksize', [3, 1])
@pytest.mark.parametrize('pipe', ['bf16', 'f32'])
@pytest.mark.parametrize('hw', [(13, 21), (32, 32)])
def test_wgrad_group_with_bf16_operands(ksize, pipe, hw):
    """weight gradients whose operands live in HBM as bf16 (h for conv2, dh for conv1).  pipe 'bf16' (default): the
    transposing v_mfma_f32_32x32x16_bf16 kernel, the fp32 operand is rounded to bf16 while it is staged; pipe 'f32' (test
    hook 64): the f32-pipe kernels convert the bf16 operand up.  Either way the products are exact in fp32, so the result
    equals torch autograd on the same (rounded) values up to accumulation order."""
    import sin_inn_amd
    from sin_inn_amd import ops, _lib
    torch.manual_seed(5 + ksize)
    b, (h, w) = 2, hw
    probs, wants = [], []
    for cin, n, in_b, dout_b in ((256, 48, True, False), (24, 256, False, True), (256, 192, True, False), (96, 256, False, True),
                                 (16, 128, False, True), (32, 200, True, True)):     # Cin <= 32, N >= 128: the 128 n x 32 c block shape
        conv = torch.nn.Conv2d(cin, n, ksize, padding=ksize // 2)
        x, g = torch.randn(b, cin, h, w), torch.randn(b, n, h, w)
        xv, gv = (bf(x) if (in_b or pipe == 'bf16') else x), (bf(g) if (dout_b or pipe == 'bf16') else g)
        conv(xv).backward(gv)
        xg = nhwc(x).to(torch.bfloat16) if in_b else nhwc(x)
        gg = nhwc(g).to(torch.bfloat16) if dout_b else nhwc(g)
        gw, gbias = torch.zeros_like(conv.weight).cuda(), torch.zeros_like(conv.bias).cuda()
        probs.append((xg, 0, cin, cin, gg, 0, n, n, gw, gbias, in_b, dout_b))
        wants.append((conv.weight.grad, conv.bias.grad, gw, gbias))
    try:
        _lib.lib().sininn_wgrad_test_hooks(64 if pipe == 'f32' else 0)
        ops.wgrad_group(probs, b, h, w, ksize)
    finally:
        _lib.lib().sininn_wgrad_test_hooks(0)
    for gw_ref, gb_ref, gw, gbias in wants:
        assert relerr(gw, gw_ref) < 1e-4 and relerr(gbias, gb_ref) < 1e-4


def _nets(size, num_coupling, lr_window=1, seed=0):
    import archs
    from oracle import sininn_oracle as O
    from test_gpu_model import make_opt, copy_weights
    torch.manual_seed(seed)
    opt = make_opt(num_coupling=num_coupling, lr_window=lr_window)
    net = archs.UncondSRFlow(3, size[0], size[1], opt)
    ref = O.SRFlowOracle(3, size[0], size[1], scale=4, num_coupling=num_coupling)
    copy_weights(ref, net)
    emu = O.SRFlowOracle(3, size[0], size[1], scale=4, num_coupling=num_coupling)
    copy_weights(emu, net)
    for m in emu.modules():
        if isinstance(m, O.GlowBlock):
            m.emulate_bf16 = True
    net.cuda().set_precision('bf16')
    return net, ref, emu, opt


@pytest.mark.parametrize('size,num_coupling', [((32, 48), 2), ((40, 56), 1)])
def test_bf16_network_matches_emulating_oracle(size, num_coupling):
    import sin_inn_amd
    net, ref, emu, opt = _nets(size, num_coupling, seed=3)
    x = torch.rand(2, 3, *size)
    xg = x.cuda().requires_grad_(True); xe = x.clone().requires_grad_(True)
    yg, ye = net(xg), emu(xe)
    assert relerr(yg, ye) < 2e-2 and rel_l2(yg, ye) < 3e-3
    assert relerr(net.log_jacobian(), emu.log_jacobian()) < 2e-2
    with torch.no_grad():
        assert rel_l2(yg, ref(x)) < 5e-2                     # distance to the fp32 reference arithmetic, for the record
    wgt = torch.randn_like(ye)
    (ye * wgt).sum().backward(); (yg * wgt.cuda()).sum().backward()
    assert rel_l2(xg.grad, xe.grad) < 2e-2 and relerr(xg.grad, xe.grad) < 5e-2
    sin_inn_amd.modules.join_side_streams()
    for (n, pg), (_, pe) in zip(net.named_parameters(), emu.named_parameters()):
        assert rel_l2(pg.grad, pe.grad) < 5e-2, n
    # reverse direction + round trip: the inverse recomputes the same bf16 subnet outputs from (nearly) the same inputs
    z = torch.randn(2, 192, size[0] // 8, size[1] // 8)
    with torch.no_grad():
        hg, he = net(z.cuda(), rev=True), emu(z, rev=True)
        assert relerr(hg, he) < 2e-2 and rel_l2(hg, he) < 3e-3
        back = net(net(x.cuda()), rev=True)
        assert rel_l2(back, x) < 5e-3


def test_bf16_training_step_runs_and_tracks_fp32():
    """one training step in both precisions from the same weights / frames / latents: the losses agree to bf16 accuracy and
    the parameter updates point the same way."""
    import lit_wrapper
    from data import FrameStore
    from sin_inn_amd.functional import sample_windows
    from test_gpu_model import make_opt
    results = {}
    for prec in ('fp32', 'bf16'):
        torch.manual_seed(11)
        opt = make_opt(num_coupling=2, lr_window=2, precision=prec)
        model = lit_wrapper.SingleVideoINN(3, 64, 64, opt).cuda()
        optim = model.attach_optimizer()
        store = FrameStore.synthetic(12, 64, 64).to('cuda')
        idx = torch.tensor([3, 4, 6, 8]).cuda()
        hr, lr = sample_windows(store.hr, store.lr, idx, 2)
        z = torch.randn(4, opt.z_dims, 8, 8, generator=torch.Generator().manual_seed(2))
        real = lit_wrapper._latent
        lit_wrapper._latent = lambda b, zd, h, w, device, temp=1.0: z.to(device)
        try:
            model.training_step([{'hr': hr, 'lr': lr}, {'hr': hr, 'lr': lr}], 0)
        finally:
            lit_wrapper._latent = real
        results[prec] = (float(model._logged['train']), optim.flat_grads()[0].clone())
    (l32, g32), (l16, g16) = results['fp32'], results['bf16']
    assert abs(l16 / l32 - 1) < 2e-2
    assert rel_l2(g16, g32) < 6e-2
    cos = float((g16 * g32).sum() / (g16.norm() * g32.norm()))
    assert cos > 0.998


@pytest.mark.parametrize('shape,num_coupling,with_grads', [((512, 512), 4, True), ((720, 1280), 12, False)])
def test_bf16_at_baseline_config_shapes(shape, num_coupling, with_grads):
    """BASELINE configs[3] (512x512, -c 4) and configs[4] (1280x720, -c 12) at their own shapes, batch 1, bf16 path against
    the oracle's bf16 emulation: forward values and log-det (both configs), the reverse direction and gradients (config 3;
    the 24-block 720p backward is minutes of CPU oracle time), plus the HIP round trip."""
    import sin_inn_amd
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    net, ref, emu, opt = _nets(shape, num_coupling, lr_window=10, seed=5)
    g = torch.Generator().manual_seed(9)
    x = torch.rand(1, 3, *shape, generator=g)
    xg = x.cuda().requires_grad_(with_grads)
    xe = x.clone().requires_grad_(with_grads)
    if with_grads:
        yg, ye = net(xg), emu(xe)
    else:
        with torch.no_grad():
            yg, ye = net(xg), emu(xe)
    assert yg.shape == (1, 192, shape[0] // 8, shape[1] // 8)
    # 16 / 48 subnets deep: a hidden value on a bf16 rounding boundary lands on the other side in a few places
    assert rel_l2(yg, ye) < 1e-2 and relerr(yg, ye) < 1e-1
    assert relerr(net.log_jacobian(), emu.log_jacobian()) < 2e-2
    with torch.no_grad():
        back = net(yg.detach(), rev=True)
        assert rel_l2(back, x) < 1e-2
    if with_grads:
        wgt = torch.randn(ye.shape, generator=g)
        (ye * wgt).sum().backward(); (yg * wgt.cuda()).sum().backward()
        assert rel_l2(xg.grad, xe.grad) < 5e-2
        sin_inn_amd.modules.join_side_streams()
        flat_g = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).cpu()
        flat_e = torch.cat([p.grad.reshape(-1) for p in emu.parameters()])
        assert rel_l2(flat_g, flat_e) < 5e-2
        z = torch.randn(1, 192, shape[0] // 8, shape[1] // 8, generator=g)
        with torch.no_grad():
            assert rel_l2(net(z.cuda(), rev=True), emu(z, rev=True)) < 1e-2


def test_bf16_backward_at_config4_shape():
    """BASELINE configs[4]'s frame size (1280x720, lr_window 10) with `-c 2` (4 GLOW blocks: seconds of CPU oracle time),
    batch 1, in the mixed-precision arithmetic the config names: the backward / weight-gradient kernels at THAT size (level 0:
    57 600 pixels in tiles that are ragged in y; level 1: 90 x 160) against the oracle's bf16 emulation at the mixed-precision
    budget stated at the top of this file -- forward values, log-det, input gradient and every parameter gradient, both
    directions.  (The fp32 arithmetic at this size: tests/test_gpu_gates.py, float64 with forced gates, 1e-4 max-norm.)"""
    import sin_inn_amd
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    net, ref, emu, opt = _nets((720, 1280), 2, lr_window=10, seed=7)
    g = torch.Generator().manual_seed(19)
    x = torch.rand(1, 3, 720, 1280, generator=g)
    xg = x.cuda().requires_grad_(True); xc = x.clone().requires_grad_(True)
    yg, yc = net(xg), emu(xc)
    wgt = torch.randn(yc.shape, generator=g)
    (yc * wgt).sum().backward(); (yg * wgt.cuda()).sum().backward()
    sin_inn_amd.modules.join_side_streams()
    assert rel_l2(yg, yc) < 1e-2 and relerr(yg, yc) < 1e-1
    assert relerr(net.log_jacobian(), emu.log_jacobian()) < 2e-2
    assert rel_l2(xg.grad, xc.grad) < 5e-2
    for (n, pg), (_, pc) in zip(net.named_parameters(), emu.named_parameters()):
        assert rel_l2(pg.grad, pc.grad) < 5e-2, (n, rel_l2(pg.grad, pc.grad))
    # reverse direction at this size: values and gradients
    net.zero_grad(); emu.zero_grad()
    zin = yc.detach()
    zg = zin.cuda().requires_grad_(True); zc = zin.clone().requires_grad_(True)
    w2 = torch.randn(1, 3, 720, 1280, generator=g)
    hg, hc = net(zg, rev=True), emu(zc, rev=True)
    (hg * w2.cuda()).sum().backward(); (hc * w2).sum().backward()
    sin_inn_amd.modules.join_side_streams()
    assert rel_l2(hg, hc) < 1e-2
    assert rel_l2(zg.grad, zc.grad) < 5e-2
    for (n, pg), (_, pc) in zip(net.named_parameters(), emu.named_parameters()):
        assert rel_l2(pg.grad, pc.grad) < 5e-2, (n, rel_l2(pg.grad, pc.grad))


@pytest.mark.parametrize('rev', [False, True])
@pytest.mark.parametrize('channels,hw', [(48, (13, 21)), (192, (6, 18)), (48, (64, 64)), (192, (9, 33)), (96, (4, 16))])
def test_fused_3x3_subnet_matches_the_two_launch_path(rev, channels, hw):
    """north_star's single fused coupling kernel (conv_sub3_bf16.hip, sininn_conv_sub3): in a no-grad pass on the mixed-
    precision path a 3x3 subnet + affine coupling + log-det can run as ONE launch with the hidden tile in LDS (SININN_SUB3=1;
    not the default dispatch: measured slower than the two launches it replaces).  Against the same block
    through the two-launch path (bf16 hidden tensor in HBM): the hidden values are rounded to bf16 once in both, from fp32 sums
    accumulated in another order -> a value next to a rounding boundary lands one bf16 ulp apart in a few channels (budget
    8e-3 of the max-norm, 1.5e-3 L2); ragged image sizes (tiles cut by the border in x and y), both coupling
    widths (24 | 24: 16-column interleave, 96 | 96: 32-column), both directions; and against the oracle's bf16 emulation."""
    import archs
    import sin_inn_amd as S
    from sin_inn_amd import _lib
    from oracle import sininn_oracle as O
    torch.manual_seed(channels + hw[0])
    h, w = hw
    blk = S.GLOWCouplingBlock([(channels, h, w)], subnet_constructor=archs.subnet_conv, clamp=1.2)
    emu = O.GlowBlock(channels, 3, 1.2)
    emu.load_state_dict({k: v.clone() for k, v in blk.state_dict().items()})
    emu.emulate_bf16 = True
    for net in (blk, emu):
        for p in net.parameters():
            p.data.mul_(3.0)
    blk.cuda()
    blk.precision = 'bf16'
    x = torch.randn(2, channels, h, w)
    res = []
    try:
        for hook in (5, 3):                      # bit 2: fused 3x3 subnet forced on; bit 1: forced off -> two launches
            _lib.lib().sininn_pair_k1_test_hook(hook)
            with torch.no_grad():
                y = blk([x.cuda()], rev=rev)[0]
            res.append((y.clone(), blk.last_jac.clone()))
    finally:
        _lib.lib().sininn_pair_k1_test_hook(1)
    (y_f, ld_f), (y_2, ld_2) = res
    # (one case of ten measured 3.7e-3 in max-norm: a handful of hidden values one bf16 ulp apart under weights scaled x3)
    assert relerr(y_f, y_2) < 8e-3 and rel_l2(y_f, y_2) < 1.5e-3 and relerr(ld_f, ld_2) < 3e-3
    if h * w >= 512:       # tap-major vs chunk-major fp32 summation: the two paths are different kernels, not one kernel run twice
        assert not torch.equal(y_f, y_2)
    with torch.no_grad():
        y_e = emu(x, rev=rev)
    assert relerr(y_f, y_e) < 2e-2 and rel_l2(y_f, y_e) < 3e-3
    assert relerr(ld_f, emu.last_jac) < 2e-2
    # the differentiable pass (two launches, hidden tensor saved) and the fused no-grad pass agree too
    yg = blk([x.cuda().requires_grad_(True)], rev=rev)[0]
    assert relerr(y_f, yg) < 8e-3 and rel_l2(y_f, yg) < 1.5e-3


def test_fused_3x3_subnet_through_the_c_abi():
    """sininn_conv_sub3 with raw descriptors against sininn_conv(first) + sininn_conv(second) on the same packs."""
    import ctypes as C
    import sin_inn_amd
    from sin_inn_amd import ops, _lib
    torch.manual_seed(3)
    lib = _lib.lib()
    b, h, w, cin, co = 2, 11, 37, 24, 24
    dev = torch.device('cuda')
    c1 = torch.nn.Conv2d(cin, 256, 3, padding=1).cuda()
    c2 = torch.nn.Conv2d(256, 2 * co, 3, padding=1).cuda()
    cmap = ops.coupling_colmap(co, dev)
    w1, b1, _ = ops.pack_conv_bf16(c1.weight.detach().contiguous(), c1.bias.detach().contiguous(), None, False)
    w2, b2, _ = ops.pack_conv_bf16(c2.weight.detach().contiguous(), c2.bias.detach().contiguous(), cmap, False)
    x = torch.randn(b, h, w, cin, device=dev)
    v = torch.randn(b, h, w, co, device=dev)
    outs = []
    for fused in (True, False):
        hid = torch.empty(b, h, w, 256, device=dev, dtype=torch.bfloat16)
        out = torch.zeros(b, h, w, co, device=dev)
        ld = torch.zeros(b, device=dev)
        f = _lib.ConvArgs(inp=x.data_ptr(), in_stride=cin, Cin=cin, w=w1.data_ptr(), bias=b1.data_ptr(), Np=256, B=b, H=h, W=w,
                          ksize=3, mode=_lib.CONV_RELU, out=None if fused else hid.data_ptr(), out_stride=256, N=256,
                          w_bf16=1, in_bf16=0, out_bf16=1)
        s = _lib.ConvArgs(inp=hid.data_ptr(), in_stride=256, Cin=256, w=w2.data_ptr(), bias=b2.data_ptr(), Np=2 * co, B=b, H=h, W=w,
                          ksize=3, mode=_lib.CONV_COUPLE_FWD, out=out.data_ptr(), out_stride=co, N=co, v=v.data_ptr(), v_stride=co,
                          logdet=ld.data_ptr(), Co=co, clamp=1.2, col_tile=ops.coupling_tile(co), w_bf16=1, in_bf16=1)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        if fused:
            assert lib.sininn_conv_sub3_supported(C.byref(f), C.byref(s)) == 1
            _lib.check(lib.sininn_conv_sub3(C.byref(f), C.byref(s), st))
        else:
            _lib.check(lib.sininn_conv(C.byref(f), st))
            _lib.check(lib.sininn_conv(C.byref(s), st))
        outs.append((out, ld))
    assert relerr(outs[0][0], outs[1][0]) < 3e-3 and relerr(outs[0][1], outs[1][1]) < 3e-3
    # a first conv that stores its hidden tensor is not this kernel's case
    f.out = outs[0][0].data_ptr()
    assert lib.sininn_conv_sub3_supported(C.byref(f), C.byref(s)) == 0


@pytest.mark.parametrize('rev', [False, True])
@pytest.mark.parametrize('channels,hw', [(48, (13, 21)), (48, (64, 64)), (16, (9, 33)), (32, (5, 16)), (48, (70, 40)),
                                         (192, (6, 18)), (192, (19, 40)), (192, (45, 80))])    # 192: the level-1 kernels (wide forward / backward + dW1 / dW2)
def test_fused_1x1_subnet_bf16_matches_the_pair_path(rev, channels, hw):
    """Round 4: on the mixed-precision path the level-0 1x1 subnets run as the persistent fused kernels too (conv_sub1_bf16.hip:
    forward without storing h; the whole backward -- recompute, both data gradients, both weight gradients -- in one launch with h /
    dh on chip).  Same block, same inputs with the switch off (bf16 pair kernels + grouped bf16 weight gradients, h / dh in HBM
    as bf16).  From the same fp32 input a subnet's h and dh are bitwise the same in both paths (the same chain of 32x32x16 MFMA steps
    from zero, one rounding to bf16); the second GEMMs and the weight-gradient sums run in another order, so the fp32 tensors
    BETWEEN the two subnets of a block (the first half's output / the fused coupling-backward's dr) differ in the last bit, and a
    value next to a bf16 rounding boundary then lands one bf16 ulp apart in the next subnet (the budget of the fused 3x3 test
    above: 8e-3 of the max-norm -- a flipped ReLU gate in one unit; 5e-4 in L2).  db2 is summed from the fp32 dr here, from its bf16 rounding in the
    grouped kernel (1 - 2e-3 apart).  And against the oracle's bf16 emulation at the budget of this file.  Ragged sizes (tiles
    cut in x and y), the three shapes the kernels serve, both directions, gradients accumulate."""
    import archs
    import sin_inn_amd as S
    from sin_inn_amd import _lib
    from oracle import sininn_oracle as O
    torch.manual_seed(channels + hw[0])
    h, w = hw
    blk = S.GLOWCouplingBlock([(channels, h, w)], subnet_constructor=archs.subnet_conv_1x1, clamp=1.2)
    emu = O.GlowBlock(channels, 1, 1.2)
    emu.load_state_dict({k: v.clone() for k, v in blk.state_dict().items()})
    emu.emulate_bf16 = True
    for net in (blk, emu):
        for p in net.parameters():
            p.data.mul_(3.0)
    blk.cuda()
    blk.precision = 'bf16'
    x = torch.randn(2, channels, h, w)
    wgt, ld_w = torch.randn_like(x), torch.randn(2)
    res = []
    try:
        for fused in (1, 0):
            _lib.lib().sininn_sub1_bwd_test_hook(fused)
            blk.zero_grad()
            for _ in range(2):                          # twice: the gradients accumulate
                xg = x.cuda().requires_grad_(True)
                y = blk([xg], rev=rev)[0]
                ((y * wgt.cuda()).sum() + (blk.last_jac * ld_w.cuda()).sum()).backward()
            S.modules.join_side_streams()
            res.append([y.detach(), blk.last_jac.detach().clone(), xg.grad] + [p.grad.clone() for p in blk.parameters()])
    finally:
        _lib.lib().sininn_sub1_bwd_test_hook(1)
    names = ['y', 'logdet', 'dx'] + [n for n, _ in blk.named_parameters()]
    for i, (a, b) in enumerate(zip(*res)):
        assert relerr(a, b) < 8e-3 and rel_l2(a, b) < (2e-3 if names[i].endswith('2.bias') else 5e-4), (names[i], relerr(a, b), rel_l2(a, b))
    xe = x.clone().requires_grad_(True)
    ye = emu(xe, rev=rev)
    ((ye * wgt).sum() + (emu.last_jac * ld_w).sum()).backward()
    y_f, ld_f, dx_f = res[0][:3]
    assert relerr(y_f, ye) < 2e-2 and rel_l2(y_f, ye) < 3e-3 and relerr(ld_f, emu.last_jac) < 2e-2
    assert rel_l2(dx_f, xe.grad) < 2e-2
    for (n, pe), g in zip(emu.named_parameters(), res[0][3:]):
        assert rel_l2(g, 2 * pe.grad) < 5e-2, n       # two accumulated passes on the device


@pytest.mark.parametrize('co,b,hw,no_dx', [(24, 2, (13, 21), False), (24, 16, (128, 128), False), (8, 3, (9, 33), False), (16, 1, (2, 16), False),
                                           (24, 2, (7, 40), True)])
def test_fused_1x1_subnet_bf16_c_abi(co, b, hw, no_dx):
    """sininn_conv_sub1_fwd / sininn_conv_sub1_bwd with bf16 weight packs, called directly, against torch arithmetic on the SAME
    bf16-rounded operands (x, dr, h, dh and the weights rounded to bf16, products exact, sums in float64): what remains is the
    fp32 accumulation order of the MFMAs, and -- in a handful of the 10^5 .. 10^8 hidden values -- a sum that sits on a bf16 rounding
    boundary and lands one ulp apart (or, for a pre-activation within rounding distance of 0, on the other side of the gate):
    2e-5 in L2 and 2e-3 of the max-norm for y / s / dx, 1e-4 of the max-norm for the log-det and the weight gradients (sums over
    pixels); db2 is summed from the fp32 dr.  Partial tiles, more tiles than persistent blocks (4096 on 256), a single tile, the three shapes, no_dx."""
    import ctypes as C
    import sin_inn_amd
    from sin_inn_amd import _lib, ops
    lib = _lib.lib()
    dev = torch.device('cuda')
    torch.manual_seed(co + b)
    h, w = hw
    k1, k2, m = co, 2 * co, b * h * w
    cx = 2 * co + 8                                               # x lives inside a wider tensor (channel offset 8)
    bft = torch.bfloat16
    xfull = torch.randn(m, cx, device=dev)
    conv1 = torch.nn.Conv2d(k1, 256, 1).to(dev)
    conv2 = torch.nn.Conv2d(256, k2, 1).to(dev)
    with torch.no_grad():
        conv2.weight.mul_(0.3)
    cmap = ops.coupling_colmap(co, dev)
    pk1 = ops.pack_conv_bf16(conv1.weight.detach(), conv1.bias.detach(), None, True)
    pk2 = ops.pack_conv_bf16(conv2.weight.detach(), conv2.bias.detach(), cmap, True)
    dr = torch.randn(m, k2, device=dev)
    addend = torch.randn(m, k1, device=dev)
    vfull = torch.randn(m, cx, device=dev)

    def args(**kw):
        a = _lib.ConvArgs()
        for k, v in kw.items():
            setattr(a, 'inp' if k == 'in_' else k, v)
        return a
    pb = lambda t: ops.ptr(t, dtype=bft)
    common = dict(B=b, H=h, W=w, ksize=1, w_bf16=1)
    # ---- reference on bf16-rounded operands (float64 sums) --------------------------------------------------------------------
    x = xfull[:, 8:8 + k1]
    xb, drb = bf(x).double(), bf(dr).double()
    w1, w2 = bf(conv1.weight.detach().reshape(256, k1)).double(), bf(conv2.weight.detach().reshape(k2, 256)).double()
    hid = bf(torch.relu(xb @ w1.t() + conv1.bias.detach().double()).float()).double()          # one rounding of the fp32 sum
    st = hid @ w2.t() + conv2.bias.detach().double()
    s_ref, t_ref = st[:, :co], st[:, co:]
    L = 1.2 * 0.636 * torch.atan(s_ref / 1.2)
    v = vfull[:, :co].double()
    y_ref = torch.exp(L) * v + t_ref
    ld_ref = L.reshape(b, -1).sum(1)
    dh = bf(((drb @ w2) * (hid > 0)).float()).double()
    dx_ref = dh @ w1 + addend.double()
    # ---- forward ----------------------------------------------------------------------------------------------------------------
    out = torch.full((m, cx), float('nan'), device=dev); sb = torch.empty(m, co, device=dev); ld = torch.zeros(b, device=dev)
    y2 = torch.empty(m, co, device=dev)
    f1 = args(in_=ops.ptr(xfull, 8), in_stride=cx, Cin=k1, w=pb(pk1[0]), bias=ops.ptr(pk1[1]), Np=256, mode=_lib.CONV_RELU, out_stride=256,
              N=256, out_bf16=1, **common)
    f2 = args(in_stride=256, Cin=256, w=pb(pk2[0]), bias=ops.ptr(pk2[1]), Np=k2, mode=_lib.CONV_COUPLE_FWD, out=ops.ptr(out),
              out_stride=cx, v=ops.ptr(vfull), v_stride=cx, sbuf=ops.ptr(sb), logdet=ops.ptr(ld), Co=co, clamp=1.2, out2=ops.ptr(y2),
              out2_stride=co, col_tile=ops.coupling_tile(co), in_bf16=1, **common)
    assert lib.sininn_conv_sub1_fwd_supported(C.byref(f1), C.byref(f2)) == 1
    _lib.check(lib.sininn_conv_sub1_fwd(C.byref(f1), C.byref(f2), ops._stream()))
    torch.cuda.synchronize()
    close = lambda a_, b_: relerr(a_, b_) < 2e-3 and rel_l2(a_, b_) < 2e-5
    assert close(out[:, :co], y_ref.float()) and torch.equal(out[:, :co], y2) and bool(torch.isnan(out[:, co:]).all())
    assert close(sb, s_ref.float()) and relerr(ld, ld_ref.float()) < 1e-4
    # ---- backward ---------------------------------------------------------------------------------------------------------------
    g0 = [torch.randn_like(conv2.weight), torch.randn_like(conv2.bias), torch.randn_like(conv1.weight), torch.randn_like(conv1.bias)]
    gw2, gb2, gw1, gb1 = (g.clone().contiguous() for g in g0)
    dx = torch.full((m, k1), float('nan'), device=dev)
    rc = args(in_=ops.ptr(xfull, 8), in_stride=cx, Cin=k1, w=pb(pk1[0]), bias=ops.ptr(pk1[1]), Np=256, **common)
    d2 = args(in_=ops.ptr(dr), in_stride=k2, Cin=k2, w=pb(pk2[2]), Np=256, mode=_lib.CONV_MASK, out_stride=256, N=256, mask_stride=256,
              out_bf16=1, mask_bf16=1, **common)
    d1 = args(in_stride=256, Cin=256, w=pb(pk1[2]), Np=ops.pad16(k1), mode=_lib.CONV_ADD, out=ops.ptr(dx), out_stride=k1, N=k1,
              addend=ops.ptr(addend), addend_stride=k1, in_bf16=1, **common)
    nbytes = lib.sininn_conv_sub1_bwd_workspace_bytes(k1, co)
    ws = torch.empty(nbytes // 4, device=dev)
    _lib.check(lib.sininn_conv_sub1_bwd(C.byref(rc), C.byref(d2), C.byref(d1), int(no_dx), ops.ptr(gw2), ops.ptr(gb2), ops.ptr(gw1),
                                        ops.ptr(gb1), ops.ptr(ws), nbytes, ops._stream()))
    torch.cuda.synchronize()
    if no_dx:
        assert bool(torch.isnan(dx).all())                         # untouched
    else:
        assert close(dx, dx_ref.float()), (relerr(dx, dx_ref.float()), rel_l2(dx, dx_ref.float()))
    # dr is in conv2's OIHW channel order (ds | dt); only the FORWARD pack of conv2 is column-interleaved
    want = [g0[0].double() + (drb.t() @ hid).reshape(k2, 256, 1, 1), g0[1].double() + dr.double().sum(0),
            g0[2].double() + (dh.t() @ xb).reshape(256, k1, 1, 1), g0[3].double() + dh.sum(0)]
    for name, got, ref_ in zip(('gw2', 'gb2', 'gw1', 'gb1'), (gw2, gb2, gw1, gb1), want):
        assert relerr(got, ref_.float()) < 1e-4, (name, relerr(got, ref_.float()))


@pytest.mark.parametrize('rev', [False, True])
@pytest.mark.parametrize('channels,hw', [(48, (13, 21)), (48, (40, 33)), (16, (9, 33)), (32, (18, 16))])
def test_small_k_3x3_kernels_match_the_general_conv(rev, channels, hw):
    """Round 4: on the mixed-precision path the two "fat output" 3x3 convs of a level-0 subnet -- conv1 (Cin <= 32 -> 256, ReLU) and
    the masked data gradient of conv2 (2 Co <= 48 -> 256) -- run on a persistent kernel with register-resident weights
    (conv3_smallk_bf16.hip); in a training pass conv1 also writes the ReLU gates as a bit mask (a wave ballot per accumulator
    register) that the data gradient reads instead of the 256-channel hidden tensor.  Same block, same inputs with the switch off
    (general bf16 conv, mask read from h): every tensor within the one-bf16-ulp budget of the other A/B tests of this file, ragged
    sizes (16 x 16 tiles cut in x and y), the shapes the kernel serves, both directions; a wrong gate bit would be an O(1) error in dx."""
    import archs
    import sin_inn_amd as S
    from sin_inn_amd import _lib
    torch.manual_seed(channels + hw[1])
    h, w = hw
    blk = S.GLOWCouplingBlock([(channels, h, w)], subnet_constructor=archs.subnet_conv, clamp=1.2)
    for p in blk.parameters():
        p.data.mul_(3.0)
    blk.cuda()
    blk.precision = 'bf16'
    x = torch.randn(2, channels, h, w)
    wgt, ld_w = torch.randn_like(x), torch.randn(2)
    res = []
    try:
        for on in (1, 0):
            _lib.lib().sininn_sub1_bwd_test_hook(on)
            blk.zero_grad()
            xg = x.cuda().requires_grad_(True)
            y = blk([xg], rev=rev)[0]
            ((y * wgt.cuda()).sum() + (blk.last_jac * ld_w.cuda()).sum()).backward()
            S.modules.join_side_streams()
            res.append([y.detach(), blk.last_jac.detach().clone(), xg.grad] + [p.grad.clone() for p in blk.parameters()])
    finally:
        _lib.lib().sininn_sub1_bwd_test_hook(1)
    names = ['y', 'logdet', 'dx'] + [n for n, _ in blk.named_parameters()]
    for n, a, b in zip(names, *res):
        assert relerr(a, b) < 8e-3 and rel_l2(a, b) < 5e-4, (n, relerr(a, b), rel_l2(a, b))


@pytest.mark.parametrize('b,hw,inverse', [(2, (19, 40), False), (1, (6, 18), True), (16, (64, 64), False)])
def test_wide_1x1_bf16_kernels_through_the_pair_entry_point(b, hw, inverse):
    """The level-1 kernels of conv_sub1_bf16.hip (96 -> 256 -> 192) behind sininn_conv_pair_k1, called directly: the persistent forward
    (h stored and not stored, both coupling directions) and the persistent backward pair WITHOUT the weight-gradient rider (the
    block executor always uses the rider; this is the instantiation a caller of the C ABI gets) against torch arithmetic on the
    same bf16-rounded operands, float64 sums: budgets of test_fused_1x1_subnet_bf16_c_abi (2e-5 L2 / 2e-3 max-norm for tensors that
    pass through a rounding to bf16, 1e-4 for the log-det); h and dh themselves: equal up to one bf16 ulp in a few places."""
    import ctypes as C
    import sin_inn_amd
    from sin_inn_amd import _lib, ops
    lib = _lib.lib()
    dev = torch.device('cuda')
    torch.manual_seed(b + hw[0])
    h, w = hw
    co, k1, k2, m = 96, 96, 192, b * h * w
    bft = torch.bfloat16
    x = torch.randn(m, k1, device=dev)
    v = torch.randn(m, co, device=dev)
    conv1 = torch.nn.Conv2d(k1, 256, 1).to(dev)
    conv2 = torch.nn.Conv2d(256, k2, 1).to(dev)
    with torch.no_grad():
        conv2.weight.mul_(0.3)
    pk1 = ops.pack_conv_bf16(conv1.weight.detach(), conv1.bias.detach(), None, True)
    pk2 = ops.pack_conv_bf16(conv2.weight.detach(), conv2.bias.detach(), ops.coupling_colmap(co, dev), True)

    def args(**kw):
        a = _lib.ConvArgs()
        for k, val in kw.items():
            setattr(a, 'inp' if k == 'in_' else k, val)
        return a
    pb = lambda t: ops.ptr(t, dtype=bft)
    common = dict(B=b, H=h, W=w, ksize=1, w_bf16=1)
    close = lambda a_, b_: relerr(a_, b_) < 2e-3 and rel_l2(a_, b_) < 2e-5
    # ---- reference ------------------------------------------------------------------------------------------------------------------
    xb = bf(x).double()
    w1, w2 = bf(conv1.weight.detach().reshape(256, k1)).double(), bf(conv2.weight.detach().reshape(k2, 256)).double()
    hid32 = torch.relu(xb @ w1.t() + conv1.bias.detach().double()).float()
    hid = bf(hid32).double()
    st = hid @ w2.t() + conv2.bias.detach().double()
    s_ref, t_ref = st[:, :co], st[:, co:]
    L = 1.2 * 0.636 * torch.atan(s_ref / 1.2)
    if not inverse:
        y_ref, ld_ref = torch.exp(L) * v.double() + t_ref, L.reshape(b, -1).sum(1)
    else:
        y_ref, ld_ref = (v.double() - t_ref) / torch.exp(L), -L.reshape(b, -1).sum(1)
    # ---- forward, h stored / not stored ------------------------------------------------------------------------------------------------
    for store_h in (True, False):
        out = torch.full((m, co), float('nan'), device=dev); sb = torch.empty(m, co, device=dev); ld = torch.zeros(b, device=dev)
        hs = torch.full((m, 256), float('nan'), device=dev, dtype=bft)
        f1 = args(in_=ops.ptr(x), in_stride=k1, Cin=k1, w=pb(pk1[0]), bias=ops.ptr(pk1[1]), Np=256, mode=_lib.CONV_RELU,
                  out=pb(hs) if store_h else None, out_stride=256, N=256, out_bf16=1, **common)
        f2 = args(in_=pb(hs), in_stride=256, Cin=256, w=pb(pk2[0]), bias=ops.ptr(pk2[1]), Np=k2,
                  mode=_lib.CONV_COUPLE_INV if inverse else _lib.CONV_COUPLE_FWD, out=ops.ptr(out), out_stride=co, v=ops.ptr(v), v_stride=co,
                  sbuf=ops.ptr(sb), logdet=ops.ptr(ld), Co=co, clamp=1.2, col_tile=ops.coupling_tile(co), in_bf16=1, **common)
        assert lib.sininn_conv_pair_k1_supported(C.byref(f1), C.byref(f2)) == 1
        _lib.check(lib.sininn_conv_pair_k1(C.byref(f1), C.byref(f2), ops._stream()))
        torch.cuda.synchronize()
        assert close(out, y_ref.float()) and close(sb, s_ref.float()) and relerr(ld, ld_ref.float()) < 1e-4
        if store_h:
            d = (hs.float() - hid.float()).abs()
            assert float(d.max()) <= 2.0 ** -7 * float(hid.abs().max()) and float((d > 0).float().mean()) < 1e-3
        else:
            assert bool(torch.isnan(hs.float()).all())
    # ---- backward pair: dh = (dr W2) . [h > 0] (stored), dx = dh W1 + addend --------------------------------------------------------------
    dr = torch.randn(m, k2, device=dev)
    addend = torch.randn(m, k1, device=dev)
    hmask = bf(hid32).to(bft)                                       # the h the forward pass stores (reference rounding)
    dh_ref = bf(((bf(dr).double() @ w2) * (hid > 0)).float()).double()
    dx_ref = dh_ref @ w1 + addend.double()
    dh = torch.full((m, 256), float('nan'), device=dev, dtype=bft)
    dx = torch.full((m, k1), float('nan'), device=dev)
    d2 = args(in_=ops.ptr(dr), in_stride=k2, Cin=k2, w=pb(pk2[2]), Np=256, mode=_lib.CONV_MASK, out=pb(dh), out_stride=256, N=256,
              mask=pb(hmask), mask_stride=256, out_bf16=1, mask_bf16=1, **common)
    d1 = args(in_=pb(dh), in_stride=256, Cin=256, w=pb(pk1[2]), Np=ops.pad16(k1), mode=_lib.CONV_ADD, out=ops.ptr(dx), out_stride=k1, N=k1,
              addend=ops.ptr(addend), addend_stride=k1, in_bf16=1, **common)
    assert lib.sininn_conv_pair_k1_supported(C.byref(d2), C.byref(d1)) == 1
    _lib.check(lib.sininn_conv_pair_k1(C.byref(d2), C.byref(d1), ops._stream()))
    torch.cuda.synchronize()
    assert close(dx, dx_ref.float()), (relerr(dx, dx_ref.float()), rel_l2(dx, dx_ref.float()))
    d = (dh.float() - dh_ref.float()).abs()
    assert float(d.max()) <= 2.0 ** -7 * float(dh_ref.abs().max()) and float((d > 0).float().mean()) < 1e-3
