"""GPU parity: every HIP kernel (called through the C ABI via sin_inn_amd.ops) against the CPU oracle on the
same seeded inputs.  Tolerance for this path (north_star): 1e-4 relative, fp32."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

RTOL = 1e-4


def relerr(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


@pytest.fixture(scope='module')
def env():
    import sin_inn_amd
    from oracle import sininn_oracle as O
    assert torch.cuda.is_available(), 'GPU tests need a GPU'
    return sin_inn_amd, O, torch.device('cuda', 0)


def nhwc(t):           # (B,C,H,W) cpu -> contiguous (B,H,W,C) cuda
    return t.permute(0, 2, 3, 1).contiguous().cuda()


def nchw(t):           # (B,H,W,C) cuda -> (B,C,H,W) cpu
    return t.permute(0, 3, 1, 2).contiguous().cpu()


# ---------------------------------------------------------------------------------------------------
# conv engine
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('ksize', [3, 1])
@pytest.mark.parametrize('cin,n,hw', [(24, 256, (16, 32)), (96, 256, (8, 16)), (256, 48, (16, 16)),
                                      (256, 192, (12, 20)), (48, 256, (9, 17)), (8, 24, (5, 7))])
@pytest.mark.parametrize('force', [(0, 0), (1, 8), (2, 16), (10, 0), (12, 16), (100, 0), (200, 0)])   # >=10: pin the 16-wide MFMA kernel
def test_conv_relu_and_linear(env, ksize, cin, n, hw, force):
    S, O, dev = env
    from sin_inn_amd import ops, _lib
    torch.manual_seed(cin + n + ksize)
    h, w = hw
    conv = torch.nn.Conv2d(cin, n, ksize, padding=ksize // 2)
    x = torch.randn(2, cin, h, w)
    want = conv(x)
    xg = nhwc(x)
    wf, bf, _ = ops.pack_conv(conv.weight.detach().cuda().contiguous(), conv.bias.detach().cuda().contiguous(), None, False)
    _lib.lib().sininn_conv_test_hooks(*force)
    try:
        for mode, ref in ((_lib.CONV_LINEAR, want), (_lib.CONV_RELU, F.relu(want))):
            out = torch.full((2, h, w, n), float('nan'), device=dev)
            ops.conv(in_=ops.ptr(xg), in_stride=cin, Cin=cin, w=ops.ptr(wf), bias=ops.ptr(bf), Np=ops.pad16(n),
                     B=2, H=h, W=w, ksize=ksize, mode=mode, out=ops.ptr(out), out_stride=n, N=n)
            assert relerr(nchw(out), ref) < RTOL
    finally:
        _lib.lib().sininn_conv_test_hooks(0, 0)


@pytest.mark.parametrize('ksize', [3, 1])
def test_conv_strided_input_and_mask_add(env, ksize):
    """channel sub-range input (x2 = x[:, 24:]), MASK and ADD (+map, in place) epilogues = the dgrad path."""
    S, O, dev = env
    from sin_inn_amd import ops, _lib
    torch.manual_seed(5)
    b, h, w = 2, 10, 18
    x = torch.randn(b, 48, h, w)
    conv = torch.nn.Conv2d(24, 256, ksize, padding=ksize // 2)
    xg = nhwc(x)
    wf, bf, wd = ops.pack_conv(conv.weight.detach().cuda().contiguous(), conv.bias.detach().cuda().contiguous(), None, True)
    out = torch.empty((b, h, w, 256), device=dev)
    ops.conv(in_=ops.ptr(xg, 24), in_stride=48, Cin=24, w=ops.ptr(wf), bias=ops.ptr(bf), Np=256, B=b, H=h, W=w,
             ksize=ksize, mode=_lib.CONV_RELU, out=ops.ptr(out), out_stride=256, N=256)
    hid = F.relu(conv(x[:, 24:]))
    assert relerr(nchw(out), hid) < RTOL
    # data gradient through the conv: d/dx of sum(conv(x) * g), masked / added
    g = torch.randn(b, 256, h, w)
    xin = x[:, 24:].clone().requires_grad_(True)
    conv(xin).backward(g)
    gg = nhwc(g)
    add = torch.randn(b, h, w, 48, device=dev)
    amap = torch.randperm(48)[:24].to(torch.int32).cuda()
    dcond = torch.empty((b, h, w, 24), device=dev)
    ops.conv(in_=ops.ptr(gg), in_stride=256, Cin=256, w=ops.ptr(wd), Np=32, B=b, H=h, W=w, ksize=ksize,
             mode=_lib.CONV_ADD, out=ops.ptr(dcond), out_stride=24, N=24, addend=ops.ptr(add), addend_stride=48,
             addend_map=ops.ptr(amap, dtype=torch.int32))
    want = xin.grad + nchw(add)[:, amap.cpu().long()]
    assert relerr(nchw(dcond), want) < RTOL
    # in-place accumulate into a channel sub-range
    acc = add.clone()
    ops.conv(in_=ops.ptr(gg), in_stride=256, Cin=256, w=ops.ptr(wd), Np=32, B=b, H=h, W=w, ksize=ksize,
             mode=_lib.CONV_ADD, out=ops.ptr(acc, 24), out_stride=48, N=24, addend=ops.ptr(acc, 24), addend_stride=48)
    assert relerr(nchw(acc)[:, 24:], xin.grad + nchw(add)[:, 24:]) < RTOL
    assert torch.equal(acc[..., :24], add[..., :24])
    # MASK: dgrad of the second conv through the ReLU
    conv2 = torch.nn.Conv2d(256, 48, ksize, padding=ksize // 2)
    _, _, wd2 = ops.pack_conv(conv2.weight.detach().cuda().contiguous(), conv2.bias.detach().cuda().contiguous(), None, True)
    g2 = torch.randn(b, 48, h, w)
    hin = hid.detach().clone().requires_grad_(True)
    conv2(hin).backward(g2)
    dh = torch.empty((b, h, w, 256), device=dev)
    ops.conv(in_=ops.ptr(nhwc(g2)), in_stride=48, Cin=48, w=ops.ptr(wd2), Np=256, B=b, H=h, W=w, ksize=ksize,
             mode=_lib.CONV_MASK, out=ops.ptr(dh), out_stride=256, N=256, mask=ops.ptr(out), mask_stride=256)
    assert relerr(nchw(dh), hin.grad * (hid > 0)) < RTOL


@pytest.mark.parametrize('ksize', [3, 1])
@pytest.mark.parametrize('cin,n', [(256, 48), (24, 256), (96, 256), (256, 192)])
@pytest.mark.parametrize('force16', [0, 1, 2, 3, 8, 12])   # bit0: 16-wide MFMA tiles, bit1: no Winograd wgrad for 3x3,
                                                            # bit2: 8-row tiles, bit3: 8-wave blocks with in-block k split
def test_wgrad(env, ksize, cin, n, force16):
    S, O, dev = env
    from sin_inn_amd import ops, _lib
    _lib.lib().sininn_wgrad_test_hooks(force16)
    torch.manual_seed(cin * 3 + n)
    b, h, w = 3, 13, 21                               # odd sizes: partial pixel tiles and partial Winograd tiles
    conv = torch.nn.Conv2d(cin, n, ksize, padding=ksize // 2)
    x = torch.randn(b, cin + 8, h, w)
    g = torch.randn(b, n, h, w)
    conv(x[:, 8:]).backward(g)
    gw = torch.ones_like(conv.weight).cuda()          # wgrad accumulates (+=)
    gb = torch.ones_like(conv.bias).cuda()
    try:
        ops.wgrad(nhwc(x), 8, cin + 8, cin, nhwc(g), n, n, b, h, w, ksize, gw, gb)
    finally:
        _lib.lib().sininn_wgrad_test_hooks(0)
    assert relerr(gw.cpu() - 1, conv.weight.grad) < RTOL
    assert relerr(gb.cpu() - 1, conv.bias.grad) < RTOL


@pytest.mark.parametrize('ksize', [3, 1])
@pytest.mark.parametrize('level', [0, 1])
def test_wgrad_group(env, ksize, level):
    """The four weight gradients of a GLOW block in one grouped launch pair (sininn_wgrad_group) against torch autograd:
    level-0 shapes (24 -> 256, 256 -> 48) and level-1 shapes (96 -> 256, 256 -> 192), odd image sizes (partial pixel /
    Winograd tiles), strided inputs, += semantics; run twice -> bitwise identical."""
    S, O, dev = env
    from sin_inn_amd import ops
    torch.manual_seed(40 + ksize + level)
    b, h, w = 3, 13, 21
    half = 24 if level == 0 else 96
    shapes = [(256, 2 * half), (half, 256), (256, 2 * half), (half, 256)]
    problems, wants = [], []
    for cin, n in shapes:
        conv = torch.nn.Conv2d(cin, n, ksize, padding=ksize // 2)
        x = torch.randn(b, cin + 8, h, w)
        g = torch.randn(b, n + 4, h, w)
        conv(x[:, 8:]).backward(g[:, :n])
        wants.append((conv.weight.grad, conv.bias.grad))
        problems.append([nhwc(x), 8, cin + 8, cin, nhwc(g), 0, n + 4, n])

    def run():
        outs = []
        probs = []
        for pr, (gw_ref, gb_ref) in zip(problems, wants):
            gw, gb = torch.ones_like(gw_ref).cuda(), torch.ones_like(gb_ref).cuda()
            outs.append((gw, gb))
            probs.append(tuple(pr) + (gw, gb))
        ops.wgrad_group(probs, b, h, w, ksize)
        torch.cuda.synchronize()
        return outs

    first, second = run(), run()
    for (gw, gb), (gw2, gb2), (gw_ref, gb_ref) in zip(first, second, wants):
        assert relerr(gw.cpu() - 1, gw_ref) < RTOL
        assert relerr(gb.cpu() - 1, gb_ref) < RTOL
        assert torch.equal(gw, gw2) and torch.equal(gb, gb2)


def test_golden_subnets_on_gpu(env, golden):
    """the reference's own subnet_conv / subnet_conv_1x1 outputs (tests/golden) through the HIP conv engine."""
    S, O, dev = env
    from sin_inn_amd import ops, _lib
    for tag, k in (('3x3', 3), ('1x1', 1)):
        x = torch.from_numpy(golden[f'g3_{tag}_x'])
        b, c, h, w = x.shape
        w0 = torch.from_numpy(golden[f'g3_{tag}_w0']).cuda(); b0 = torch.from_numpy(golden[f'g3_{tag}_b0']).cuda()
        w2 = torch.from_numpy(golden[f'g3_{tag}_w2']).cuda(); b2 = torch.from_numpy(golden[f'g3_{tag}_b2']).cuda()
        p0 = ops.pack_conv(w0, b0, None, False); p2 = ops.pack_conv(w2, b2, None, False)
        hid = torch.empty((b, h, w, 256), device=dev); out = torch.empty((b, h, w, 48), device=dev)
        ops.conv(in_=ops.ptr(nhwc(x)), in_stride=c, Cin=c, w=ops.ptr(p0[0]), bias=ops.ptr(p0[1]), Np=256, B=b, H=h, W=w,
                 ksize=k, mode=_lib.CONV_RELU, out=ops.ptr(hid), out_stride=256, N=256)
        ops.conv(in_=ops.ptr(hid), in_stride=256, Cin=256, w=ops.ptr(p2[0]), bias=ops.ptr(p2[1]), Np=48, B=b, H=h, W=w,
                 ksize=k, mode=_lib.CONV_LINEAR, out=ops.ptr(out), out_stride=48, N=48)
        assert relerr(nchw(out), torch.from_numpy(golden[f'g3_{tag}_y'])) < RTOL


# ---------------------------------------------------------------------------------------------------
# index maps, losses, warps, sampler, adam
# ---------------------------------------------------------------------------------------------------
def test_squeeze_permute(env):
    S, O, dev = env
    from sin_inn_amd.modules import squeeze_op, PermuteRandom, IRevNetDownsampling
    x = torch.randn(2, 3, 16, 24)
    y = squeeze_op(x.cuda(), 2)                      # NCHW in, pixel-major out, two levels fused
    assert torch.equal(y.cpu(), O.squeeze_fwd(O.squeeze_fwd(x)))
    back = squeeze_op(y, 2, inverse=True, out_pixel_major=False)
    assert back.is_contiguous() and torch.equal(back.cpu(), x)
    op = IRevNetDownsampling([(3, 16, 24)])
    assert torch.equal(op([x.cuda()])[0].cpu(), O.squeeze_fwd(x))
    assert torch.equal(op([op([x.cuda()])[0]], rev=True)[0].cpu(), x)
    p = PermuteRandom([(48, 4, 4)], seed=3)
    perm, inv = O.permutation(48, 3)
    z = torch.randn(2, 48, 4, 4)
    assert torch.equal(p([z.cuda()])[0].cpu(), z[:, perm])
    assert torch.equal(p([z.cuda()], rev=True)[0].cpu(), z[:, inv])
    # autograd: gradient of a permutation is the inverse permutation
    zc = z.cuda().requires_grad_(True)
    (p([zc])[0] * torch.arange(48, device=dev).view(1, -1, 1, 1)).sum().backward()
    want = torch.zeros(48); want[perm] = torch.arange(48.)
    assert torch.equal(zc.grad[0, :, 0, 0].cpu(), want)


@pytest.mark.parametrize('c,levels', [(3, 2), (3, 1), (48, 1), (12, 2), (48, 0)])
def test_squeeze_rows_fast_path_is_bit_exact(env, c, levels):
    """The gather-form kernel for dense pixel-major tensors (16-byte stores) against the oracle's index formula: forward,
    inverse, with the channel map on the fine side in read form (forward) and in written form (inverse), and the adjoints
    autograd asks for."""
    S, O, dev = env
    from sin_inn_amd import ops
    torch.manual_seed(c + levels)
    b, h, w = 2, 8, 12
    x = torch.randn(b, c, h, w)
    want = x
    for _ in range(levels):
        want = O.squeeze_fwd(want)
    xg = x.cuda().contiguous(memory_format=torch.channels_last)
    f = 1 << levels
    cc = c * f * f

    def coarse_buf():
        return torch.empty((b, h // f, w // f, cc), device=dev).permute(0, 3, 1, 2)

    def fine_buf():
        return torch.empty((b, h, w, c), device=dev).permute(0, 3, 1, 2)

    assert ops._dense_pixel_major(xg) and ops._dense_pixel_major(coarse_buf())
    out = coarse_buf()
    ops.squeeze(xg, out, b, c, h, w, levels, False)
    assert torch.equal(out.cpu(), want)
    back = fine_buf()
    ops.squeeze(out, back, b, c, h, w, levels, True)
    assert torch.equal(back.cpu(), x)
    perm = torch.randperm(c).to(torch.int32).cuda()
    pl = perm.long().cpu()
    # forward, map on the side READ (fine): coarse = squeeze(x[:, perm])
    ops.squeeze(xg, out, b, c, h, w, levels, False, perm, False)
    w2 = x[:, pl]
    for _ in range(levels):
        w2 = O.squeeze_fwd(w2)
    assert torch.equal(out.cpu(), w2)
    # inverse, map on the side WRITTEN (fine): fine[:, perm[c]] = unsqueeze(coarse)[:, c]
    ops.squeeze(want.cuda().contiguous(memory_format=torch.channels_last), back, b, c, h, w, levels, True, perm, True)
    w3 = torch.empty_like(x); w3[:, pl] = x
    assert torch.equal(back.cpu(), w3)


def test_losses(env, golden):
    S, O, dev = env
    import loss
    x, y = torch.from_numpy(golden['g1_x']), torch.from_numpy(golden['g1_y'])
    assert abs(float(loss.reconstruction(x.cuda(), y.cuda())) / float(golden['g1_rec']) - 1) < 1e-5
    assert abs(float(loss.latent_nll(x.cuda())) / float(golden['g1_nll']) - 1) < 1e-5
    # strided views (channel slice of a pixel-major tensor) + gradients
    torch.manual_seed(0)
    a = torch.randn(3, 20, 6, 5); b = torch.randn(3, 12, 6, 5)
    ag = a.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    ac = a.clone().requires_grad_(True)
    lg = 0.7 * loss.reconstruction(ag[:, :12], b.cuda()) + 0.3 * loss.latent_nll(ag[:, 12:])
    lc = 0.7 * O.reconstruction(ac[:, :12], b) + 0.3 * O.latent_nll(ac[:, 12:])
    lg.backward(); lc.backward()
    assert abs(float(lg) / float(lc) - 1) < 1e-5
    assert relerr(ag.grad, ac.grad) < RTOL


@pytest.mark.parametrize('rev', [False, True])
def test_mmd(env, rev):
    S, O, dev = env
    import loss
    torch.manual_seed(1)
    x = (torch.randn(6, 5, 4, 3) * 0.3); y = (torch.randn(6, 5, 4, 3) * 0.3)
    xc, yc = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
    xg = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    yg = y.cuda().requires_grad_(True)
    lc = O.mmd(xc, yc, rev=rev); lc.backward()
    lg = loss.mmd(xg, yg, rev=rev); lg.backward()
    assert abs(float(lg) - float(lc)) < 1e-5 * max(1.0, abs(float(lc)))
    assert relerr(xg.grad, xc.grad) < 1e-3 and relerr(yg.grad, yc.grad) < 1e-3


@pytest.mark.parametrize('tag', ['a', 'b'])
@pytest.mark.parametrize('rev', [False, True])
def test_mmd_matches_reference_fixture(env, golden, tag, rev):
    """HIP loss.mmd against fixture G7 = the reference's own loss.mmd (loss.py:9-36) evaluated on CPU."""
    import loss
    r = 'rev' if rev else 'fwd'
    x = torch.from_numpy(golden[f'g7_{tag}_x']).cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = torch.from_numpy(golden[f'g7_{tag}_y']).cuda().requires_grad_(True)
    val = loss.mmd(x, y, rev=rev)
    val.backward()
    want = float(golden[f'g7_{tag}_{r}'])
    assert abs(float(val) - want) <= 1e-5 * max(1.0, abs(want))
    # gradient tolerance: the b x b distance matrix is a difference of Gram entries (r_i + r_j - 2 g_ij) whose fp32
    # cancellation error is amplified by d/dd (C+d)^-a for small d; 1e-3 of the max-norm (stated in DESIGN 4)
    assert relerr(x.grad, torch.from_numpy(golden[f'g7_{tag}_{r}_gx'])) < 1e-3
    assert relerr(y.grad, torch.from_numpy(golden[f'g7_{tag}_{r}_gy'])) < 1e-3


def test_tcr_affine_warp(env):
    S, O, dev = env
    from tcr import TCR
    torch.manual_seed(2)
    img = torch.rand(3, 5, 12, 16)
    rand = torch.rand(3, 3)
    for scale in (1, 0.25):
        want = O.tcr_warp(img, rand, 5.0, 5.0, scale=scale)
        got = TCR(5.0, 5.0)(img.cuda(), rand, scale=scale)
        assert relerr(got, want) < RTOL
    ic = img.clone().requires_grad_(True)
    ig = img.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wgt = torch.randn(3, 5, 12, 16)
    (O.tcr_warp(ic, rand, 5.0, 5.0) * wgt).sum().backward()
    (TCR(5.0, 5.0)(ig, rand) * wgt.cuda()).sum().backward()
    assert relerr(ig.grad, ic.grad) < RTOL


def test_flow_warp_l1(env):
    S, O, dev = env
    from sin_inn_amd.functional import flow_warp_l1
    torch.manual_seed(3)
    img = torch.rand(2, 3, 20, 70); tgt = torch.rand(2, 3, 20, 70)
    flow = torch.randn(2, 2, 20, 70) * 2.0
    flow[0] = 0.3                                        # constant flow -> exercises the shuffle-shared taps
    fc, ic = flow.clone().requires_grad_(True), img.clone().requires_grad_(True)
    wc = O.flow_warp(ic, fc); mc = O.photometric_l1(tgt, wc)
    fg, ig = flow.cuda().requires_grad_(True), img.cuda().requires_grad_(True)
    wg, mg = flow_warp_l1(ig, fg, tgt.cuda())
    assert relerr(wg, wc) < RTOL and relerr(mg, mc) < RTOL
    k1, k2 = torch.randn_like(wc), torch.rand_like(mc)
    ((wc * k1).sum() + (mc * k2).sum()).backward()
    ((wg * k1.cuda()).sum() + (mg * k2.cuda()).sum()).backward()
    assert relerr(ig.grad, ic.grad) < 1e-3 and relerr(fg.grad, fc.grad) < 1e-3


def test_sampler(env):
    S, O, dev = env
    from data import FrameStore
    from sin_inn_amd.functional import sample_windows
    st = FrameStore.synthetic(12, 32, 48)
    idx = torch.tensor([3, 7, 5])
    hr, lr = sample_windows(st.hr.cuda(), st.lr.cuda(), idx.cuda(), 2)
    for n, i in enumerate(idx.tolist()):
        h0, l0 = O.gather_window(st.lr, st.hr, i, 2)
        assert torch.equal(hr[n].cpu(), h0) and torch.equal(lr[n].cpu(), l0)


def test_fused_adam_matches_torch(env):
    S, O, dev = env
    torch.manual_seed(4)
    ps = [torch.randn(13, 7), torch.randn(5), torch.randn(3, 3, 3, 3)]
    ref = [p.clone().requires_grad_(True) for p in ps]
    mine = [torch.nn.Parameter(p.clone().cuda()) for p in ps]
    o_ref = torch.optim.Adam(ref, lr=1e-3, betas=(0.9, 0.99), weight_decay=1e-5)
    o_mine = S.FusedAdam(mine, lr=1e-3, betas=(0.9, 0.99), weight_decay=1e-5)
    for _ in range(3):
        o_mine.zero_grad()
        for r, m in zip(ref, mine):
            g = torch.randn_like(r)
            r.grad = g.clone(); m.grad.copy_(g.cuda())
        o_ref.step(); o_mine.step()
    for r, m in zip(ref, mine):
        assert relerr(m, r) < 1e-5


@pytest.mark.parametrize('scale,reduction', [(4, 'mean'), (2, 'mean'), (2, 'sum')])
def test_bayer_bin_bit_exact(env, scale, reduction):
    """datasets/prepare.py LR synthesis: byte-exact against the float64 numpy restatement."""
    S, O, dev = env
    from sin_inn_amd.functional import bayer_bin
    from data import FrameStore
    g = torch.Generator().manual_seed(9)
    hr = torch.randint(0, 256, (3, 32, 48, 3), generator=g, dtype=torch.uint8)
    if reduction == 'sum':
        hr = hr // 4
    lr = bayer_bin(hr.cuda(), scale, reduction)
    want = O.bayer_bin(hr.numpy(), scale, reduction)
    assert np.array_equal(lr.cpu().numpy(), want)
    st = FrameStore.from_hr_clip(hr.cuda(), scale, reduction)
    assert st.lr.shape == (3, 32 // (2 * scale), 48 // (2 * scale), 4)


@pytest.mark.parametrize('reduction', ['mean', 'sum'])
@pytest.mark.parametrize('scale', [1, 2, 4])
def test_bayer_bin_matches_reference_fixture(env, golden_data, scale, reduction):
    """Fixture G8: the bytes the reference's own datasets/prepare.py (extract_bayer + binning + quantisation, :35-82,164) produced
    for these frames -- incl. saturated 'sum' blocks and near-black blocks whose truncation differs from rounding."""
    from sin_inn_amd.functional import bayer_bin
    frames = golden_data['g8_frames']
    want = np.stack([golden_data[f'g8_lr_u8_{t}_{reduction}_{scale}'] for t in range(len(frames))])
    got = bayer_bin(torch.from_numpy(frames).cuda(), scale, reduction)
    assert np.array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize('cin,n,hw', [(24, 256, (16, 32)), (256, 96, (12, 20)), (48, 256, (9, 17)), (8, 24, (5, 7)),
                                      (256, 48, (16, 16)), (192, 256, (40, 24)), (64, 128, (19, 33))])
@pytest.mark.parametrize('cg', [1, 2, 5, 9])    # 32- / 64-column blocks; +4 never / +8 always the 32x32x2 kernel
def test_winograd_conv(env, cin, n, hw, cg):
    """Winograd F(2x2,3x3) kernel against torch conv2d: forward pack + LINEAR/RELU, data-gradient pack + ADD."""
    S, O, dev = env
    from sin_inn_amd import ops, _lib
    try:
        _lib.lib().sininn_conv_test_hooks(cg, 0)
        _winograd_conv_case(dev, ops, _lib, cin, n, hw)
    finally:
        _lib.lib().sininn_conv_test_hooks(0, 0)


def _winograd_conv_case(dev, ops, _lib, cin, n, hw, nb=2):
    torch.manual_seed(cin + n)
    h, w = hw
    conv = torch.nn.Conv2d(cin, n, 3, padding=1)
    x = torch.randn(nb, cin, h, w)
    want = conv(x)
    wf, bf, wd = ops.pack_conv(conv.weight.detach().cuda().contiguous(), conv.bias.detach().cuda().contiguous(), None, True,
                               wino_fwd=True, wino_dgrad=True)
    npk = ops.pad16(n)
    xg = nhwc(x)
    for mode, ref in ((_lib.CONV_LINEAR, want), (_lib.CONV_RELU, F.relu(want))):
        out = torch.full((nb, h, w, n), float('nan'), device=dev)
        ops.conv(in_=ops.ptr(xg), in_stride=cin, Cin=cin, w=ops.ptr(wf), bias=ops.ptr(bf), Np=npk, winograd=1,
                 B=nb, H=h, W=w, ksize=3, mode=mode, out=ops.ptr(out), out_stride=n, N=n)
        assert relerr(nchw(out), ref) < RTOL
    # data gradient: d/dx sum(conv(x) * g) + addend
    g = torch.randn(nb, n, h, w)
    xin = x.clone().requires_grad_(True)
    conv(xin).backward(g)
    add = torch.randn(nb, h, w, cin, device=dev)
    dx = torch.empty((nb, h, w, cin), device=dev)
    ops.conv(in_=ops.ptr(nhwc(g)), in_stride=n, Cin=n, w=ops.ptr(wd), Np=ops.pad32(cin), winograd=1, B=nb, H=h, W=w, ksize=3,
             mode=_lib.CONV_ADD, out=ops.ptr(dx), out_stride=cin, N=cin, addend=ops.ptr(add), addend_stride=cin)
    assert relerr(nchw(dx), xin.grad + nchw(add)) < RTOL


def test_pack_batch_matches_single_packs(env):
    """sininn_pack_batch (one launch for all convs) writes exactly what the per-conv pack entry points write."""
    S, O, dev = env
    from sin_inn_amd import ops
    torch.manual_seed(3)
    cases = [(256, 24, 3, None, True, True), (48, 256, 3, ops.coupling_colmap(24, dev), True, True),
             (256, 96, 1, None, False, False), (192, 256, 1, ops.coupling_colmap(96, dev), False, False),
             (256, 40, 3, None, True, False)]
    singles, descs, batched, keep = [], [], [], []
    for n, cin, k, cmap, wf, wd in cases:
        w = torch.randn(n, cin, k, k, device=dev); b = torch.randn(n, device=dev)
        keep += [w, b]                                 # the descriptors hold raw pointers
        singles.append(ops.pack_conv(w, b, cmap, True, wf, wd))
        out = tuple(torch.full_like(t, float('nan')) for t in singles[-1])
        batched.append(out)
        descs.append(ops.pack_desc(w, b, cmap, out, wf, wd))
    ops.pack_batch_run(ops.pack_batch(descs, dev))
    for (n, cin, k, cmap, wf, wd), a, b in zip(cases, singles, batched):
        for i, (x, y) in enumerate(zip(a, b)):
            assert torch.isfinite(y).all(), (n, cin, k, i)                 # every element written
            if (i == 0 and wf) or (i == 2 and wd):                         # Winograd: G g G^T may contract differently
                assert (x - y).abs().max() <= 1e-6 * x.abs().max(), (n, cin, k, i)
            else:
                assert torch.equal(x, y), (n, cin, k, i)


@pytest.mark.parametrize('scale,reduction', [(4, 'mean'), (2, 'mean'), (1, 'mean'), (2, 'sum')])
def test_bayer_demosaic_preview_is_byte_exact(env, scale, reduction):
    """datasets/prepare.py's lr_frames_demosaiced preview: binned RGGB planes -> mosaic -> bilinear demosaic -> uint8."""
    S, O, dev = env
    from sin_inn_amd.functional import bayer_demosaic
    g = torch.Generator().manual_seed(13 + scale)
    hr = torch.randint(0, 256, (2, 32, 48, 3), generator=g, dtype=torch.uint8)
    if reduction == 'sum':
        hr = hr // 3                              # some sums exceed 1 and are clipped, most are not
    got = bayer_demosaic(hr.cuda(), scale, reduction)
    want = O.bayer_demosaic(hr.numpy(), scale, reduction)
    assert got.shape == (2, 32 // scale, 48 // scale, 3)
    assert np.array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize('cin,n,hw', [(24, 256, (64, 64)), (256, 48, (64, 64)), (96, 256, (32, 32)), (256, 192, (32, 32))])
def test_conv_kernels_at_baseline_config_shapes(env, cin, n, hw):
    """The LINEAR pieces of the step at BASELINE configs[1]'s own shapes (batch 16; level 0: 64x64, level 1: 32x32), with
    the dispatch bench.py gets (no test hooks): 3x3 Winograd forward / data-gradient convs and the weight gradients (per
    conv and grouped), against torch's conv2d and its autograd on the CPU.  These kernels are linear in their inputs, so
    unlike the composed network (whose ReLU gates can flip between two fp32 evaluations) the max-norm bound of the path
    holds at full size."""
    S, O, dev = env
    from sin_inn_amd import ops, _lib
    torch.set_num_threads(min(16, len(__import__('os').sched_getaffinity(0))))
    _winograd_conv_case(dev, ops, _lib, cin, n, hw, nb=16)
    h, w = hw
    torch.manual_seed(n)
    for ksize in (3, 1):
        conv = torch.nn.Conv2d(cin, n, ksize, padding=ksize // 2)
        x = torch.randn(16, cin, h, w)
        g = torch.randn(16, n, h, w) * 0.1
        conv(x).backward(g)
        xg, gg = nhwc(x), nhwc(g)
        gw, gb = torch.zeros_like(conv.weight).cuda(), torch.zeros_like(conv.bias).cuda()
        ops.wgrad(xg, 0, cin, cin, gg, n, n, 16, h, w, ksize, gw, gb)
        assert relerr(gw, conv.weight.grad) < RTOL and relerr(gb, conv.bias.grad) < RTOL
        gw2, gb2 = torch.zeros_like(gw), torch.zeros_like(gb)
        ops.wgrad_group([(xg, 0, cin, cin, gg, 0, n, n, gw2, gb2)], 16, h, w, ksize)
        assert relerr(gw2, conv.weight.grad) < RTOL and relerr(gb2, conv.bias.grad) < RTOL


# ---------------------------------------------------------------------------------------------------
# fused 1x1 conv pair through the C ABI (sininn_conv_pair_k1) against two sininn_conv launches on the same packs
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('co,hw', [(24, (13, 21)), (96, (6, 18)), (8, (9, 33)), (16, (70, 50)), (24, (64, 64))])
def test_conv_pair_c_abi_matches_two_launches(env, co, hw):
    import ctypes as C
    S, O, dev = env
    from sin_inn_amd import _lib, ops
    lib = _lib.lib()
    torch.manual_seed(co)
    b, (h, w) = 2, hw
    c, m = 2 * co, 2 * hw[0] * hw[1]
    x = torch.randn(m, c, device=dev)
    w1 = torch.randn(256 * co, device=dev) * 0.2; b1 = torch.randn(256, device=dev) * 0.1
    w2 = torch.randn(2 * co * 256, device=dev) * 0.05; b2 = torch.randn(2 * co, device=dev) * 0.1

    def args(**kw):
        a = _lib.ConvArgs()
        for k, v in kw.items():
            setattr(a, 'inp' if k == 'in_' else k, v)
        return a

    def run(fused, store_hidden=True):
        hid = torch.zeros(m, 256, device=dev); out = torch.zeros(m, c, device=dev)
        sb = torch.zeros(m, co, device=dev); ld = torch.zeros(b, device=dev)
        common = dict(B=b, H=h, W=w, ksize=1)
        f = args(in_=ops.ptr(x, co), in_stride=c, Cin=co, w=ops.ptr(w1), bias=ops.ptr(b1), Np=256, mode=_lib.CONV_RELU,
                 out=ops.ptr(hid) if store_hidden else None, out_stride=256, N=256, **common)
        s = args(in_=ops.ptr(hid), in_stride=256, Cin=256, w=ops.ptr(w2), bias=ops.ptr(b2), Np=2 * co, mode=_lib.CONV_COUPLE_FWD,
                 out=ops.ptr(out), out_stride=c, v=ops.ptr(x), v_stride=c, sbuf=ops.ptr(sb), logdet=ops.ptr(ld), Co=co,
                 clamp=1.2, col_tile=ops.coupling_tile(co), **common)
        if fused:
            assert lib.sininn_conv_pair_k1_supported(C.byref(f), C.byref(s)) == 1
            _lib.check(lib.sininn_conv_pair_k1(C.byref(f), C.byref(s), ops._stream()))
        else:
            _lib.check(lib.sininn_conv(C.byref(f), ops._stream()))
            _lib.check(lib.sininn_conv(C.byref(s), ops._stream()))
        return hid, out[:, :co], sb, ld

    ref = run(False)
    got = run(True)
    for a, r in zip(got, ref):
        assert relerr(a, r) < 1e-5
    ng = run(True, store_hidden=False)                     # no-grad form: the hidden tensor never reaches HBM
    assert float(ng[0].abs().max()) == 0.0
    for a, r in zip(ng[1:], ref[1:]):
        assert relerr(a, r) < 1e-5
    # the persistent twin for the level-0 shapes (sininn_conv_sub1_fwd): same outputs, s and log-det, both coupling directions
    for mode in (_lib.CONV_COUPLE_FWD, _lib.CONV_COUPLE_INV):
        outs = []
        for persistent in (False, True):
            out = torch.zeros(m, c, device=dev); y2 = torch.zeros(m, co, device=dev)
            sb = torch.zeros(m, co, device=dev); ld = torch.zeros(b, device=dev)
            common = dict(B=b, H=h, W=w, ksize=1)
            f = args(in_=ops.ptr(x, co), in_stride=c, Cin=co, w=ops.ptr(w1), bias=ops.ptr(b1), Np=256, mode=_lib.CONV_RELU, out_stride=256,
                     N=256, **common)
            s = args(in_=ops.ptr(x), in_stride=256, Cin=256, w=ops.ptr(w2), bias=ops.ptr(b2), Np=2 * co, mode=mode, out=ops.ptr(out),
                     out_stride=c, v=ops.ptr(x), v_stride=c, sbuf=ops.ptr(sb), logdet=ops.ptr(ld), Co=co, clamp=1.2, out2=ops.ptr(y2),
                     out2_stride=co, col_tile=ops.coupling_tile(co), **common)
            if persistent:
                assert lib.sininn_conv_sub1_fwd_supported(C.byref(f), C.byref(s)) == (1 if co in (8, 16, 24) else 0)
                if co not in (8, 16, 24):
                    assert lib.sininn_conv_sub1_fwd(C.byref(f), C.byref(s), ops._stream()) != 0
                    continue
                _lib.check(lib.sininn_conv_sub1_fwd(C.byref(f), C.byref(s), ops._stream()))
            else:
                _lib.check(lib.sininn_conv_pair_k1(C.byref(f), C.byref(s), ops._stream()))
            outs.append((out[:, :co], y2, sb, ld))
        if len(outs) == 2:
            for a, r in zip(outs[1], outs[0]):
                assert relerr(a, r) < 1e-5
    # a 3x3 first conv is not a pair
    f3 = args(in_=ops.ptr(x, co), in_stride=c, Cin=co, w=ops.ptr(w1), bias=ops.ptr(b1), Np=256, mode=_lib.CONV_RELU, out_stride=256,
              N=256, B=b, H=h, W=w, ksize=3)
    s1 = args(in_=ops.ptr(x), in_stride=256, Cin=256, w=ops.ptr(w2), Np=2 * co, mode=_lib.CONV_LINEAR, B=b, H=h, W=w, ksize=1)
    assert lib.sininn_conv_pair_k1_supported(C.byref(f3), C.byref(s1)) == 0


# ---------------------------------------------------------------------------------------------------
# the whole backward of a 1x1 subnet in one persistent launch (sininn_conv_sub1_bwd): h recomputed, dh on chip
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('co,b,hw,no_dx', [(24, 2, (13, 21), False), (24, 16, (64, 64), False), (8, 3, (9, 33), False), (16, 1, (2, 16), False),
                                           (24, 2, (7, 40), True)])
def test_fused_1x1_subnet_backward_c_abi(env, co, b, hw, no_dx):
    """dx: BITWISE what the data-gradient pair produces from the h the forward pair stored (the recompute uses the same k order);
    weight / bias gradients: the torch matmul reference at 1e-4 (max-norm), accumulated (+=) onto non-zero gradients; partial
    tiles (odd sizes), more tiles than persistent blocks (16 x 64 x 64 = 2048 tiles on 512 blocks), a single tile."""
    import ctypes as C
    S, O, dev = env
    from sin_inn_amd import _lib, ops
    lib = _lib.lib()
    torch.manual_seed(co + b)
    h, w = hw
    k1, k2, m = co, 2 * co, b * h * w
    cx = 2 * co + 8                                               # x lives inside a wider tensor (channel offset 8)
    xfull = torch.randn(m, cx, device=dev)
    conv1 = torch.nn.Conv2d(k1, 256, 1).to(dev)
    conv2 = torch.nn.Conv2d(256, k2, 1).to(dev)
    with torch.no_grad():
        conv2.weight.mul_(0.3)
    pk1 = ops.pack_conv(conv1.weight.detach(), conv1.bias.detach(), None, True)
    pk2 = ops.pack_conv(conv2.weight.detach(), conv2.bias.detach(), ops.coupling_colmap(co, dev), True)
    dr = torch.randn(m, k2, device=dev)
    addend = torch.randn(m, k1, device=dev)

    def args(**kw):
        a = _lib.ConvArgs()
        for k, v in kw.items():
            setattr(a, 'inp' if k == 'in_' else k, v)
        return a
    common = dict(B=b, H=h, W=w, ksize=1)
    # the h the forward pass would have used: stage 1 of the pair kernel (second conv: a throw-away linear conv)
    hid = torch.zeros(m, 256, device=dev)
    dump = torch.zeros(m, k2, device=dev)
    f = args(in_=ops.ptr(xfull, 8), in_stride=cx, Cin=k1, w=ops.ptr(pk1[0]), bias=ops.ptr(pk1[1]), Np=256, mode=_lib.CONV_RELU,
             out=ops.ptr(hid), out_stride=256, N=256, **common)
    s2 = args(in_=ops.ptr(hid), in_stride=256, Cin=256, w=ops.ptr(pk2[0]), bias=ops.ptr(pk2[1]), Np=k2, mode=_lib.CONV_LINEAR,
              out=ops.ptr(dump), out_stride=k2, N=k2, **common)
    _lib.check(lib.sininn_conv_pair_k1(C.byref(f), C.byref(s2), ops._stream()))

    def dgrad_descs(dx, dh):
        d2 = args(in_=ops.ptr(dr), in_stride=k2, Cin=k2, w=ops.ptr(pk2[2]), Np=256, mode=_lib.CONV_MASK, out=ops.ptr(dh), out_stride=256,
                  N=256, mask=ops.ptr(hid), mask_stride=256, **common)
        d1 = args(in_=ops.ptr(dh), in_stride=256, Cin=256, w=ops.ptr(pk1[2]), Np=ops.pad16(k1), mode=_lib.CONV_ADD, out=ops.ptr(dx),
                  out_stride=k1, N=k1, addend=ops.ptr(addend), addend_stride=k1, **common)
        return d2, d1
    dx_ref, dh_ref = torch.zeros(m, k1, device=dev), torch.zeros(m, 256, device=dev)
    d2, d1 = dgrad_descs(dx_ref, dh_ref)
    _lib.check(lib.sininn_conv_pair_k1(C.byref(d2), C.byref(d1), ops._stream()))

    g0 = [torch.randn_like(conv2.weight), torch.randn_like(conv2.bias), torch.randn_like(conv1.weight), torch.randn_like(conv1.bias)]
    gw2, gb2, gw1, gb1 = (g.clone().contiguous() for g in g0)
    dx = torch.full((m, k1), float('nan'), device=dev)
    rc = args(in_=ops.ptr(xfull, 8), in_stride=cx, Cin=k1, w=ops.ptr(pk1[0]), bias=ops.ptr(pk1[1]), Np=256, **common)
    d2f, d1f = dgrad_descs(dx, dh_ref)
    d2f.mask, d2f.out, d1f.inp = None, None, None                 # ignored by the fused kernel: h is recomputed, dh stays on chip
    nbytes = lib.sininn_conv_sub1_bwd_workspace_bytes(k1, co)
    assert nbytes > 0 and lib.sininn_conv_sub1_bwd_workspace_bytes(96, 96) == 0
    ws = torch.empty(nbytes // 4, device=dev)
    _lib.check(lib.sininn_conv_sub1_bwd(C.byref(rc), C.byref(d2f), C.byref(d1f), int(no_dx), ops.ptr(gw2), ops.ptr(gb2), ops.ptr(gw1),
                                        ops.ptr(gb1), ops.ptr(ws), nbytes, ops._stream()))
    torch.cuda.synchronize()
    if no_dx:
        assert bool(torch.isnan(dx).all())                         # untouched
    else:
        assert torch.equal(dx, dx_ref)
    x = xfull[:, 8:8 + k1].double()
    hd, dhd, drd = hid.double(), dh_ref.double(), dr.double()
    want = [g0[0].double() + (drd.t() @ hd).reshape(k2, 256, 1, 1), g0[1].double() + drd.sum(0),
            g0[2].double() + (dhd.t() @ x).reshape(256, k1, 1, 1), g0[3].double() + dhd.sum(0)]
    for got, ref_ in zip((gw2, gb2, gw1, gb1), want):
        assert relerr(got, ref_.float()) < 1e-4
    # unsupported shapes are refused, not mis-run
    bad = args(in_=ops.ptr(xfull, 8), in_stride=cx, Cin=k1, w=ops.ptr(pk1[0]), bias=ops.ptr(pk1[1]), Np=256, B=b, H=h, W=w, ksize=3)
    assert lib.sininn_conv_sub1_bwd(C.byref(bad), C.byref(d2f), C.byref(d1f), 0, None, None, None, None, ops.ptr(ws), nbytes, ops._stream()) != 0


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_sample_pairs_planar_is_bit_exact(dtype):
    """frame-pair sampler of the flow path: (clip[idx], clip[idx + gap]) / 255 as planar (n,3,H,W), fp32 bit-exact with the
    reference's FloatTensor / 255 (true division), bf16 = the correctly rounded value of that; indices past the clip clamp."""
    from sin_inn_amd.functional import sample_pairs
    g = torch.Generator().manual_seed(3)
    clip = torch.randint(0, 256, (9, 12, 20, 3), generator=g, dtype=torch.uint8)
    idx = torch.tensor([0, 4, 7, 8, 3], dtype=torch.int32)
    for gap in (1, 2):
        a, b = sample_pairs(clip.cuda(), idx.cuda(), gap=gap, dtype=dtype)
        want_a = (clip[idx.long()].float() / 255.).permute(0, 3, 1, 2)
        want_b = (clip[(idx.long() + gap).clamp(max=8)].float() / 255.).permute(0, 3, 1, 2)
        assert a.dtype == dtype and a.is_contiguous() and a.shape == (5, 3, 12, 20)
        assert torch.equal(a.cpu(), want_a.to(dtype)) and torch.equal(b.cpu(), want_b.to(dtype))
