"""CPU: the oracle restatement against the fixtures produced by the reference's own code
(tests/golden/make_golden.py) and against known-answer properties for the unpinned parts."""
import math
import numpy as np
import pytest
import torch

from oracle import sininn_oracle as O

T = torch.from_numpy


def test_g1_losses(golden):
    x, y = T(golden['g1_x']), T(golden['g1_y'])
    assert torch.allclose(O.reconstruction(x, y), T(golden['g1_rec']), rtol=1e-6, atol=0)
    assert torch.allclose(O.latent_nll(x), T(golden['g1_nll']), rtol=1e-6, atol=0)


def test_g2_haar(golden):
    x = T(golden['g2_x'])
    y = O.haar_fwd(x)
    assert torch.allclose(y, T(golden['g2_fwd']), rtol=1e-6, atol=1e-7)
    assert torch.allclose(O.haar_inv(y), T(golden['g2_rev']), rtol=1e-6, atol=1e-6)
    assert math.isclose(O.haar_last_jac(x.shape[1:]), float(golden['g2_jac_fwd']), rel_tol=1e-12)
    assert math.isclose(O.haar_last_jac(y.shape[1:], rev=True), float(golden['g2_jac_rev']), rel_tol=1e-12)


@pytest.mark.parametrize('tag,k', [('3x3', 3), ('1x1', 1)])
def test_g3_subnets(golden, tag, k):
    net = O.make_subnet(24, 48, k)
    with torch.no_grad():
        net[0].weight.copy_(T(golden[f'g3_{tag}_w0'])); net[0].bias.copy_(T(golden[f'g3_{tag}_b0']))
        net[2].weight.copy_(T(golden[f'g3_{tag}_w2'])); net[2].bias.copy_(T(golden[f'g3_{tag}_b2']))
        y = net(T(golden[f'g3_{tag}_x']))
    assert torch.allclose(y, T(golden[f'g3_{tag}_y']), rtol=1e-5, atol=1e-6)


def test_g3_default_init_matches_reference(golden):
    # same constructor order under the same seed -> identical weights (what "same random-init weights" relies on)
    torch.manual_seed(3)
    net = O.make_subnet(24, 48, 3)
    assert torch.equal(net[0].weight.detach(), T(golden['g3_3x3_w0']))
    assert torch.equal(net[2].bias.detach(), T(golden['g3_3x3_b2']))


def test_g4_invblock(golden):
    blk = O.InvBlockExpOracle(8, 4)
    sd = {}
    for k in golden.files:
        if k.startswith('g4_sd_'):
            name = k[len('g4_sd_'):]
            sub, conv, leaf = name.split('.')
            sd[f'{sub}.convs.{int(conv[4:]) - 1}.{leaf}'] = T(golden[k])
    blk.load_state_dict(sd)
    x = T(golden['g4_x'])
    with torch.no_grad():
        y = blk(x)
        assert torch.allclose(blk.F(x[:, 4:]), T(golden['g4_dense_F']), rtol=1e-5, atol=1e-6)
        assert torch.allclose(y, T(golden['g4_fwd']), rtol=1e-5, atol=1e-6)
        assert torch.allclose(blk(y, rev=True), T(golden['g4_rev']), rtol=1e-4, atol=1e-5)


def test_g6_permutations(golden):
    for c in (48, 192):
        for k in range(12):
            perm, inv = O.permutation(c, k)
            assert np.array_equal(perm, golden[f'g6_perm_{c}_{k}'])
            assert np.array_equal(perm[inv], np.arange(c))


# ---- known-answer tests for the unpinned (FrEIA / kornia) restatements -------------------------
def test_squeeze_formula_and_roundtrip():
    x = torch.arange(2 * 3 * 4 * 6, dtype=torch.float32).reshape(2, 3, 4, 6)
    y = O.squeeze_fwd(x)
    for hb in range(2):
        for wb in range(2):
            for c in range(3):
                assert torch.equal(y[:, (hb * 2 + wb) * 3 + c], x[:, c, hb::2, wb::2])
    assert torch.equal(O.squeeze_inv(y), x)


@pytest.mark.parametrize('k', [1, 3])
def test_glow_roundtrip_and_logdet(k):
    torch.manual_seed(0)
    blk = O.GlowBlock(4, k, clamp=1.2).double()
    x = torch.randn(1, 4, 3, 3, dtype=torch.float64)
    y = blk(x)
    jac_fwd = blk.last_jac.clone()
    xr = blk(y, rev=True)
    assert (xr - x).abs().max() < 1e-10
    assert torch.allclose(blk.last_jac, -jac_fwd)
    J = torch.autograd.functional.jacobian(lambda v: blk(v.reshape(1, 4, 3, 3)).reshape(-1), x.reshape(-1))
    assert torch.allclose(torch.linalg.slogdet(J)[1], jac_fwd[0], rtol=1e-8, atol=1e-8)


def test_srflow_shapes_and_roundtrip():
    torch.manual_seed(0)
    net = O.SRFlowOracle(3, 32, 32, scale=4, num_coupling=2)
    x = torch.rand(2, 3, 32, 32)
    with torch.no_grad():
        y = net(x)
        assert y.shape == (2, 192, 4, 4)
        assert (net(y, rev=True) - x).abs().max() < 1e-4
    keys = list(net.state_dict().keys())
    assert keys[0] == 'module_list.3.s1.0.weight'
    n = sum(p.numel() for p in O.SRFlowOracle(3, 64, 64, num_coupling=4).parameters())
    assert n == 3692416                                  # SRF -c 4


def test_mmd_hand_case():
    # b=2, one feature: x=[0,1], y=[0,3]; fwd kernels; computed by hand from loss.py:31-36
    x = torch.tensor([[0.0], [1.0]]).reshape(2, 1, 1, 1)
    y = torch.tensor([[0.0], [3.0]]).reshape(2, 1, 1, 1)
    def k(d):
        return sum(c ** a * ((c + d) / a) ** (-a) for c, a in O.MMD_KERNELS_FWD)
    dxx = np.array([[0, 1], [1, 0.0]]); dyy = np.array([[0, 9], [9, 0.0]]); dxy = np.array([[0, 9], [1, 4.0]])
    want = np.mean(np.vectorize(k)(dxx) + np.vectorize(k)(dyy) - 2 * np.vectorize(k)(dxy))
    assert math.isclose(float(O.mmd(x, y)), want, rel_tol=1e-5)


@pytest.mark.parametrize('tag', ['a', 'b'])
@pytest.mark.parametrize('rev', [False, True])
def test_g7_mmd_matches_reference(golden, tag, rev):
    """loss.mmd as the reference itself computes it (loss.py:9-36 run on CPU by make_golden.py), both kernel sets."""
    r = 'rev' if rev else 'fwd'
    x = torch.from_numpy(golden[f'g7_{tag}_x']).requires_grad_(True)
    y = torch.from_numpy(golden[f'g7_{tag}_y']).requires_grad_(True)
    val = O.mmd(x, y, rev=rev)
    val.backward()
    want = float(golden[f'g7_{tag}_{r}'])
    assert abs(float(val) - want) <= 1e-6 * max(1.0, abs(want))
    for got, key in ((x.grad, 'gx'), (y.grad, 'gy')):
        w = torch.from_numpy(golden[f'g7_{tag}_{r}_{key}'])
        assert float((got - w).abs().max() / w.abs().max()) < 1e-5


def test_tcr_identity_and_translation():
    img = torch.rand(1, 2, 8, 8)
    rand = torch.tensor([[0.5, 0.5, 0.5]])               # zero angle, zero shift
    out = O.tcr_warp(img, rand, 5.0, 5.0)
    # kornia's (W-1) normalisation mixed with align_corners=False sampling: identity up to a sub-pixel rescale
    assert out.shape == img.shape
    m = O.tcr_matrix(torch.tensor([[0.5, 1.0, 0.0]]), 8, 8, 5.0, 5.0, scale=0.25)
    assert torch.allclose(m[0, :, 2], torch.tensor([20.0, -20.0]))        # quirk C-6: LR shift is x4


def test_flow_warp_zero_flow_is_not_identity_quirk():
    img = torch.rand(1, 3, 6, 6)
    out = O.flow_warp(img, torch.zeros(1, 2, 6, 6))
    assert out.shape == img.shape
    # source x = x'*W/(W-1) - 0.5 (SURVEY C-18): pixel 0 samples at -0.5 -> half weight on zero padding
    assert torch.allclose(out[0, :, 0, 0], img[0, :, 0, 0] * 0.25, atol=1e-6)


def test_sampler_indices():
    assert O.train_indices(240, 10)[:3] == [11, 23, 35]
    assert O.train_indices(7, 1) == [2]                  # config 1: 8 frames, fps 1
    assert O.all_indices(7, 1) == [2, 3, 4, 5]
    perm = list(range(240 - 20))
    v = O.val_indices(240, 10, 10, 5, perm)
    assert all((i + 13) % 12 != 0 for i in v) and len(v) == 5


def test_adam_matches_torch():
    torch.manual_seed(1)
    p = torch.randn(50); g = torch.randn(50)
    q = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([q], lr=1e-4, betas=(0.9, 0.99), weight_decay=1e-5)
    m, v = torch.zeros(50), torch.zeros(50)
    for step in (1, 2, 3):
        q.grad = g.clone(); opt.step()
        O.adam_step(p, g, m, v, step, 1e-4, 0.9, 0.99, 1e-8, 1e-5)
    assert torch.allclose(p, q.detach(), rtol=1e-6, atol=1e-7)


def test_g5_irn_full_net(golden):
    # weights regenerated exactly as make_golden.py did: same seed, same constructor + init order as
    # archs.py:201-222 / 74-133, then conv5 of every DenseBlock re-drawn from generator 55
    torch.manual_seed(5)
    net = O.IRNOracle(3, 12, scale=4, num_coupling=4)
    haar_params = 4 * 4 * (3 + 12 + 48)                  # reference registers haar_weights as nn.Parameter
    assert sum(p.numel() for p in net.parameters()) + haar_params == int(golden['g5_nparams'])
    g5 = torch.Generator().manual_seed(55)
    for m in net.modules():
        if isinstance(m, O.DenseBlockOracle):
            m.convs[4].weight.data = torch.randn(m.convs[4].weight.shape, generator=g5) * 0.02
    x = T(golden['g5_x'])
    with torch.no_grad():
        y = net(x)
        xr = net(y, rev=True)
    assert torch.allclose(y[:, ::16, ::2, ::2], T(golden['g5_out_slice']), rtol=1e-4, atol=1e-5)
    assert torch.allclose(y.norm(), T(golden['g5_out_norm']), rtol=1e-5)
    assert (xr - x).abs().max() < 1e-4


def test_bayer_bin_oracle_hand_case():
    # one 8x8 frame, scale 2: R plane = channel 0 at even rows / even cols
    hr = np.zeros((1, 8, 8, 3), np.uint8)
    hr[0, ::2, ::2, 0] = np.array([[10, 20, 30, 40], [50, 60, 70, 80], [1, 2, 3, 4], [5, 6, 7, 9]])
    lr = O.bayer_bin(hr, scale=2)
    assert lr.shape == (1, 2, 2, 4)
    assert lr[0, :, :, 0].tolist() == [[35, 55], [3, 5]]          # floor(mean); (1+2+5+6)/4=3.5 -> 3, (3+4+7+9)/4=5.75 -> 5
    assert int(lr[0, :, :, 1:].sum()) == 0


def test_forced_gates_with_the_oracles_own_gates_change_nothing():
    """checker feature used by tests/test_gpu_gates.py: forcing the gates an evaluation took itself reproduces it exactly
    (outputs and gradients), for the GLOW subnets (ReLU) and the DenseBlock (LeakyReLU 0.2)."""
    import torch
    import torch.nn.functional as F
    from oracle import sininn_oracle as O
    torch.manual_seed(0)
    blk = O.GlowBlock(16, 3).double()
    x = torch.randn(2, 16, 6, 7, dtype=torch.float64)
    for rev in (False, True):
        blk.forced_gates = None
        xa = x.clone().requires_grad_(True)
        ya = blk(xa, rev=rev)
        ya.square().sum().backward()
        x1, x2 = x[:, :8], x[:, 8:]
        with torch.no_grad():       # inputs of the two subnets in this direction
            if not rev:
                in2 = x2; in1 = ya[:, :8]
            else:
                in1 = x1; in2 = ya[:, 8:]
            gates = {'s1': blk.s1[0](in1) > 0, 's2': blk.s2[0](in2) > 0}
        ga = [p.grad.clone() for p in blk.parameters()]
        blk.zero_grad(); blk.forced_gates = gates
        xb = x.clone().requires_grad_(True)
        yb = blk(xb, rev=rev)
        yb.square().sum().backward()
        assert torch.allclose(ya, yb, rtol=0, atol=1e-12) and torch.allclose(xa.grad, xb.grad, rtol=0, atol=1e-10)
        for a, p in zip(ga, blk.parameters()):
            assert torch.allclose(a, p.grad, rtol=0, atol=1e-10)
        blk.zero_grad()
    dense = O.DenseBlockOracle(8, 4).double()
    dense.convs[4].weight.data.normal_(0, 0.05)
    x = torch.randn(2, 8, 5, 6, dtype=torch.float64, requires_grad=True)
    ya = dense(x)
    feats, gates = [x.detach()], []
    with torch.no_grad():
        for i in range(4):
            pre = dense.convs[i](torch.cat(feats, 1))
            gates.append(pre > 0)
            feats.append(F.leaky_relu(pre, 0.2))
    dense.forced_gates = gates
    assert torch.allclose(ya, dense(x), rtol=0, atol=1e-12)
