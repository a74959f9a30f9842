"""CPU: the flow-loss oracle (oracle/flow_oracle.py) against fixtures produced by the reference's own code
(tests/golden/golden_flow.npz <- video-interpolation/my_utils/{occlusions,loss}.py), plus the properties that anchor the
softsplat restatement (the reference's softsplat needs cupy and cannot run in the build container)."""
import os

import numpy as np
import pytest
import torch

from oracle import flow_oracle as FO

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope='module')
def gold():
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(HERE, 'golden', 'golden_flow.npz')).items()}


def test_occlusion_wang_matches_reference(gold):
    b, _, h, w = gold['f1_flow21'].shape
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing='ij')
    grid = torch.stack([xs, ys], 0)[None].expand(b, 2, h, w)
    corr = FO.get_corresponding_map(grid + gold['f1_flow21'])
    assert torch.allclose(corr, gold['f1_corr'], rtol=1e-6, atol=1e-6)
    mask = FO.occlusion_wang(gold['f1_flow12'], gold['f1_flow21'], 0.7)
    assert torch.equal(mask, gold['f1_mask'])
    assert 0 < mask.mean() < 1                      # the fixture exercises both outcomes


@pytest.mark.parametrize('md', [2, 3])
def test_census_matches_reference(gold, md):
    im = gold[f'f2_{md}_im'].clone().requires_grad_(True)
    imw = gold[f'f2_{md}_imw'].clone().requires_grad_(True)
    loss = FO.census_loss(im, imw, gold[f'f2_{md}_mask'], 0.1, md)
    loss.backward()
    assert abs(float(loss) / float(gold[f'f2_{md}_loss']) - 1) < 1e-5
    for got, want in ((im.grad, gold[f'f2_{md}_gim']), (imw.grad, gold[f'f2_{md}_gimw'])):
        assert float((got - want).abs().max() / want.abs().max()) < 1e-4


def test_softsplat_properties():
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 3, 9, 11, generator=g)
    # zero flow is the identity; an integer shift moves pixels and drops what leaves the image
    assert torch.allclose(FO.softsplat_sum(x, torch.zeros(2, 2, 9, 11)), x)
    shift = torch.zeros(2, 2, 9, 11); shift[:, 0] = 2; shift[:, 1] = -1
    y = FO.softsplat_sum(x, shift)
    assert torch.allclose(y[:, :, :-1, 2:], x[:, :, 1:, :-2]) and float(y[:, :, -1].abs().max()) == 0
    # mass conservation away from the border: the four weights of a tap sum to one
    flow = torch.randn(2, 2, 9, 11, generator=g) * 0.4
    inner = torch.zeros(2, 3, 9, 11); inner[:, :, 2:-2, 2:-2] = x[:, :, 2:-2, 2:-2]
    assert torch.allclose(FO.softsplat_sum(inner, flow).sum((2, 3)), inner.sum((2, 3)), rtol=1e-5, atol=1e-5)
    # softmax mode: a constant image stays constant wherever anything lands
    ones = torch.ones(2, 3, 9, 11)
    sm = FO.function_softsplat(ones, flow, torch.randn(2, 1, 9, 11, generator=g), 'softmax')
    assert torch.allclose(sm[sm != 0], torch.ones_like(sm[sm != 0]), rtol=1e-5)


def test_masked_losses_with_trainer_masks_match_reference(gold):
    """3-channel masks (trainer.py:64) and 1-channel masks: CensusLoss and L1Loss values + gradients."""
    for tag, fn, mk in (('census3', lambda a, b, m: FO.census_loss(a, b, m, 0.1, 3), 'f3_mask3'),
                        ('l1_3', lambda a, b, m: FO.l1_loss(a, b, m, 1), 'f3_mask3'),
                        ('l1_1', lambda a, b, m: FO.l1_loss(a, b, m, 0.7), 'f3_mask1'),
                        ('ssim1', lambda a, b, m: FO.ssim_loss(a, b, m, 0.4, 1), 'f3_mask1'),
                        ('ssim3', lambda a, b, m: FO.ssim_loss(a, b, m, 0.4, 1), 'f3_mask3'),
                        ('ssim1_md2', lambda a, b, m: FO.ssim_loss(a, b, m, 0.4, 2), 'f3_mask1')):
        im = gold['f3_im'].clone().requires_grad_(True)
        imw = gold['f3_imw'].clone().requires_grad_(True)
        loss = fn(im, imw, gold[mk])
        loss.backward()
        assert abs(float(loss) / float(gold[f'f3_{tag}_loss']) - 1) < 1e-5
        for got, want in ((im.grad, gold[f'f3_{tag}_gim']), (imw.grad, gold[f'f3_{tag}_gimw'])):
            assert float((got - want).abs().max() / want.abs().max()) < 1e-4


@pytest.mark.parametrize('fun,k', [('gauss', 150.0), ('exp', 20.0)])
@pytest.mark.parametrize('order', [1, 2])
def test_bilateral_smooth_matches_reference(gold, fun, k, order):
    flow = gold['f4_flow'].clone().requires_grad_(True)
    loss = FO.bilateral_smooth(gold['f4_img'], flow, 0.1, fun, k, order)
    loss.backward()
    assert abs(float(loss) / float(gold[f'f4_{fun}_{order}_loss']) - 1) < 1e-5
    want = gold[f'f4_{fun}_{order}_gflow']
    assert float((flow.grad - want).abs().max() / want.abs().max()) < 1e-4


def test_resample2d_and_metric_match_reference(gold):
    """F6: the oracle's flow_warp / photometric_l1 against Resample2d.forward + the trainer's metric as the reference's own
    code evaluates them (resample2d.py:52-72, trainer.py:61-62), values and both gradients."""
    from oracle import sininn_oracle as O
    img = gold['f6_img'].clone().requires_grad_(True)
    flow = gold['f6_flow'].clone().requires_grad_(True)
    warped = O.flow_warp(img, flow)
    metric = O.photometric_l1(gold['f6_tgt'], warped)
    assert torch.allclose(warped, gold['f6_warped'], rtol=1e-6, atol=1e-6)
    assert torch.allclose(metric, gold['f6_metric'], rtol=1e-6, atol=1e-6)
    ((warped * gold['f6_gw']).sum() + (metric * gold['f6_gm']).sum()).backward()
    assert torch.allclose(img.grad, gold['f6_gimg'], rtol=1e-5, atol=1e-6)
    assert torch.allclose(flow.grad, gold['f6_gflow'], rtol=1e-5, atol=1e-5)
    zero = O.flow_warp(gold['f6_img'], torch.zeros_like(gold['f6_flow']))
    assert torch.allclose(zero, gold['f6_zero_flow_warped'], rtol=1e-6, atol=1e-6)
    assert not torch.allclose(zero, gold['f6_img'], atol=1e-3)           # quirk C-18 is in the fixture


def test_flow_warp_oracle_on_bf16_representable_operands_f6b(gold):
    """F6b: the same operator on bf16-representable images, the metric taken on the warped value rounded to bf16 (what the
    mixed-precision kernels store): oracle == the reference's own evaluation."""
    from oracle import sininn_oracle as O
    img = gold['f6b_img'].clone().requires_grad_(True)
    flow = gold['f6b_flow'].clone().requires_grad_(True)
    warped = O.flow_warp(img, flow)
    metric = O.photometric_l1(gold['f6b_tgt'], O.bf16_round(warped))
    assert torch.allclose(warped, gold['f6b_warped'], rtol=1e-6, atol=1e-6)
    assert torch.allclose(metric, gold['f6b_metric'], rtol=1e-6, atol=1e-6)
    ((warped * gold['f6b_gw']).sum() + (metric * gold['f6b_gm']).sum()).backward()
    assert torch.allclose(img.grad, gold['f6b_gimg'], rtol=1e-5, atol=1e-6)
    assert torch.allclose(flow.grad, gold['f6b_gflow'], rtol=1e-5, atol=1e-5)


def test_occlusion_brox_matches_reference(gold):
    want = gold['f6_brox'].bool()
    got = FO.occlusion_brox(gold['f6_fw'], gold['f6_bw'], 0.5)
    assert got.shape == want.shape and torch.equal(got, want)
    assert 0.02 < float(want.float().mean()) < 0.98
