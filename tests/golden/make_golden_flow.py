"""Generate golden_flow.npz FROM THE REFERENCE'S OWN CODE (flow trainer utilities).

Run once in the build container (needs /root/reference; never runs on the GPU box):
    python tests/golden/make_golden_flow.py
Imports video-interpolation/my_utils/occlusions.py and my_utils/loss.py unmodified (pure torch; occlusions.py's
`Resample2d` import resolves because resample2d.py only references its absent CUDA extension inside the dead
`Resample2d_old` class; the live `Resample2d` is grid_sample and runs on CPU: fixture F6).  softsplat.py needs cupy and is NOT imported (its oracle is pinned by properties instead).
Outputs are data only: inputs and expected outputs.
"""
import os, sys
import numpy as np
import torch

REF = '/root/reference/video-interpolation'
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    sys.path.insert(0, REF)
    import my_utils.occlusions as ref_occ          # noqa: E402
    import my_utils.loss as ref_loss               # noqa: E402
    sys.path.pop(0)
    g = torch.Generator().manual_seed(21)
    out = {}
    # F1 occlusion_wang (occlusions.py:96-103) + its range map (occlusions.py:29-77); flows that leave the image too
    flow12 = torch.randn(2, 2, 14, 19, generator=g) * 2.5
    flow21 = torch.randn(2, 2, 14, 19, generator=g) * 2.5
    flow21[0, :, :2] *= 4
    base = ref_occ.mesh_grid(2, 14, 19).type_as(flow21)
    out.update(f1_flow12=flow12, f1_flow21=flow21, f1_corr=ref_occ.get_corresponding_map(base + flow21),
               f1_mask=ref_occ.occlusion_wang(flow12, flow21, 0.7))
    # F2 CensusLoss (loss.py:30-72) for the reference default (2) and the trainer's (3) max_distance, values + gradients
    for md in (2, 3):
        im = torch.rand(2, 3, 20, 23, generator=g).requires_grad_(True)
        im_w = (im.detach() + 0.08 * torch.randn(2, 3, 20, 23, generator=g)).clamp(0, 1).requires_grad_(True)
        mask = (torch.rand(2, 1, 20, 23, generator=g) > 0.25).float()
        loss = ref_loss.CensusLoss(0.1, max_distance=md)(im, im_w, mask)
        loss.backward()
        out.update({f'f2_{md}_im': im.detach(), f'f2_{md}_imw': im_w.detach(), f'f2_{md}_mask': mask,
                    f'f2_{md}_loss': loss.detach(), f'f2_{md}_gim': im.grad, f'f2_{md}_gimw': im_w.grad})
    # F3 CensusLoss / L1Loss with the trainer's 3-channel mask (trainer.py:64: occlusion mask * (softmax != 0))
    im = torch.rand(2, 3, 18, 21, generator=g).requires_grad_(True)
    im_w = (im.detach() + 0.08 * torch.randn(2, 3, 18, 21, generator=g)).clamp(0, 1).requires_grad_(True)
    mask3 = (torch.rand(2, 3, 18, 21, generator=g) > 0.25).float()
    mask1 = (torch.rand(2, 1, 18, 21, generator=g) > 0.25).float()
    out.update(f3_im=im.detach(), f3_imw=im_w.detach(), f3_mask3=mask3, f3_mask1=mask1)
    for tag, fn, m in (('census3', ref_loss.CensusLoss(0.1, max_distance=3), mask3), ('l1_3', ref_loss.L1Loss(1), mask3),
                       ('l1_1', ref_loss.L1Loss(0.7), mask1)):
        im.grad = im_w.grad = None
        loss = fn(im, im_w, m)
        loss.backward()
        out.update({f'f3_{tag}_loss': loss.detach(), f'f3_{tag}_gim': im.grad.clone(), f'f3_{tag}_gimw': im_w.grad.clone()})
    # F5 SSIMLoss (loss.py:75-103), 1- and 3-channel masks, md 1 and 2
    for tag, m, md in (('ssim1', mask1, 1), ('ssim3', mask3, 1), ('ssim1_md2', mask1, 2)):
        im.grad = im_w.grad = None
        loss = ref_loss.SSIMLoss(0.4, md=md)(im, im_w, m)
        loss.backward()
        out.update({f'f3_{tag}_loss': loss.detach(), f'f3_{tag}_gim': im.grad.clone(), f'f3_{tag}_gimw': im_w.grad.clone()})
    # F4 BilateralSmooth (loss.py:106-132): both edge functions and orders; gradient w.r.t. the flow
    img = torch.rand(2, 3, 17, 22, generator=g)
    flow = (torch.randn(2, 2, 17, 22, generator=g) * 1.5).requires_grad_(True)
    out.update(f4_img=img, f4_flow=flow.detach())
    for fun, k in (('gauss', 150.0), ('exp', 20.0)):
        for order in (1, 2):
            flow.grad = None
            loss = ref_loss.BilateralSmooth(0.1, fun, k, order)(img, flow)
            loss.backward()
            out.update({f'f4_{fun}_{order}_loss': loss.detach(), f'f4_{fun}_{order}_gflow': flow.grad.clone()})
    # F6 Resample2d.forward (resample2d.py:52-72, pure torch: runs on CPU) + the trainer's photometric metric
    # (trainer.py:61-62) with gradients w.r.t. image and flow, and occlusion_brox (occlusions.py:111-118) built on it.
    # Flows include sub-pixel, multi-pixel and out-of-image displacements (zeros padding + the (W-1,H-1) / align_corners
    # =False half-pixel quirk, SURVEY C-18).
    resample = ref_occ.Resample2d()
    img = torch.rand(2, 3, 13, 18, generator=g).requires_grad_(True)
    tgt = torch.rand(2, 3, 13, 18, generator=g)
    flow = (torch.randn(2, 2, 13, 18, generator=g) * 2.0)
    flow[1, :, :3] *= 5                                      # some taps far outside the image
    flow = flow.requires_grad_(True)
    warped = resample(img, flow)
    metric = torch.nn.functional.l1_loss(tgt, warped, reduction='none').mean(1, True)
    gw = torch.randn(2, 3, 13, 18, generator=g)
    gm = torch.randn(2, 1, 13, 18, generator=g)
    ((warped * gw).sum() + (metric * gm).sum()).backward()
    out.update(f6_img=img.detach(), f6_tgt=tgt, f6_flow=flow.detach(), f6_warped=warped.detach(),
               f6_metric=metric.detach(), f6_gw=gw, f6_gm=gm, f6_gimg=img.grad.clone(), f6_gflow=flow.grad.clone())
    zero = resample(img.detach(), torch.zeros(2, 2, 13, 18))  # quirk C-18: zero flow is NOT the identity
    out.update(f6_zero_flow_warped=zero)
    fw = torch.randn(2, 2, 13, 18, generator=g) * 1.5
    bw = -fw + 0.6 * torch.randn(2, 2, 13, 18, generator=g)
    out.update(f6_fw=fw, f6_bw=bw, f6_brox=ref_occ.occlusion_brox(fw, bw, 0.5).to(torch.uint8))
    # F6b: the same operator on operands that are exactly representable in bf16 (the arithmetic BASELINE configs[3] names for
    # the warp: bf16 images in HBM, fp32 interpolation).  The reference evaluates in fp32 on those values; the HIP bf16 kernels
    # must reproduce it up to the rounding of their bf16 OUTPUT (appended after every other draw: earlier arrays are unchanged).
    def bf(t):
        return t.to(torch.bfloat16).float()
    img_b = bf(torch.rand(2, 3, 21, 26, generator=g)).requires_grad_(True)
    tgt_b = bf(torch.rand(2, 3, 21, 26, generator=g))
    flow_b = torch.randn(2, 2, 21, 26, generator=g) * 2.0
    flow_b[0, :, -4:] *= 6
    flow_b = flow_b.requires_grad_(True)
    warped_b = resample(img_b, flow_b)
    warped_q = warped_b + (bf(warped_b) - warped_b).detach()           # what a bf16 consumer reads back (straight-through)
    metric_b = torch.nn.functional.l1_loss(tgt_b, warped_q, reduction='none').mean(1, True)
    gw_b = bf(torch.randn(2, 3, 21, 26, generator=g))
    gm_b = torch.randn(2, 1, 21, 26, generator=g)
    ((warped_b * gw_b).sum() + (metric_b * gm_b).sum()).backward()
    out.update(f6b_img=img_b.detach(), f6b_tgt=tgt_b, f6b_flow=flow_b.detach(), f6b_warped=warped_b.detach(),
               f6b_metric=metric_b.detach(), f6b_gw=gw_b, f6b_gm=gm_b, f6b_gimg=img_b.grad.clone(), f6b_gflow=flow_b.grad.clone())
    np.savez_compressed(os.path.join(HERE, 'golden_flow.npz'),
                        **{k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in out.items()})
    print('wrote golden_flow.npz with', len(out), 'arrays')


if __name__ == '__main__':
    main()
