"""Host-side mirror of the flow trainer's photometric-loss utilities, same names, argument meaning and
error behaviour as the reference; all arithmetic runs in libsininn.so (csrc/flowloss.hip):

    FunctionSoftsplat / ModuleSoftsplat   video-interpolation/my_utils/softsplat.py:331-371
    occlusion_wang / _brox / _unity       video-interpolation/my_utils/occlusions.py:96-118
    L1Loss / CensusLoss / SSIMLoss / BilateralSmooth   video-interpolation/my_utils/loss.py:17-132

Tensors are NCHW fp32 CUDA tensors (made contiguous like the reference does); CPU tensors raise NotImplementedError
(softsplat.py:289-290).  The Resample2d warp is `sin_inn_amd.functional.flow_warp_l1`.
"""
import torch

from . import _lib
from .ops import _stream, ptr

check = _lib.check
ACC_FLOATS = 130        # SININN_CENSUS_ACC_FLOATS: two sums + 64 x 2 partial slots


def _prep(t):
    if not t.is_cuda:
        raise NotImplementedError('sin-inn_amd flow-loss operators run on the GPU only (got a CPU tensor)')
    assert t.dtype == torch.float32, 'fp32 tensors expected'
    return t.contiguous()


class _FunctionSoftsplat(torch.autograd.Function):
    """softsplat.py:239-329 (_FunctionSoftsplat): summation splatting and its two gradients."""

    @staticmethod
    def forward(ctx, input, flow):
        assert flow.shape[1] == 2
        assert input.shape[2] == flow.shape[2]
        assert input.shape[3] == flow.shape[3]
        assert input.shape[0] == flow.shape[0]
        input, flow = _prep(input), _prep(flow)
        b, c, h, w = input.shape
        output = input.new_zeros([b, c, h, w])
        check(_lib.lib().sininn_softsplat(ptr(input), ptr(flow), b, c, h, w, ptr(output), _stream()))
        ctx.save_for_backward(input, flow)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        input, flow = ctx.saved_tensors
        b, c, h, w = input.shape
        grad_output = _prep(grad_output)
        gin = torch.empty_like(input) if ctx.needs_input_grad[0] else None
        gflow = torch.empty_like(flow) if ctx.needs_input_grad[1] else None
        if gin is not None or gflow is not None:
            check(_lib.lib().sininn_softsplat_bwd(ptr(input), ptr(flow), ptr(grad_output), b, c, h, w, ptr(gin),
                                                  ptr(gflow), _stream()))
        return gin, gflow


def FunctionSoftsplat(tenInput, tenFlow, tenMetric, strType):
    """softsplat.py:331-358: 'summation' | 'average' | 'linear' | 'softmax' splatting of tenInput along tenFlow."""
    assert tenMetric is None or tenMetric.shape[1] == 1
    assert strType in ['summation', 'average', 'linear', 'softmax']
    if strType == 'average':
        tenInput = torch.cat([tenInput, tenInput.new_ones(tenInput.shape[0], 1, tenInput.shape[2], tenInput.shape[3])], 1)
    elif strType == 'linear':
        tenInput = torch.cat([tenInput * tenMetric, tenMetric], 1)
    elif strType == 'softmax':
        tenInput = torch.cat([tenInput * tenMetric.exp(), tenMetric.exp()], 1)
    tenOutput = _FunctionSoftsplat.apply(tenInput, tenFlow)
    if strType != 'summation':
        tenNormalize = tenOutput[:, -1:, :, :]
        tenNormalize = torch.where(tenNormalize == 0.0, torch.ones_like(tenNormalize), tenNormalize)
        tenOutput = tenOutput[:, :-1, :, :] / tenNormalize
    return tenOutput


class ModuleSoftsplat(torch.nn.Module):
    def __init__(self, strType):
        super().__init__()
        self.strType = strType

    def forward(self, tenInput, tenFlow, tenMetric):
        return FunctionSoftsplat(tenInput, tenFlow, tenMetric, self.strType)


def get_corresponding_map(data_minus_grid):
    """Range map of a flow field (occlusions.py:29-77), given the FLOW (the reference passes base_grid + flow; the pixel
    grid is added inside the kernel).  Returns (B,1,H,W)."""
    flow = _prep(data_minus_grid)
    b, two, h, w = flow.shape
    assert two == 2
    corr = flow.new_zeros(b, 1, h, w)
    check(_lib.lib().sininn_occlusion_wang(ptr(flow), b, h, w, 0.0, ptr(corr), None, _stream()))
    return corr


def occlusion_wang(flow12, flow21, thresh):
    """occlusions.py:96-103: 1 where the range map of flow21 exceeds `thresh`, else 0 (float mask, no gradient)."""
    flow21 = _prep(flow21.detach())
    b, two, h, w = flow21.shape
    assert two == 2
    corr = flow21.new_zeros(b, 1, h, w)
    mask = torch.empty_like(corr)
    check(_lib.lib().sininn_occlusion_wang(ptr(flow21), b, h, w, float(thresh), ptr(corr), ptr(mask), _stream()))
    return mask


def occlusion_brox(orig_fw, orig_bw, thresh):
    """occlusions.py:111-118: forward-backward consistency (True = inconsistent; `thresh` is unused, as in the reference).
    The warp is Resample2d (sin_inn_amd.functional.flow_warp_l1), the test one small kernel."""
    from .functional import flow_warp_l1
    fw, bw = _prep(orig_fw.detach()), _prep(orig_bw.detach())
    b, two, h, w = fw.shape
    assert two == 2 and bw.shape == fw.shape
    warped_bw, _ = flow_warp_l1(bw, fw)
    warped_bw = warped_bw.contiguous()
    mask = torch.empty((b, 1, h, w), device=fw.device, dtype=torch.uint8)
    check(_lib.lib().sininn_occlusion_brox(ptr(fw), ptr(warped_bw), b, h, w, mask.data_ptr(), _stream()))
    return mask.bool()


def occlusion_unity(flow, *args):
    """occlusions.py:106-108: placeholder all-True mask."""
    return torch.ones_like(flow[:, 0], dtype=torch.bool).unsqueeze(1)


class _CensusFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, im, im_warp, mask, weight, max_distance):
        im, im_warp, mask = _prep(im), _prep(im_warp), _prep(mask.to(torch.float32))
        b, c, h, w = im.shape
        assert c == 3 and im_warp.shape == im.shape, 'CensusLoss expects two (B,3,H,W) images'
        assert mask.shape in ((b, 1, h, w), (b, 3, h, w)), 'CensusLoss expects a (B,1,H,W) or (B,3,H,W) mask'
        acc = im.new_zeros(ACC_FLOATS)
        out = im.new_empty(1)
        check(_lib.lib().sininn_census(ptr(im), ptr(im_warp), ptr(mask), mask.shape[1], b, h, w, int(max_distance),
                                       float(weight), ptr(acc), ptr(out), _stream()))
        ctx.save_for_backward(im, im_warp, mask, acc)
        ctx.weight, ctx.md = float(weight), int(max_distance)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        im, im_warp, mask, acc = ctx.saved_tensors
        b, _, h, w = im.shape
        g1 = torch.empty_like(im) if ctx.needs_input_grad[0] else None
        g2 = torch.empty_like(im_warp) if ctx.needs_input_grad[1] else None
        if g1 is not None or g2 is not None:
            gs = g.reshape(1).to(torch.float32).contiguous()
            check(_lib.lib().sininn_census_bwd(ptr(im), ptr(im_warp), ptr(mask), mask.shape[1], b, h, w, ctx.md, ctx.weight,
                                               ptr(acc), ptr(gs), ptr(g1), ptr(g2), _stream()))
        return g1, g2, None, None, None


class BaseLoss(torch.nn.Module):
    """loss.py:6-14."""

    def __init__(self, weight=0):
        super().__init__()
        self.weight = weight

    def forward(self, *args):
        return 0


def _expand_mask(mask, ref):
    """The trainer passes (B,1,H,W) or (B,3,H,W) float masks, or the scalar placeholder `torch.ones(2)[i]`; a scalar v
    gives the same loss value as an all-v (B,1,H,W) map (numel / sum cancel the same way)."""
    if not torch.is_tensor(mask):
        mask = torch.as_tensor(float(mask), device=ref.device)
    if mask.numel() == 1:
        mask = mask.to(ref.device, torch.float32).reshape(1, 1, 1, 1).expand(ref.shape[0], 1, ref.shape[2], ref.shape[3])
    return mask.to(torch.float32)


class _MaskedL1Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, im1, im2, mask, weight):
        im1, im2, mask = _prep(im1), _prep(im2), _prep(mask)
        b, c, h, w = im1.shape
        assert im2.shape == im1.shape and mask.shape in ((b, 1, h, w), (b, c, h, w))
        acc = im1.new_zeros(ACC_FLOATS)
        out = im1.new_empty(1)
        check(_lib.lib().sininn_masked_l1(ptr(im1), ptr(im2), ptr(mask), mask.shape[1], b, c, h, w, float(weight), ptr(acc),
                                          ptr(out), _stream()))
        ctx.save_for_backward(im1, im2, mask, acc)
        ctx.weight = float(weight)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        im1, im2, mask, acc = ctx.saved_tensors
        b, c, h, w = im1.shape
        g1 = torch.empty_like(im1) if ctx.needs_input_grad[0] else None
        g2 = torch.empty_like(im2) if ctx.needs_input_grad[1] else None
        if g1 is not None or g2 is not None:
            gs = g.reshape(1).to(torch.float32).contiguous()
            check(_lib.lib().sininn_masked_l1_bwd(ptr(im1), ptr(im2), ptr(mask), mask.shape[1], b, c, h, w, ctx.weight,
                                                  ptr(acc), ptr(gs), ptr(g1), ptr(g2), _stream()))
        return g1, g2, None, None


class L1Loss(BaseLoss):
    """loss.py:17-27: occlusion-masked L1, rescaled by numel(mask) / sum(mask); one fused reduction kernel."""

    def __init__(self, weight):
        super().__init__(weight)

    def forward(self, im1, im2, mask):
        if self.weight == 0:
            return super().forward()
        return _MaskedL1Fn.apply(im1, im2, _expand_mask(mask, im1), self.weight)


class _SSIMFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, mask, weight, md):
        x, y, mask = _prep(x), _prep(y), _prep(mask)
        b, c, h, w = x.shape
        assert y.shape == x.shape and mask.shape in ((b, 1, h, w), (b, c, h, w))
        acc = x.new_zeros(ACC_FLOATS)
        out = x.new_empty(1)
        check(_lib.lib().sininn_ssim(ptr(x), ptr(y), ptr(mask), mask.shape[1], b, c, h, w, int(md), float(weight), ptr(acc),
                                     ptr(out), _stream()))
        ctx.save_for_backward(x, y, mask, acc)
        ctx.cfg = (float(weight), int(md))
        return out[0]

    @staticmethod
    def backward(ctx, g):
        x, y, mask, acc = ctx.saved_tensors
        b, c, h, w = x.shape
        weight, md = ctx.cfg
        g1 = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        g2 = torch.empty_like(y) if ctx.needs_input_grad[1] else None
        if g1 is not None or g2 is not None:
            gs = g.reshape(1).to(torch.float32).contiguous()
            check(_lib.lib().sininn_ssim_bwd(ptr(x), ptr(y), ptr(mask), mask.shape[1], b, c, h, w, md, weight, ptr(acc), ptr(gs),
                                             ptr(g1), ptr(g2), _stream()))
        return g1, g2, None, None, None


class SSIMLoss(BaseLoss):
    """loss.py:75-103: (1 - SSIM) / 2 over unpadded (2 md + 1)^2 windows of the masked images, clamped to [0, 1]."""

    def __init__(self, weight, md=1):
        super().__init__(weight=weight)
        self.md = md

    def forward(self, x, y, mask):
        if self.weight == 0:
            return super().forward()
        return _SSIMFn.apply(x, y, _expand_mask(mask, x), self.weight, self.md)


class _SmoothFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, flow, weight, gauss, edge_constant, order):
        img, flow = _prep(img), _prep(flow)
        b, c, h, w = img.shape
        assert flow.shape == (b, 2, h, w)
        acc = img.new_zeros(ACC_FLOATS)
        out = img.new_empty(1)
        check(_lib.lib().sininn_bilateral_smooth(ptr(img), ptr(flow), b, c, h, w, int(order), int(gauss),
                                                 float(edge_constant), float(weight), ptr(acc), ptr(out), _stream()))
        ctx.save_for_backward(img, flow)
        ctx.cfg = (float(weight), int(gauss), float(edge_constant), int(order))
        return out[0]

    @staticmethod
    def backward(ctx, g):
        img, flow = ctx.saved_tensors
        b, c, h, w = img.shape
        weight, gauss, k, order = ctx.cfg
        gflow = None
        if ctx.needs_input_grad[1]:
            gflow = torch.empty_like(flow)
            gs = g.reshape(1).to(torch.float32).contiguous()
            check(_lib.lib().sininn_bilateral_smooth_bwd(ptr(img), ptr(flow), b, c, h, w, order, gauss, k, weight, ptr(gs),
                                                         ptr(gflow), _stream()))
        return None, gflow, None, None, None, None


class BilateralSmooth(BaseLoss):
    """loss.py:106-132: edge-aware first / second order smoothness (the guiding image gets no gradient: it is a frame)."""

    def __init__(self, weight, abs_fun, edge_constant, order):
        super().__init__(weight)
        assert abs_fun in ('exp', 'gauss') and order in (1, 2)
        self.gauss = abs_fun == 'gauss'
        self.edge_constant = edge_constant
        self.order = order

    def forward(self, img, flow):
        if self.weight == 0:
            return super().forward()
        return _SmoothFn.apply(img, flow, self.weight, self.gauss, self.edge_constant, self.order)


class CensusLoss(BaseLoss):
    """loss.py:30-72.  One fused kernel: grey levels of both masked images staged in LDS, ternary census over the
    (2*max_distance+1)^2 patch, soft Hamming distance, inner-region mask and the reduction; the mask gets no gradient
    (the trainer builds it from comparisons)."""

    def __init__(self, weight, max_distance=2):
        super().__init__(weight=weight)
        self.max_distance = max_distance
        self.patch_size = 2 * max_distance + 1

    def forward(self, im, im_warp, mask):
        if self.weight == 0:
            return super().forward()
        return _CensusFn.apply(im, im_warp, _expand_mask(mask, im), self.weight, self.max_distance)
