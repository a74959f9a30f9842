"""Host-side mirror of the flow trainer's photometric-loss utilities, same names, argument meaning and
error behaviour as the reference; all arithmetic runs in libsininn.so (csrc/flowloss.hip):

    FunctionSoftsplat / ModuleSoftsplat   video-interpolation/my_utils/softsplat.py:331-371
    occlusion_wang / occlusion_unity      video-interpolation/my_utils/occlusions.py:96-109
    CensusLoss                            video-interpolation/my_utils/loss.py:30-72

Tensors are NCHW fp32 CUDA tensors (made contiguous like the reference does); CPU tensors raise NotImplementedError
(softsplat.py:289-290).  The reference's `occlusion_brox` and the Resample2d warp are served by
`sin_inn_amd.functional.flow_warp_l1`.
"""
import torch

from . import _lib
from .ops import _stream, ptr

check = _lib.check


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


def occlusion_unity(flow, *args):
    """occlusions.py:106-108: placeholder all-True mask."""
    return torch.ones_like(flow[:, 0], dtype=torch.bool).unsqueeze(1)


class _CensusFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, im, im_warp, mask, weight, max_distance):
        im, im_warp, mask = _prep(im), _prep(im_warp), _prep(mask.to(torch.float32))
        b, c, h, w = im.shape
        assert c == 3 and im_warp.shape == im.shape, 'CensusLoss expects two (B,3,H,W) images'
        assert mask.shape == (b, 1, h, w), 'CensusLoss expects a (B,1,H,W) mask'
        acc = im.new_zeros(130)             # SININN_CENSUS_ACC_FLOATS: {sum d, sum mask} + 64 partial slots
        out = im.new_empty(1)
        check(_lib.lib().sininn_census(ptr(im), ptr(im_warp), ptr(mask), b, h, w, int(max_distance), float(weight),
                                       ptr(acc), ptr(out), _stream()))
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
            check(_lib.lib().sininn_census_bwd(ptr(im), ptr(im_warp), ptr(mask), b, h, w, ctx.md, ctx.weight, ptr(acc),
                                               ptr(gs), ptr(g1), ptr(g2), _stream()))
        return g1, g2, None, None, None


class BaseLoss(torch.nn.Module):
    """loss.py:6-14."""

    def __init__(self, weight=0):
        super().__init__()
        self.weight = weight

    def forward(self, *args):
        return 0


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
        if not torch.is_tensor(mask):
            mask = torch.as_tensor(float(mask), device=im.device)
        if mask.numel() == 1:      # the trainer's `torch.ones(2)` placeholder: same value as an all-`mask` (B,1,H,W) map
            mask = mask.to(im.device, torch.float32).reshape(1, 1, 1, 1).expand(im.shape[0], 1, im.shape[2], im.shape[3])
        return _CensusFn.apply(im, im_warp, mask, self.weight, self.max_distance)
