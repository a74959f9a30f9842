"""autograd front ends of the HBM-bound path kernels: losses (loss.py), TCR affine warp (tcr.py + kornia),
flow warp + photometric metric (video-interpolation Resample2d), frame-window sampler (data.py)."""
import torch

from . import ops


def _as4d(t):
    if t.dim() == 4:
        return t
    return t.reshape(1, 1, 1, -1) if t.is_contiguous() else t.contiguous().reshape(1, 1, 1, -1)


def _empty_like_layout(t):
    """Dense tensor with t's shape; channels_last when t is stored pixel-major (also for channel slices)."""
    if t.dim() == 4 and t.stride(1) == 1 and t.shape[1] > 1:
        b, c, h, w = t.shape
        return torch.empty((b, h, w, c), device=t.device, dtype=t.dtype).permute(0, 3, 1, 2)
    return torch.empty(t.shape, device=t.device, dtype=t.dtype)


class _SqDiffMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y):
        x4 = _as4d(x.detach())
        y4 = _as4d(y.detach()) if y is not None else None
        out = torch.zeros(1, device=x.device, dtype=torch.float32)
        ops.sqdiff_sum(x4, y4, out)
        ctx.save_for_backward(x4, y4)
        ctx.shapes = (x.shape, None if y is None else y.shape)
        return (out / x.numel()).reshape(())

    @staticmethod
    def backward(ctx, g):
        x4, y4 = ctx.saved_tensors
        gx = _empty_like_layout(x4) if ctx.needs_input_grad[0] else None
        gy = _empty_like_layout(y4) if (y4 is not None and ctx.needs_input_grad[1]) else None
        if gx is None and gy is None:
            return None, None
        ops.sqdiff_bwd(x4, y4, g.contiguous().reshape(1), 2.0 / x4.numel(), gx, gy)
        xs, ys = ctx.shapes
        return (gx.reshape(xs) if gx is not None else None), (gy.reshape(ys) if gy is not None else None)


def reconstruction(x, y):
    """loss.py:3-5: mean((x-y)^2)."""
    return _SqDiffMean.apply(x, y)


def latent_nll(z):
    """loss.py:38-39: mean(z^2)."""
    return _SqDiffMean.apply(z, None)


class _MMD(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, rev):
        x4, y4 = x.detach(), y.detach()
        assert x4.dim() == 4 and x4.shape == y4.shape
        b = x4.shape[0]
        g = torch.zeros(17 * 3 * b * b, device=x.device, dtype=torch.float32)    # result + SININN_MMD_SLOTS partial copies
        ops.mmd_gram(x4, y4, g)
        out = torch.empty(1, device=x.device, dtype=torch.float32)
        coef = torch.empty(4 * b * b, device=x.device, dtype=torch.float32)
        ops.mmd_finish(g, b, rev, out, coef)
        ctx.save_for_backward(x4, y4, coef)
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        x4, y4, coef = ctx.saved_tensors
        gx = _empty_like_layout(x4) if ctx.needs_input_grad[0] else None
        gy = _empty_like_layout(y4) if ctx.needs_input_grad[1] else None
        if gx is None and gy is None:
            return None, None, None
        ops.mmd_bwd(x4, y4, coef, g.contiguous().reshape(1), gx, gy)
        return gx, gy, None


def mmd(x, y, rev=False):
    """loss.py:9-36 (inverse multiquadric kernel MMD over the batch)."""
    return _MMD.apply(x, y, bool(rev))


class _AffineWarp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, theta):
        img4 = img.detach()
        theta = theta.detach().to(device=img.device, dtype=torch.float32).contiguous()
        out = _empty_like_layout(img4)
        ops.affine_warp(img4, theta, out)
        ctx.save_for_backward(theta)
        ctx.pm = img4.stride(1) == 1
        return out

    @staticmethod
    def backward(ctx, g):
        (theta,) = ctx.saved_tensors
        if not ctx.needs_input_grad[0]:
            return None, None
        b, c, h, w = g.shape
        if ctx.pm:
            gimg = torch.zeros((b, h, w, c), device=g.device, dtype=torch.float32).permute(0, 3, 1, 2)
        else:
            gimg = torch.zeros((b, c, h, w), device=g.device, dtype=torch.float32)
        ops.affine_warp_bwd(g, theta, gimg)
        return gimg, None


def affine_warp(img, theta):
    """affine_grid(theta, align_corners=False) + grid_sample(bilinear, zeros, align_corners=False), fused."""
    return _AffineWarp.apply(img, theta)


class _FlowWarpL1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, flow, target):
        img, flow = img.detach().contiguous(), flow.detach().contiguous()
        target = target.detach().contiguous() if target is not None else None
        warped = torch.empty_like(img)
        b, c, h, w = img.shape
        metric = torch.empty((b, 1, h, w), device=img.device, dtype=torch.float32) if target is not None else None
        ops.flow_warp_l1(img, flow, target, warped, metric)
        ctx.save_for_backward(img, flow, target, warped)
        if metric is None:
            metric = torch.zeros((), device=img.device)
            ctx.mark_non_differentiable(metric)
        return warped, metric

    @staticmethod
    def backward(ctx, gw, gm):
        img, flow, target, warped = ctx.saved_tensors
        gw = gw.contiguous() if gw is not None else None
        gm = gm.contiguous() if (gm is not None and target is not None) else None
        if gw is None and gm is None:
            return None, None, None
        # gradient accumulators are fp32 in both arithmetics (the image gradient is a sum of atomics); a bf16 image gets its
        # gradient back in its own dtype, as autograd requires
        gimg = torch.zeros(img.shape, device=img.device, dtype=torch.float32) if ctx.needs_input_grad[0] else None
        gflow = torch.empty_like(flow) if ctx.needs_input_grad[1] else None
        if gimg is None and gflow is None:
            return None, None, None
        if gw is not None and gw.dtype != img.dtype:
            gw = gw.to(img.dtype)
        ops.flow_warp_l1_bwd(img, flow, target, warped, gw, gm, gimg, gflow)
        return (gimg.to(img.dtype) if gimg is not None else None), gflow, None


def flow_warp_l1(img, flow, target=None):
    """Resample2d.forward (+ per-pixel channel-mean L1 against `target`): returns (warped, metric).  img / target fp32, or both
    bf16 (the arithmetic of BASELINE configs[3]: bf16 images in HBM, fp32 flow / interpolation / metric / gradients)."""
    _FlowWarpL1.set_materialize_grads = False
    return _FlowWarpL1.apply(img, flow, target)


def sample_pairs(hr_clip, idx, gap=1, dtype=torch.float32):
    """Frame pairs of the flow path (video-interpolation/trainer.py:49-62) from the resident uint8 clip (T,H,W,3): returns
    (clip[idx] / 255, clip[idx + gap] / 255) as planar (n,3,H,W) tensors, fp32 or bf16."""
    from . import _lib
    assert hr_clip.dtype == torch.uint8 and hr_clip.dim() == 4 and hr_clip.shape[-1] == 3 and hr_clip.is_contiguous()
    assert dtype in (torch.float32, torch.bfloat16) and idx.dtype == torch.int32
    if not hr_clip.is_cuda:
        raise NotImplementedError('sin-inn_amd ops run on the GPU only (got a CPU tensor)')
    t, h, w, _ = hr_clip.shape
    n = idx.numel()
    a = torch.empty((n, 3, h, w), device=hr_clip.device, dtype=dtype)
    b = torch.empty_like(a)
    _lib.check(_lib.lib().sininn_sample_pairs(hr_clip.data_ptr(), idx.data_ptr(), n, t, h, w, int(gap), a.data_ptr(), b.data_ptr(),
                                              int(dtype == torch.bfloat16), ops._stream()))
    return a, b


def sample_windows(hr_clip, lr_clip, idx, lr_window):
    """data.py:31-45 on an HBM-resident uint8 clip.  Returns hr (n,3,H,W) and lr (n,(2w+1)*4,h,w), both stored
    pixel-major (channels_last views)."""
    n = idx.numel()
    _, hh, ww, _ = hr_clip.shape
    _, h, w, _ = lr_clip.shape
    lrc = (2 * lr_window + 1) * 4
    hr = torch.empty((n, hh, ww, 3), device=hr_clip.device, dtype=torch.float32).permute(0, 3, 1, 2)
    lr = torch.empty((n, h, w, lrc), device=hr_clip.device, dtype=torch.float32).permute(0, 3, 1, 2)
    ops.sample_windows(hr_clip, lr_clip, idx.to(torch.int32), lr_window, hr, lr)
    return hr, lr


def bayer_bin(hr_clip, scale=4, reduction='mean'):
    """datasets/prepare.py LR synthesis on the GPU: hr (T,H,W,3) u8 -> lr (T, H/(2*scale), W/(2*scale), 4) u8 (RGGB planes)."""
    from . import _lib
    assert hr_clip.dtype == torch.uint8 and hr_clip.dim() == 4 and hr_clip.shape[-1] == 3 and hr_clip.is_contiguous()
    if not hr_clip.is_cuda:
        raise NotImplementedError('sin-inn_amd ops run on the GPU only (got a CPU tensor)')
    t, h, w, _ = hr_clip.shape
    lr = torch.empty((t, h // (2 * scale), w // (2 * scale), 4), device=hr_clip.device, dtype=torch.uint8)
    _lib.check(_lib.lib().sininn_bayer_bin(hr_clip.data_ptr(), lr.data_ptr(), t, h, w, scale,
                                           1 if reduction == 'sum' else 0, ops._stream()))
    return lr


def bayer_demosaic(hr_clip, scale=4, reduction='mean'):
    """datasets/prepare.py's demosaiced LR preview (pack_demosaic of the unquantised binned planes, :103-119,158,163-165):
    hr (T,H,W,3) u8 -> rgb (T, H/scale, W/scale, 3) u8."""
    from . import _lib
    assert hr_clip.dtype == torch.uint8 and hr_clip.dim() == 4 and hr_clip.shape[-1] == 3 and hr_clip.is_contiguous()
    if not hr_clip.is_cuda:
        raise NotImplementedError('sin-inn_amd ops run on the GPU only (got a CPU tensor)')
    t, h, w, _ = hr_clip.shape
    rgb = torch.empty((t, h // scale, w // scale, 3), device=hr_clip.device, dtype=torch.uint8)
    _lib.check(_lib.lib().sininn_bayer_demosaic(hr_clip.data_ptr(), rgb.data_ptr(), t, h, w, scale,
                                                1 if reduction == 'sum' else 0, ops._stream()))
    return rgb


def frames_to_u8(frames, wrap=False):
    """(B,C,H,W)-shaped float frames (any strides) -> (B,H,W,C) uint8 on the device.  wrap=False clamps to [0,1] before the
    *255; wrap=True reproduces torchvision's ToPILImage (mul(255).byte(), wrap-around) that the reference's infer uses."""
    from . import _lib
    assert frames.dim() == 4 and frames.dtype == torch.float32
    if not frames.is_cuda:
        raise NotImplementedError('sin-inn_amd ops run on the GPU only (got a CPU tensor)')
    b, c, h, w = frames.shape
    out = torch.empty((b, h, w, c), device=frames.device, dtype=torch.uint8)
    _lib.check(_lib.lib().sininn_frames_to_u8(ops.ptr(frames), ops.strides4(frames), out.data_ptr(), b, c, h, w,
                                              1 if wrap else 0, ops._stream()))
    return out
