"""TEST INFRASTRUCTURE ONLY -- CPU (torch) restatement of the flow trainer's photometric-loss utilities.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.

Pinning:
  * occlusion_wang / get_corresponding_map / occlusion_brox / CensusLoss / L1Loss / SSIMLoss / BilateralSmooth are checked
    against fixtures produced by the reference's OWN code (tests/golden/make_golden_flow.py imports
    video-interpolation/my_utils/{occlusions,loss}.py; golden_flow.npz).
  * softsplat: the reference implementation is three CUDA kernels compiled through cupy (absent here), so it cannot run in
    the build container: **parity unpinned** for softsplat -- restated from the kernel text
    (video-interpolation/my_utils/softsplat.py:8-177) and anchored by properties: the gradients of this restatement come
    from torch.autograd, and <splat(x, f), g> == <x, gather(g, f)> ties the forward scatter to the backward gather.
"""
import torch


def softsplat_sum(inp, flow):
    """softsplat.py:8-52 (kernel_Softsplat_updateOutput): out[b,c,Y,X] += in[b,c,y,x] * w over the four neighbours of
    (x + fx, y + fy); taps outside the image are dropped."""
    b, c, h, w = inp.shape
    ys, xs = torch.meshgrid(torch.arange(h, dtype=inp.dtype), torch.arange(w, dtype=inp.dtype), indexing='ij')
    ox = xs[None] + flow[:, 0]
    oy = ys[None] + flow[:, 1]
    nwx, nwy = torch.floor(ox), torch.floor(oy)
    out = torch.zeros(b, c, h * w, dtype=inp.dtype)
    src = inp.reshape(b, c, h * w)
    for dx, dy in ((0, 0), (1, 0), (0, 1), (1, 1)):
        tx, ty = nwx + dx, nwy + dy
        # weight of a corner = area of the opposite sub-rectangle (softsplat.py:29-32)
        wx = (nwx + 1 - ox) if dx == 0 else (ox - nwx)
        wy = (nwy + 1 - oy) if dy == 0 else (oy - nwy)
        wgt = (wx * wy).reshape(b, 1, h * w)
        valid = ((tx >= 0) & (tx < w) & (ty >= 0) & (ty < h)).reshape(b, 1, h * w)
        idx = (ty.clamp(0, h - 1) * w + tx.clamp(0, w - 1)).long().reshape(b, 1, h * w).expand(b, c, h * w)
        out = out.scatter_add(2, idx, src * wgt * valid)
    return out.reshape(b, c, h, w)


def function_softsplat(inp, flow, metric, mode):
    """softsplat.py:331-358 (FunctionSoftsplat)."""
    assert mode in ('summation', 'average', 'linear', 'softmax')
    if mode == 'average':
        inp = torch.cat([inp, inp.new_ones(inp.shape[0], 1, inp.shape[2], inp.shape[3])], 1)
    elif mode == 'linear':
        inp = torch.cat([inp * metric, metric], 1)
    elif mode == 'softmax':
        inp = torch.cat([inp * metric.exp(), metric.exp()], 1)
    out = softsplat_sum(inp, flow)
    if mode != 'summation':
        norm = out[:, -1:]
        norm = torch.where(norm == 0.0, torch.ones_like(norm), norm)
        out = out[:, :-1] / norm
    return out


def get_corresponding_map(data):
    """occlusions.py:29-77: `data` = unnormalised target coordinates (B,2,H,W); scatter of the four clamped corner weights,
    corners that needed clamping contribute 0."""
    b, _, h, w = data.shape
    x = data[:, 0].reshape(b, -1)
    y = data[:, 1].reshape(b, -1)
    x1, y1 = torch.floor(x), torch.floor(y)
    xf, yf = x1.clamp(0, w - 1), y1.clamp(0, h - 1)
    x0, y0 = x1 + 1, y1 + 1
    xc, yc = x0.clamp(0, w - 1), y0.clamp(0, h - 1)
    xco, yco, xfo, yfo = x0 != xc, y0 != yc, x1 != xf, y1 != yf
    out = torch.zeros(b, h * w, dtype=data.dtype)
    for cx, cy, bad in ((xc, yc, xco | yco), (xc, yf, xco | yfo), (xf, yc, xfo | yco), (xf, yf, xfo | yfo)):
        val = (1 - (x - cx).abs()) * (1 - (y - cy).abs())
        val = torch.where(bad, torch.zeros_like(val), val)
        out = out.scatter_add(1, (cx + cy * w).long(), val)
    return out.reshape(b, 1, h, w)


def occlusion_wang(flow12, flow21, thresh):
    """occlusions.py:96-103 (only flow21 is used)."""
    b, _, h, w = flow21.shape
    ys, xs = torch.meshgrid(torch.arange(h, dtype=flow21.dtype), torch.arange(w, dtype=flow21.dtype), indexing='ij')
    grid = torch.stack([xs, ys], 0)[None].expand(b, 2, h, w)
    corr = get_corresponding_map(grid + flow21)
    return torch.logical_not(corr <= thresh).float()


def census_loss(im, im_warp, mask, weight, max_distance=2):
    """loss.py:30-72 (CensusLoss.forward): ternary census transform of both masked grey images, soft Hamming distance,
    inner-region mask, mean, rescaled by numel(mask) / sum(mask)."""
    p = 2 * max_distance + 1

    def ternary(image):
        grey = (image[:, 0] * 0.2989 + image[:, 1] * 0.5870 + image[:, 2] * 0.1140).unsqueeze(1) * 255
        wts = torch.eye(p * p, dtype=image.dtype).view(p * p, 1, p, p)
        patches = torch.conv2d(grey, wts, padding=max_distance)
        t = patches - grey
        return t / torch.sqrt(0.81 + t ** 2)

    t1, t2 = ternary(im * mask), ternary(im_warp * mask)
    dist = (t1 - t2) ** 2
    dist = (dist / (0.1 + dist)).mean(1, keepdim=True)
    n, _, h, w = im.shape
    valid = torch.zeros(n, 1, h, w, dtype=im.dtype)
    valid[:, :, max_distance:h - max_distance, max_distance:w - max_distance] = 1
    return (dist * valid).mean() / mask.sum() * mask.numel() * weight


def l1_loss(im1, im2, mask, weight):
    """loss.py:17-27 (L1Loss.forward)."""
    return torch.nn.functional.l1_loss(im1 * mask, im2 * mask) / mask.sum() * mask.numel() * weight


def bilateral_smooth(img, flow, weight, abs_fun, edge_constant, order):
    """loss.py:106-132 (BilateralSmooth.forward) with my_utils/utils.py:6-13 (image_grads, robust_l1)."""
    def grads(t, stride=1):
        return t[:, :, stride:] - t[:, :, :-stride], t[:, :, :, stride:] - t[:, :, :, :-stride]

    def robust(x):
        return (x ** 2 + 0.001 ** 2) ** 0.5

    f = torch.abs if abs_fun == 'exp' else (lambda x: x ** 2)
    igx, igy = grads(img, order)
    fgx, fgy = grads(flow)
    wx = torch.exp(-f(edge_constant * igx).mean(1)).unsqueeze(1)
    wy = torch.exp(-f(edge_constant * igy).mean(1)).unsqueeze(1)
    if order == 1:
        loss = ((wx * robust(fgx)).mean() + (wy * robust(fgy)).mean()) / 2
    else:
        fgxx, _ = grads(fgx)
        _, fgyy = grads(fgy)
        loss = ((wx * robust(fgxx)).mean() + (wy * robust(fgyy)).mean()) / 2
    return loss * weight


def ssim_loss(x, y, mask, weight, md=1):
    """loss.py:75-103 (SSIMLoss.forward)."""
    x, y = x * mask, y * mask
    pool = torch.nn.AvgPool2d(2 * md + 1, 1, 0)
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    mu_x, mu_y = pool(x), pool(y)
    sx = pool(x * x) - mu_x.pow(2)
    sy = pool(y * y) - mu_y.pow(2)
    sxy = pool(x * y) - mu_x * mu_y
    ssim = (2 * mu_x * mu_y + c1) * (2 * sxy + c2) / ((mu_x.pow(2) + mu_y.pow(2) + c1) * (sx + sy + c2))
    dist = torch.clamp((1 - ssim) / 2, 0, 1)
    return dist.mean() / mask.sum() * mask.numel() * weight


def occlusion_brox(orig_fw, orig_bw, thresh=None):
    """occlusions.py:111-118; the warp is Resample2d restated in oracle/sininn_oracle.py::flow_warp.  Both are pinned by
    fixture F6 (the live Resample2d class is pure torch and runs on CPU; only the dead Resample2d_old needs the CUDA
    extension)."""
    from oracle import sininn_oracle as O
    warped_bw = O.flow_warp(orig_bw, orig_fw)
    sq_sum = ((orig_fw + warped_bw) ** 2).sum(1)
    sum_sq = (orig_fw ** 2 + warped_bw ** 2).sum(1)
    return (sq_sum >= 0.01 * sum_sq + 0.5).unsqueeze(1)
