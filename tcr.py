"""Transformation-consistency warp (drop-in for the reference's tcr.py, which needs kornia 0.4.1).

``TCR(angle, trans)(img, random, scale=1)`` rotates each sample about the image centre by an angle in
[-angle, angle] degrees and shifts it by up to ``trans`` pixels (divided by ``scale``), both driven by the (B,3)
uniform numbers in ``random`` (reference tcr.py:26-45).  The 2x3 matrix is built on the host exactly where the
reference builds it (it is B*6 numbers); the warp itself -- what kornia.warp_affine evaluates, SURVEY.md
Appendix B -- is one fused HIP kernel (sin-inn_amd functional.affine_warp) with an image gradient.
"""
import math

import torch
import torch.nn as nn

from sin_inn_amd.functional import affine_warp


def pixel_matrix(random, h, w, angle, trans, scale=1):
    """(B,2,3) pixel-space affine: rotation about (w/2, h/2) by (2r0-1)*angle deg, then a shift of (2r-1)*trans/scale."""
    r = random.detach().float().cpu()
    theta = torch.deg2rad((2.0 * r[:, 0] - 1.0) * angle) if angle != 0 else torch.zeros(r.shape[0])
    ca, sa = torch.cos(theta), torch.sin(theta)
    cx, cy = w / 2.0, h / 2.0
    mat = torch.zeros(r.shape[0], 2, 3)
    mat[:, 0, 0], mat[:, 0, 1], mat[:, 0, 2] = ca, sa, (1.0 - ca) * cx - sa * cy
    mat[:, 1, 0], mat[:, 1, 1], mat[:, 1, 2] = -sa, ca, sa * cx + (1.0 - ca) * cy
    mat[:, 0, 2] += (2.0 * r[:, 1] - 1.0) * trans / scale
    mat[:, 1, 2] += (2.0 * r[:, 2] - 1.0) * trans / scale
    return mat


def normalized_inverse(mat, h, w):
    """kornia.warp_affine's matrix handling: normalise by (W-1, H-1), invert, keep the top 2 rows."""
    b = mat.shape[0]
    full = torch.zeros(b, 3, 3)
    full[:, :2] = mat
    full[:, 2, 2] = 1.0
    norm = torch.tensor([[2.0 / max(w - 1, 1e-14), 0.0, -1.0], [0.0, 2.0 / max(h - 1, 1e-14), -1.0], [0.0, 0.0, 1.0]])
    return torch.inverse(norm @ full @ torch.inverse(norm))[:, :2, :].contiguous()


class TCR(nn.Module):
    def __init__(self, angle, trans):
        super().__init__()
        self.ang = angle
        self.trans = trans

    def forward(self, img, random, scale=1):
        _, _, h, w = img.shape
        theta = normalized_inverse(pixel_matrix(random, h, w, self.ang, self.trans, scale), h, w)
        if img.is_cuda:
            # B x 6 numbers: pinned + non_blocking, so the upload is stream-ordered like everything else in the step (a pageable
            # copy would make the host wait for the stream, i.e. for the passes queued before the TCR branch)
            theta = theta.pin_memory().to(img.device, non_blocking=True)
        return affine_warp(img, theta)
