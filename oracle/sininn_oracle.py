"""CPU restatement (torch fp32, CPU only) of the sin-inn single-video INN training path.

TEST INFRASTRUCTURE ONLY -- never imported by the product path (``sin-inn_amd/``,
``archs.py``, ``lit_wrapper.py`` ...).  It is the checker the HIP kernels are compared
against and the ``cpu_baseline`` ("port") that ``bench.py`` times on the host cores.

Parity status (see DESIGN.md "Oracle"):
  * pinned against the reference's own importable code through the committed fixtures in
    ``tests/golden/`` (made by ``tests/golden/make_golden.py``): conv subnets
    (archs.py:11-17), HaarDownsampling / DenseBlock / InvBlockExp / InvRescaleNet
    (archs.py:74-233), loss.reconstruction / loss.latent_nll (loss.py:3-5,38-39), loss.mmd
    (loss.py:9-36, fixture G7: the reference function itself run on CPU with its three
    `.to('cuda')` calls made no-ops by the fixture script), the legacy numpy permutation stream
    used by FrEIA's PermuteRandom, and -- through tests/golden/make_golden_flow.py, fixture F6 --
    flow_warp / photometric_l1 (= Resample2d.forward, video-interpolation/my_utils/
    resample2d.py:52-72, + the trainer's metric, trainer.py:61-62), values and gradients.
  * PARITY UNPINNED (third-party arithmetic that is not under /root/reference and is not
    installed here): FrEIA (un-pinned version, pre-0.2 API; call sites archs.py:26-71) and
    kornia==0.4.1 (requirements.txt:1; call sites tcr.py:35,43).  Their published algorithms are
    restated below and checked by known-answer tests only
    (round trip, log-det vs autograd Jacobian, identity warps).

All reference citations are file:line into /root/reference.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

GLOW_ATAN_GAIN = 0.636  # literal constant of FrEIA's GLOWCouplingBlock.e / log_e (not 2/pi)


# ----------------------------------------------------------------------------------------------
# index maps: IRevNetDownsampling / PermuteRandom  (FrEIA; archs.py:28-38,65-68)
# ----------------------------------------------------------------------------------------------
def squeeze_fwd(x):
    """Space-to-depth 2x2: out[b,(hb*2+wb)*C+c,i,j] = in[b,c,2i+hb,2j+wb]."""
    b, c, h, w = x.shape
    v = x.reshape(b, c, h // 2, 2, w // 2, 2)          # b c i hb j wb
    v = v.permute(0, 3, 5, 1, 2, 4)                    # b hb wb c i j
    return v.reshape(b, 4 * c, h // 2, w // 2).contiguous()


def squeeze_inv(y):
    b, c4, h, w = y.shape
    c = c4 // 4
    v = y.reshape(b, 2, 2, c, h, w)                    # b hb wb c i j
    v = v.permute(0, 3, 4, 1, 5, 2)                    # b c i hb j wb
    return v.reshape(b, c, 2 * h, 2 * w).contiguous()


def permutation(channels, seed):
    """FrEIA PermuteRandom: np.random.seed(seed); np.random.permutation(C) (legacy MT19937 stream)."""
    rs = np.random.RandomState(seed)
    perm = rs.permutation(channels)
    inv = np.zeros_like(perm)
    inv[perm] = np.arange(channels)
    return perm.astype(np.int64), inv.astype(np.int64)


# ----------------------------------------------------------------------------------------------
# conv subnets (archs.py:11-17)
# ----------------------------------------------------------------------------------------------
HIDDEN = 256


def make_subnet(c_in, c_out, ksize):
    """archs.py:11-13 (ksize 3, padding 1) and archs.py:15-17 (ksize 1); torch default Conv2d init."""
    pad = ksize // 2
    return nn.Sequential(nn.Conv2d(c_in, HIDDEN, ksize, padding=pad), nn.ReLU(),
                         nn.Conv2d(HIDDEN, c_out, ksize, padding=pad))


def bf16_round(t):
    """Round to bf16 (nearest even) and back, with a straight-through gradient: the checker's model of a bf16 operand."""
    return t + (t.to(torch.bfloat16).to(t.dtype) - t).detach()


def run_subnet(seq, x, emulate_bf16=False, gate=None):
    """The conv subnet (archs.py:11-17).  emulate_bf16 models the mixed-precision HIP path: inputs, weights and the
    hidden tensor are rounded to bf16, every product is accumulated in fp32, bias / ReLU / output stay fp32.
    gate (checker feature, not in the reference): a (B,256,H,W) 0/1 mask that REPLACES the ReLU's own decision, h = conv1(x)
    * gate -- with the gates another evaluation took, the network is a smooth function and two evaluations of it can be
    compared without the discrete noise of units whose pre-activation lies within rounding distance of 0."""
    c1, c2 = seq[0], seq[2]
    if gate is not None:
        assert not emulate_bf16
        return c2(c1(x) * gate.to(x.dtype))
    if not emulate_bf16:
        return seq(x)
    h = F.relu(F.conv2d(bf16_round(x), bf16_round(c1.weight), c1.bias, padding=c1.padding))
    return F.conv2d(bf16_round(h), bf16_round(c2.weight), c2.bias, padding=c2.padding)


# ----------------------------------------------------------------------------------------------
# GLOW coupling block (FrEIA GLOWCouplingBlock; called at archs.py:61-64)
# ----------------------------------------------------------------------------------------------
def log_e(s, clamp):
    return clamp * GLOW_ATAN_GAIN * torch.atan(s / clamp)


class GlowBlock(nn.Module):
    def __init__(self, channels, ksize, clamp=1.2):
        super().__init__()
        self.l1 = channels // 2
        self.l2 = channels - channels // 2
        self.clamp = clamp
        # construction order s1 then s2 == FrEIA's (fixes which random numbers each conv gets)
        self.s1 = make_subnet(self.l1, 2 * self.l2, ksize)
        self.s2 = make_subnet(self.l2, 2 * self.l1, ksize)
        self.last_jac = None
        self.emulate_bf16 = False        # checker for the mixed-precision HIP path (not a reference feature)
        self.forced_gates = None         # checker: {'s1': mask, 's2': mask} replaces the ReLU decisions (see run_subnet)

    def forward(self, x, rev=False):
        x1, x2 = x[:, :self.l1], x[:, self.l1:]
        bf = self.emulate_bf16
        g1 = g2 = None
        if self.forced_gates is not None:
            fg = self.forced_gates
            if rev in fg:                # {False: {...}, True: {...}}: a training step evaluates the block in both directions
                fg = fg[rev]
            g1, g2 = fg['s1'], fg['s2']
        if not rev:
            r2 = run_subnet(self.s2, x2, bf, g2)
            s2, t2 = r2[:, :self.l1], r2[:, self.l1:]
            y1 = torch.exp(log_e(s2, self.clamp)) * x1 + t2
            r1 = run_subnet(self.s1, y1, bf, g1)
            s1, t1 = r1[:, :self.l2], r1[:, self.l2:]
            y2 = torch.exp(log_e(s1, self.clamp)) * x2 + t1
            self.last_jac = (log_e(s1, self.clamp).sum(dim=(1, 2, 3))
                             + log_e(s2, self.clamp).sum(dim=(1, 2, 3)))
        else:
            r1 = run_subnet(self.s1, x1, bf, g1)
            s1, t1 = r1[:, :self.l2], r1[:, self.l2:]
            y2 = (x2 - t1) / torch.exp(log_e(s1, self.clamp))
            r2 = run_subnet(self.s2, y2, bf, g2)
            s2, t2 = r2[:, :self.l1], r2[:, self.l1:]
            y1 = (x1 - t2) / torch.exp(log_e(s2, self.clamp))
            self.last_jac = -(log_e(s1, self.clamp).sum(dim=(1, 2, 3))
                              + log_e(s2, self.clamp).sum(dim=(1, 2, 3)))
        return torch.cat((y1, y2), 1)


class _Index(nn.Module):
    """Parameter-free graph op (squeeze / permute / input placeholder)."""

    def __init__(self, kind, channels=0, seed=0):
        super().__init__()
        self.kind = kind
        if kind == 'permute':
            perm, inv = permutation(channels, seed)
            self.perm = torch.from_numpy(perm)
            self.perm_inv = torch.from_numpy(inv)
        self.last_jac = 0.0

    def forward(self, x, rev=False):
        if self.kind == 'squeeze':
            return squeeze_inv(x) if rev else squeeze_fwd(x)
        if self.kind == 'permute':
            return x[:, self.perm_inv] if rev else x[:, self.perm]
        return x


class SRFlowOracle(nn.Module):
    """archs.py:19-71 (UncondSRFlow graph) run by a ReversibleGraphNet-like loop (archs.py:71).

    ``module_list`` index i == node i of the reference's node list (input node at 0), so the
    state-dict keys read ``module_list.<i>.s1.0.weight`` as SURVEY Appendix A describes.
    """

    def __init__(self, c, h, w, scale=4, num_coupling=4, clamp=1.2):
        super().__init__()
        mods = [_Index('input'), _Index('squeeze')]
        ch = c * 4
        for _ in range((scale - 1).bit_length()):
            mods.append(_Index('squeeze'))
            ch *= 4
            for kk in range(num_coupling):
                mods.append(GlowBlock(ch, 3 if kk % 2 == 0 else 1, clamp))
                mods.append(_Index('permute', ch, kk))
        mods.append(_Index('output'))
        self.module_list = nn.ModuleList(mods)

    def forward(self, x, rev=False):
        seq = reversed(self.module_list) if rev else self.module_list
        for m in seq:
            x = m(x, rev=rev)
        return x

    def log_jacobian(self):
        tot = 0.0
        for m in self.module_list:
            if isinstance(m, GlowBlock):
                tot = tot + m.last_jac
        return tot


# ----------------------------------------------------------------------------------------------
# IRN architecture (archs.py:74-233) -- the only coupling code that lives in the reference tree
# ----------------------------------------------------------------------------------------------
def haar_fwd(x):
    """archs.py:187-192: 2x2 Haar analysis, /4, bands regrouped so out[:, k*C+c] = band k of channel c."""
    a = x[:, :, 0::2, 0::2]
    b = x[:, :, 0::2, 1::2]
    c = x[:, :, 1::2, 0::2]
    d = x[:, :, 1::2, 1::2]
    ll = (a + b + c + d) / 4.0
    b1 = (a - b + c - d) / 4.0     # archs.py:169-170 (right column negated)
    b2 = (a + b - c - d) / 4.0     # archs.py:172-173 (bottom row negated)
    b3 = (a - b - c + d) / 4.0     # archs.py:175-176
    return torch.cat((ll, b1, b2, b3), 1)


def haar_inv(y):
    """archs.py:194-199: inverse regroup + conv_transpose2d with the +-1 filters (no /4)."""
    ch = y.shape[1] // 4
    ll, b1, b2, b3 = y[:, :ch], y[:, ch:2 * ch], y[:, 2 * ch:3 * ch], y[:, 3 * ch:]
    bsz, _, h, w = ll.shape
    out = y.new_zeros(bsz, ch, 2 * h, 2 * w)
    out[:, :, 0::2, 0::2] = ll + b1 + b2 + b3
    out[:, :, 0::2, 1::2] = ll - b1 + b2 - b3
    out[:, :, 1::2, 0::2] = ll + b1 - b2 - b3
    out[:, :, 1::2, 1::2] = ll - b1 - b2 + b3
    return out


def haar_last_jac(shape_chw, rev=False):
    """archs.py:184-185,194-195."""
    elements = shape_chw[0] * shape_chw[1] * shape_chw[2]
    return elements / 4 * math.log(16.0 if rev else 1 / 16.0)


class DenseBlockOracle(nn.Module):
    """archs.py:74-98: five densely connected 3x3 convs (gc=32), LeakyReLU(0.2)."""

    def __init__(self, cin, cout, gc=32):
        super().__init__()
        self.convs = nn.ModuleList([nn.Conv2d(cin + i * gc, gc if i < 4 else cout, 3, 1, 1)
                                    for i in range(5)])
        # archs.py:84-86,100-132: xavier_normal*0.1 on conv1-4, kaiming_normal*0 on conv5, zero biases
        # (same RNG draw order as the reference so a seeded net reproduces its weights)
        for i, conv in enumerate(self.convs):
            if i < 4:
                nn.init.xavier_normal_(conv.weight)
                conv.weight.data *= 0.1
                conv.bias.data.zero_()
        nn.init.kaiming_normal_(self.convs[4].weight, a=0, mode='fan_in')
        self.convs[4].weight.data *= 0
        self.convs[4].bias.data.zero_()

        self.forced_gates = None         # checker: four (B,32,H,W) 0/1 masks replace the LeakyReLU decisions (run_subnet);
        self.rev = False                 # or {False: [...], True: [...]} selected by the direction the owning block runs in

    def forward(self, x):
        feats = [x]
        fg = self.forced_gates
        if isinstance(fg, dict):
            fg = fg[self.rev]
        for i, conv in enumerate(self.convs):
            y = conv(torch.cat(feats, 1))
            if i < 4:
                if fg is not None:
                    g = fg[i].to(y.dtype)
                    y = y * (0.2 + 0.8 * g)
                else:
                    y = F.leaky_relu(y, 0.2)
                feats.append(y)
        return y


class InvBlockExpOracle(nn.Module):
    """archs.py:135-160."""

    def __init__(self, channels, split1, clamp=1.0):
        super().__init__()
        self.l1, self.l2, self.clamp = split1, channels - split1, clamp
        self.F = DenseBlockOracle(self.l2, self.l1)
        self.G = DenseBlockOracle(self.l1, self.l2)
        self.H = DenseBlockOracle(self.l1, self.l2)

    def forward(self, x, rev=False):
        x1, x2 = x[:, :self.l1], x[:, self.l1:]
        self.F.rev = self.G.rev = self.H.rev = bool(rev)
        if not rev:
            y1 = x1 + self.F(x2)
            s = self.clamp * (torch.sigmoid(self.H(y1)) * 2 - 1)
            y2 = x2 * torch.exp(s) + self.G(y1)
        else:
            s = self.clamp * (torch.sigmoid(self.H(x1)) * 2 - 1)
            y2 = (x2 - self.G(x1)) / torch.exp(s)
            y1 = x1 - self.F(y2)
        return torch.cat((y1, y2), 1)


# ----------------------------------------------------------------------------------------------
# losses (loss.py)
# ----------------------------------------------------------------------------------------------
def reconstruction(x, y):
    """loss.py:3-5."""
    return ((x - y) ** 2).mean()


def latent_nll(z):
    """loss.py:38-39."""
    return (z ** 2).mean()


MMD_KERNELS_FWD = ((0.2, 2.0), (1.5, 2.0), (3.0, 2.0))     # loss.py:13
MMD_KERNELS_REV = ((0.2, 0.1), (0.2, 0.5), (0.2, 2.0))     # loss.py:11


def mmd(x, y, rev=False):
    """loss.py:9-36, device-agnostic (the reference hard-codes 'cuda' for three zero buffers)."""
    bsz = x.shape[0]
    xf, yf = x.reshape(bsz, -1), y.reshape(bsz, -1)
    xx, yy, xy = xf @ xf.t(), yf @ yf.t(), xf @ yf.t()
    rx = xx.diag().unsqueeze(0).expand_as(xx)
    ry = yy.diag().unsqueeze(0).expand_as(yy)
    dxx = (rx.t() + rx - 2.0 * xx).clamp(min=0)
    dyy = (ry.t() + ry - 2.0 * yy).clamp(min=0)
    dxy = (rx.t() + ry - 2.0 * xy).clamp(min=0)
    tot = torch.zeros_like(xx)
    for cc, a in (MMD_KERNELS_REV if rev else MMD_KERNELS_FWD):
        for d, sign in ((dxx, 1.0), (dyy, 1.0), (dxy, -2.0)):
            tot = tot + sign * cc ** a * ((cc + d) / a) ** (-a)
    return tot.mean()


# ----------------------------------------------------------------------------------------------
# TCR affine warp (tcr.py:26-45 + kornia 0.4.1)  -- PARITY UNPINNED
# ----------------------------------------------------------------------------------------------
def tcr_matrix(rand, h, w, angle_deg, trans_px, scale=1.0):
    """tcr.py:31-41 -> the 2x3 pixel-space matrix handed to kornia.warp_affine. rand: (B,3) in [0,1)."""
    rand = rand.float()
    ang = (2.0 * angle_deg) * rand[:, 0] - angle_deg                      # tcr.py:34
    rad = ang * (math.pi / 180.0)
    cos, sin = torch.cos(rad), torch.sin(rad)
    cx, cy = w / 2.0, h / 2.0                                             # tcr.py:31
    m = torch.zeros(rand.shape[0], 2, 3)
    m[:, 0, 0], m[:, 0, 1], m[:, 1, 0], m[:, 1, 1] = cos, sin, -sin, cos  # kornia angle_to_rotation_matrix
    m[:, 0, 2] = (1.0 - cos) * cx - sin * cy
    m[:, 1, 2] = sin * cx + (1.0 - cos) * cy
    m[:, 0, 2] += ((2.0 * trans_px) * rand[:, 1] - trans_px) / scale      # tcr.py:38,40
    m[:, 1, 2] += ((2.0 * trans_px) * rand[:, 2] - trans_px) / scale      # tcr.py:39,41
    return m


def tcr_theta(m, h, w):
    """kornia.warp_affine: (W-1,H-1)-normalise, invert -> theta for affine_grid(align_corners=False)."""
    bsz = m.shape[0]
    m3 = torch.zeros(bsz, 3, 3)
    m3[:, :2] = m
    m3[:, 2, 2] = 1.0
    n = torch.tensor([[2.0 / max(w - 1, 1e-14), 0.0, -1.0],
                      [0.0, 2.0 / max(h - 1, 1e-14), -1.0],
                      [0.0, 0.0, 1.0]])
    dst_norm_src_norm = n @ m3 @ torch.inverse(n)
    return torch.inverse(dst_norm_src_norm)[:, :2, :].contiguous()


def affine_warp(img, theta):
    grid = F.affine_grid(theta, list(img.shape), align_corners=False)
    return F.grid_sample(img, grid, mode='bilinear', padding_mode='zeros', align_corners=False)


def tcr_warp(img, rand, angle_deg, trans_px, scale=1.0):
    _, _, h, w = img.shape
    return affine_warp(img, tcr_theta(tcr_matrix(rand, h, w, angle_deg, trans_px, scale), h, w))


# ----------------------------------------------------------------------------------------------
# optical-flow backward warp + photometric metric
# (video-interpolation/my_utils/resample2d.py:52-72, video-interpolation/trainer.py:61-62)
# ----------------------------------------------------------------------------------------------
def flow_warp(img, flow):
    """grid = (coords+flow)/(W-1,H-1)*2-1 ; grid_sample bilinear, zeros, align_corners=False (quirk C-18)."""
    _, _, h, w = flow.shape
    ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing='ij')
    coords = torch.stack((xs, ys), 0).float()[None]
    new = (coords + flow).permute(0, 2, 3, 1)
    limits = torch.tensor([w - 1.0, h - 1.0])
    grid = new / limits * 2 - 1
    return F.grid_sample(img, grid, mode='bilinear', padding_mode='zeros', align_corners=False)


def photometric_l1(target, warped):
    """trainer.py:62: per-pixel channel-mean L1, shape (B,1,H,W)."""
    return (target - warped).abs().mean(1, keepdim=True)


# ----------------------------------------------------------------------------------------------
# frame sampler index arithmetic (data.py:55-59,72-76,87-99,112-115) and window gather (data.py:31-45)
# ----------------------------------------------------------------------------------------------
def train_indices(num_lr, fps):
    return list(range(1 + fps, num_lr - fps, 120 // fps))


def all_indices(num_lr, fps):
    return list(range(1 + fps, num_lr - fps))


def val_indices(num_lr, fps, lr_window, k, perm):
    """perm = torch.randperm(num_lr - 2*lr_window) drawn by the caller (data.py:89)."""
    out = []
    for i in perm:
        i = int(i) + lr_window
        if (i + fps + 3) % (120 // fps) == 0:
            continue
        out.append(i)
        if len(out) == k:
            break
    return out


def gather_window(lr_clip_u8, hr_clip_u8, idx, lr_window):
    """data.py:35-40 on an in-memory clip: lr_clip (T,h,w,4) u8, hr_clip (T,H,W,3) u8 -> float CHW /255."""
    win = [lr_clip_u8[x] for x in range(idx - lr_window, idx + lr_window + 1)]
    lr = torch.cat(win, dim=-1).permute(2, 0, 1).float() / 255.0
    hr = hr_clip_u8[idx].permute(2, 0, 1).float() / 255.0
    return hr, lr


# ----------------------------------------------------------------------------------------------
# Adam as configured by lit_wrapper.py:131-138 (torch.optim.Adam: L2 weight decay, not AdamW)
# ----------------------------------------------------------------------------------------------
def adam_step(p, g, m, v, step, lr, beta1, beta2, eps, weight_decay):
    g = g + weight_decay * p
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)
    return p


# ----------------------------------------------------------------------------------------------
# one training step (lit_wrapper.py:29-77) on explicit inputs; returns losses, leaves .grad populated
# ----------------------------------------------------------------------------------------------
def training_step(inn, hr, lr, z, lam, lr_dims, tcr=None):
    """lam: dict(fwd_rec, fwd_mmd, latent_nll, bwd_rec, bwd_mmd).  tcr: optional dict
    (hr_u, lr_u, rands[list of (B,3)], zs[list], weight, angle, trans, scale)."""
    for p in inn.parameters():
        p.grad = None
    lr_z = torch.cat((lr, z), 1)
    out = inn(hr)
    fwd = lam['fwd_rec'] * reconstruction(out[:, :lr_dims], lr)
    fwd = fwd + lam['fwd_mmd'] * mmd(out, lr_z)
    fwd = fwd + lam['latent_nll'] * latent_nll(out[:, lr_dims:])
    fwd.backward()
    hr_hat = inn(lr_z, rev=True)
    bwd = lam['bwd_rec'] * reconstruction(hr_hat, hr)
    bwd = bwd + lam['bwd_mmd'] * mmd(hr_hat, hr, rev=True)
    bwd.backward()
    tcr_loss = torch.zeros(())
    if tcr is not None and tcr['weight'] > 0:
        iters = len(tcr['rands'])
        for rand, zt in zip(tcr['rands'], tcr['zs']):
            lr_zu = torch.cat((tcr['lr_u'], zt), 1)
            warped_in = torch.cat((tcr_warp(tcr['lr_u'], rand, tcr['angle'], tcr['trans'],
                                            scale=1.0 / tcr['scale']), zt), 1)
            a = inn(warped_in, rev=True)
            b = tcr_warp(inn(lr_zu, rev=True), rand, tcr['angle'], tcr['trans'])
            tcr_loss = tcr['weight'] / iters * reconstruction(a, b)
            tcr_loss.backward()
    return fwd.detach(), bwd.detach(), tcr_loss.detach(), out.detach(), hr_hat.detach()


def synthetic_clip(t, h, w, seed_hr=0, seed_lr=1, derived_lr=False):
    """SURVEY 8(d) synthetic inputs: HR u8 (T,H,W,3) seed 0; LR u8 (T,H/8,W/8,4) seed 1 (or box-mean of HR)."""
    g = torch.Generator().manual_seed(seed_hr)
    hr = torch.randint(0, 256, (t, h, w, 3), generator=g, dtype=torch.uint8)
    if derived_lr:
        f = hr.float().reshape(t, h // 8, 8, w // 8, 8, 3).mean(dim=(2, 4))
        lr = torch.cat((f, f[..., 1:2]), -1).round().clamp(0, 255).to(torch.uint8)
    else:
        g = torch.Generator().manual_seed(seed_lr)
        lr = torch.randint(0, 256, (t, h // 8, w // 8, 4), generator=g, dtype=torch.uint8)
    return hr, lr


class IRNOracle(nn.Module):
    """archs.py:201-233 InvRescaleNet: [Haar, (Haar, c x InvBlockExp) x levels]."""

    def __init__(self, c, lr_dims, scale=4, num_coupling=4):
        super().__init__()
        ops, ch = ['haar'], c * 4
        blocks = []
        for _ in range((scale - 1).bit_length()):
            ops.append('haar')
            ch *= 4
            for _ in range(num_coupling):
                blocks.append(InvBlockExpOracle(ch, min(lr_dims, ch // 2)))   # archs.py:218
                ops.append(len(blocks) - 1)
        self.ops = ops
        self.blocks = nn.ModuleList(blocks)

    def forward(self, x, rev=False):
        for op in (reversed(self.ops) if rev else self.ops):
            if op == 'haar':
                x = haar_inv(x) if rev else haar_fwd(x)
            else:
                x = self.blocks[op](x, rev=rev)
        return x


def load_reference_irn_state(oracle_net, ref_state):
    """Map reference InvRescaleNet keys (operations.N.{F,G,H}.convK.*) onto IRNOracle (blocks.M.{F,G,H}.convs.K-1.*)."""
    op_ids = sorted({int(k.split('.')[1]) for k in ref_state if '.conv' in k})
    remap = {op: i for i, op in enumerate(op_ids)}
    new = {}
    for k, v in ref_state.items():
        parts = k.split('.')
        if len(parts) < 5 or not parts[3].startswith('conv'):
            continue                                   # haar_weights buffers
        new[f'blocks.{remap[int(parts[1])]}.{parts[2]}.convs.{int(parts[3][4:]) - 1}.{parts[4]}'] = v
    oracle_net.load_state_dict(new)


# ----------------------------------------------------------------------------------------------
# LR synthesis of datasets/prepare.py (extract_bayer :35-52 without the optional Lanczos resize, binning :54-82,
# quantisation :127-128,164) in numpy float64, the reference's own dtype
# ----------------------------------------------------------------------------------------------
def bayer_planes(frame_u8, scale, reduction):
    """prepare.py:127-128 (u8 -> [0,1] float64), extract_bayer :35-52 (no resize), binning :54-82 -> (mosaic, [R, G1, G2, B] binned
    planes), all float64 like the reference."""
    red = {'mean': np.mean, 'sum': np.sum}[reduction]
    f = frame_u8 / 255
    bayer = np.empty(f.shape[:2])
    bayer[::2, ::2] = f[::2, ::2, 0]
    bayer[::2, 1::2] = f[::2, 1::2, 1]
    bayer[1::2, ::2] = f[1::2, ::2, 1]
    bayer[1::2, 1::2] = f[1::2, 1::2, 2]
    planes = []
    for plane in (bayer[::2, ::2], bayer[::2, 1::2], bayer[1::2, ::2], bayer[1::2, 1::2]):
        ph, pw = plane.shape
        blocks = plane[:, :, None].reshape(ph // scale, scale, pw // scale, scale, 1)
        planes.append(red(red(blocks, 1), -2).squeeze(-1))
    return bayer, planes


def bayer_mosaic(frame_u8, scale=4, reduction='mean'):
    """prepare.py:103-116: the UNQUANTISED binned RGGB planes packed back into one Bayer mosaic (what pack_demosaic hands to the
    demosaicer).  One frame (H,W,3) uint8 -> (H/s, W/s) float64.  Pinned by fixture G8."""
    _, planes = bayer_planes(frame_u8, scale, reduction)
    h, w = planes[0].shape
    cfa = np.empty((2 * h, 2 * w))
    cfa[::2, ::2], cfa[::2, 1::2], cfa[1::2, ::2], cfa[1::2, 1::2] = planes
    return cfa


def bayer_demosaic(hr_u8, scale=4, reduction='mean'):
    """datasets/prepare.py:103-119,158,163-165: pack the UNQUANTISED binned RGGB planes into a Bayer mosaic (pinned, G8) and
    demosaic it bilinearly, then clip and quantise.  The demosaic restates colour_demosaicing 0.1.6 (pinned in requirements.txt,
    absent here -> parity unpinned for this step) `demosaicing_CFA_Bayer_bilinear(CFA, 'RGGB')`:
    R/B = convolve(CFA * mask, [[1,2,1],[2,4,2],[1,2,1]]/4), G = convolve(CFA * mask, [[0,1,0],[1,4,1],[0,1,0]]/4) with
    scipy.ndimage.convolve's default 'reflect' boundary.  hr (T,H,W,3) uint8 -> (T,H/s,W/s,3) uint8."""
    from scipy.ndimage import convolve
    h_g = np.array([[0, 1, 0], [1, 4, 1], [0, 1, 0]], dtype=np.float64) / 4
    h_rb = np.array([[1, 2, 1], [2, 4, 2], [1, 2, 1]], dtype=np.float64) / 4
    out = []
    for frame in hr_u8:
        cfa = bayer_mosaic(frame, scale, reduction)
        r_m = np.zeros_like(cfa); r_m[::2, ::2] = 1
        b_m = np.zeros_like(cfa); b_m[1::2, 1::2] = 1
        g_m = 1 - r_m - b_m
        rgb = np.stack([convolve(cfa * r_m, h_rb), convolve(cfa * g_m, h_g), convolve(cfa * b_m, h_rb)], -1)
        out.append((np.clip(rgb, 0, 1) * 255).astype(np.uint8))
    return np.stack(out)


def bayer_bin(hr_u8, scale=4, reduction='mean'):
    """hr (T,H,W,3) uint8 numpy -> lr (T,H/(2s),W/(2s),4) uint8 (prepare.py:35-82 + the quantisation of :164).  Pinned by G8."""
    out = []
    for frame in hr_u8:
        _, planes = bayer_planes(frame, scale, reduction)
        out.append((np.clip(np.stack(planes, -1), 0, 1) * 255).astype(np.uint8))
    return np.stack(out)
