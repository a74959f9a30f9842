"""IRN architecture on the HIP kernels (reference archs.py:74-233): HaarDownsampling, DenseBlock, InvBlockExp,
InvRescaleNet.  Same class names, constructor signatures, parameter names (``operations.N.F.conv1.weight`` ...)
and initialisation as the reference, so seeded construction reproduces its weights and its checkpoints load.

Execution is pixel-major (NHWC) like the SRF path and re-uses the same conv / wgrad engine:
  * DenseBlock: the five dense-connected 3x3 convs write their 32-channel outputs into channel slots of ONE
    feature buffer [pixels][cin_pad + 128] (the torch.cat chain of archs.py:90-94 costs nothing); LeakyReLU is a
    conv epilogue; conv5 carries the InvBlockExp tail as its epilogue (ADD: y1 = x1 + F(x2);
    IRN_FWD / IRN_INV: y2 = x2*exp(s) + G(y1) and its inverse, s = clamp*(2*sigmoid(H(y1)) - 1)).
  * backward: hand-written chain (irn_coupling_bwd -> wgrad/dgrad of conv5 -> [lrelu_bwd, wgrad, dgrad-accumulate]
    for conv4..conv1) on the same kernels.
"""
import numpy as np
import torch
import torch.nn as nn

from . import _lib, ops
from ._lib import CONV_ADD, CONV_LINEAR, check
from .modules import USE_SIDE_STREAM, USE_WINOGRAD, WEIGHTS_EPOCH, _grad_buf, _side_stream, import_nchw

CONV_LRELU, CONV_IRN_FWD, CONV_IRN_INV = 6, 7, 8
GC = 32
SLOPE = 0.2


def _vp(t, off=0, dtype=torch.float32):
    return ops.ptr(t, off, dtype)


def _pixel_view(x):
    """(B,H,W,c) channel-slice view of a pixel-major tensor -> (tensor, pixel stride); rows must be dense in pixels."""
    b, h, w, c = x.shape
    assert x.stride(3) == 1 and x.stride(1) == w * x.stride(2) and x.stride(0) == h * w * x.stride(2), \
        'expected a channel slice of a contiguous (B,H,W,C) tensor'
    return x, x.stride(2)


# ------------------------------------------------------------------------------------------------
# Haar (archs.py:162-199)
# ------------------------------------------------------------------------------------------------
class _HaarFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, rev):
        x = x.detach()
        b, c, h, w = x.shape                       # NCHW-shaped, any strides
        if not rev:
            out = torch.empty((b, h // 2, w // 2, 4 * c), device=x.device, dtype=torch.float32).permute(0, 3, 1, 2)
            check(_lib.lib().sininn_haar(_vp(x), ops.strides4(x), _vp(out), ops.strides4(out), b, c, h, w, 0,
                                         ops._stream()))
        else:
            out = torch.empty((b, 2 * h, 2 * w, c // 4), device=x.device, dtype=torch.float32).permute(0, 3, 1, 2)
            check(_lib.lib().sininn_haar(_vp(x), ops.strides4(x), _vp(out), ops.strides4(out), b, c // 4, 2 * h, 2 * w, 1,
                                         ops._stream()))
        ctx.rev = rev
        return out

    @staticmethod
    def backward(ctx, g):
        # analysis A = S^T / 4 (S = synthesis): d/dx of A x is S g / 4 ; d/dy of S y is 4 A g
        gx = _HaarFn.apply(g, not ctx.rev)
        return gx * (4.0 if ctx.rev else 0.25), None


class HaarDownsampling(nn.Module):
    def __init__(self, channel_in):
        super().__init__()
        self.channel_in = channel_in
        w = torch.ones(4, 1, 2, 2)
        w[1, 0, 0, 1] = w[1, 0, 1, 1] = -1
        w[2, 0, 1, 0] = w[2, 0, 1, 1] = -1
        w[3, 0, 1, 0] = w[3, 0, 0, 1] = -1
        # kept only so the state-dict matches the reference (archs.py:179-181); the kernel hard-codes the filters
        self.haar_weights = nn.Parameter(torch.cat([w] * channel_in, 0), requires_grad=False)

    def forward(self, x, rev=False):
        c = x.shape[1]
        self.elements = x.shape[1] * x.shape[2] * x.shape[3]
        self.last_jac = self.elements / 4 * np.log(16. if rev else 1 / 16.)
        assert c == (self.channel_in * 4 if rev else self.channel_in)
        return _HaarFn.apply(x, bool(rev))


# ------------------------------------------------------------------------------------------------
# DenseBlock (archs.py:74-133) with fused tails
# ------------------------------------------------------------------------------------------------
def _pad8(n):
    return (n + 7) // 8 * 8


class _DensePacks:
    """Packed weights of the five convs for the padded feature-buffer channel order [x | pad | f1 | f2 | f3 | f4]."""

    def __init__(self):
        self.key, self.packs = None, None

    def get(self, block):
        convs = block.convs()
        key = tuple((c.weight.data_ptr(), c.weight._version, c.bias._version) for c in convs) + (WEIGHTS_EPOCH[0], USE_WINOGRAD[0])
        if key != self.key:
            cin, cinp = block.channel_in, block.cinp
            packs = []
            for i, cv in enumerate(convs):
                w = cv.weight.detach()
                bias = cv.bias.detach()
                nout = _pad8(w.shape[0])                       # conv5: its data-gradient conv has K = cout
                if cinp != cin or nout != w.shape[0]:
                    wp = torch.zeros((nout, cinp + GC * i, 3, 3), device=w.device, dtype=torch.float32)
                    wp[:w.shape[0], :cin] = w[:, :cin]
                    if i:
                        wp[:w.shape[0], cinp:] = w[:, cin:]
                    bp = torch.zeros(nout, device=w.device, dtype=torch.float32)
                    bp[:w.shape[0]] = bias
                    w, bias = wp, bp
                # all five convs are 3x3: Winograd F(2x2,3x3) packs for the forward and the data-gradient conv
                packs.append(ops.pack_conv(w.contiguous(), bias.contiguous(), None, True, wino_fwd=USE_WINOGRAD[0],
                                           wino_dgrad=USE_WINOGRAD[0]))
            self.key, self.packs = key, packs
        return self.packs


class _DenseFn(torch.autograd.Function):
    """out = tail(conv5(dense(x))) ; tail in {linear, add(aux1), irn_fwd(v=aux1, h=aux2), irn_inv(v=aux1, h=aux2)}."""

    @staticmethod
    def forward(ctx, x, aux1, aux2, block, mode, clamp, *params):
        dev = x.device
        if not x.is_cuda:
            raise NotImplementedError('sin-inn_amd ops run on the GPU only (got a CPU tensor)')
        xd, xs = _pixel_view(x.detach())
        b, h, w, cin = xd.shape
        m = b * h * w
        cinp, cout = block.cinp, block.channel_out
        bw = cinp + 4 * GC
        packs = block._packs.get(block)
        # only the pad channels [cin, cinp) must be zero: x fills [0, cin), the four convs their 32-channel slots
        buf = torch.empty((m, bw), device=dev, dtype=torch.float32)
        if cinp != cin:
            buf[:, cin:cinp].zero_()
        ar = block.arange(dev)
        check(_lib.lib().sininn_permute_channels(_vp(xd), xs, _vp(buf), bw, m, cin, _vp(ar, dtype=torch.int32),
                                                 ops._stream()))
        for i in range(4):
            k = cinp + GC * i
            wf, bf, _ = packs[i]
            ops.conv(in_=_vp(buf), in_stride=bw, Cin=k, w=_vp(wf), bias=_vp(bf), Np=GC, B=b, H=h, W=w, ksize=3,
                     winograd=int(USE_WINOGRAD[0]), mode=CONV_LRELU, clamp=SLOPE, out=_vp(buf, k), out_stride=bw, N=GC)
        wf, bf, _ = packs[4]
        out = torch.empty((b, h, w, cout), device=dev, dtype=torch.float32)
        kw = dict(in_=_vp(buf), in_stride=bw, Cin=bw, w=_vp(wf), bias=_vp(bf), Np=ops.pad16(_pad8(cout)), B=b, H=h, W=w,
                  ksize=3, winograd=int(USE_WINOGRAD[0]), out=_vp(out), out_stride=cout, N=cout)
        a1 = a2 = None
        if mode == 'linear':
            kw.update(mode=CONV_LINEAR)
        elif mode == 'add':
            a1, s1 = _pixel_view(aux1.detach())
            kw.update(mode=CONV_ADD, addend=_vp(a1), addend_stride=s1)
        else:
            a1, s1 = _pixel_view(aux1.detach())
            a2 = aux2.detach().contiguous()
            kw.update(mode=CONV_IRN_FWD if mode == 'irn_fwd' else CONV_IRN_INV, v=_vp(a1), v_stride=s1,
                      mask=_vp(a2), mask_stride=cout, clamp=clamp)
        ops.conv(**kw)
        if any(ctx.needs_input_grad):
            ctx.block, ctx.mode, ctx.clamp, ctx.shape = block, mode, clamp, (b, h, w, cin)
            ctx.save_for_backward(buf, out, a1 if a1 is not None else buf, a2 if a2 is not None else buf)
        return out

    @staticmethod
    def backward(ctx, dout):
        block, mode, clamp = ctx.block, ctx.mode, ctx.clamp
        buf, out, a1, a2 = ctx.saved_tensors
        b, h, w, cin = ctx.shape
        m = b * h * w
        dev = buf.device
        cinp, cout = block.cinp, block.channel_out
        bw = cinp + 4 * GC
        packs = block._packs.get(block)
        convs = block.convs()
        dout = dout.contiguous()
        coutp = _pad8(cout)
        g_aux1 = g_aux2 = None
        lib = _lib.lib()

        def padded(t):                       # [m][cout] -> [m][coutp] with zero pad columns (K of the conv5 dgrad)
            if coutp == cout:
                return t
            tp = torch.zeros((m, coutp), device=dev, dtype=torch.float32)
            check(lib.sininn_permute_channels(_vp(t), cout, _vp(tp), coutp, m, cout,
                                              _vp(block.arange_out(dev), dtype=torch.int32), ops._stream()))
            return tp

        if mode == 'linear':
            dD = padded(dout)
        elif mode == 'add':
            dD, g_aux1 = padded(dout), dout
        else:
            inv = 1 if mode == 'irn_inv' else 0
            dG = torch.empty((m, cout), device=dev, dtype=torch.float32)
            dh = torch.empty((m, cout), device=dev, dtype=torch.float32)
            dv = torch.empty((b, h, w, cout), device=dev, dtype=torch.float32)
            vy, vs = (out, cout) if inv else _pixel_view(a1)
            check(lib.sininn_irn_coupling_bwd(_vp(dout), cout, _vp(vy), vs, _vp(a2), m, cout, clamp, inv, _vp(dG),
                                              _vp(dh), _vp(dv), cout, ops._stream()))
            dD = padded(dG)
            g_aux1, g_aux2 = dv, dh.view(b, h, w, cout)
        dF = torch.empty((m, bw), device=dev, dtype=torch.float32)      # fully written by conv5's data gradient below

        # weight gradients go to the dedicated side stream (like the GLOW executor's): they only read `buf` and a slice of
        # the gradient buffer that is final by then, so they overlap the data-gradient chain, and every `+=` into a
        # parameter gradient is issued on that ONE stream (two pass chains may then run concurrently)
        main = torch.cuda.current_stream()
        side = _side_stream(dev) if USE_SIDE_STREAM[0] else main

        def wgrad(i, k, dout_t, dout_off, dout_stride, n):
            cv = convs[i]
            if not cv.weight.requires_grad:
                return
            if side is not main:
                side.wait_stream(main)
            with torch.cuda.stream(side):
                if cinp == cin:
                    ops.wgrad(buf, 0, bw, k, dout_t, dout_stride, n, b, h, w, 3, _grad_buf(cv.weight), _grad_buf(cv.bias),
                              dout_off=dout_off)
                else:   # gradient w.r.t. the channel-padded weight, then drop the pad channels
                    gwp = torch.zeros((n, k, 3, 3), device=dev, dtype=torch.float32)
                    ops.wgrad(buf, 0, bw, k, dout_t, dout_stride, n, b, h, w, 3, gwp, _grad_buf(cv.bias), dout_off=dout_off)
                    gw = _grad_buf(cv.weight)
                    gw[:, :cin] += gwp[:, :cin]
                    if i:
                        gw[:, cin:] += gwp[:, cinp:]
            if side is not main:
                for t in (buf, dout_t):
                    t.record_stream(side)

        # conv5: weight gradient, then dF[:, :bw] = its data gradient
        wgrad(4, bw, dD, 0, coutp, cout)
        _dgrad(dD, 0, coutp, coutp, packs[4][2], bw, dF, bw, b, h, w, accumulate=False)
        for i in (3, 2, 1, 0):
            k = cinp + GC * i
            check(lib.sininn_lrelu_bwd(_vp(dF, k), bw, _vp(buf, k), bw, m, GC, SLOPE, ops._stream()))
            wgrad(i, k, dF, k, bw, GC)
            _dgrad(dF, k, bw, GC, packs[i][2], k, dF, bw, b, h, w, accumulate=True)
        dx = dF.view(b, h, w, bw)[..., :cin]
        return (dx, g_aux1, g_aux2, None, None, None) + (None,) * 10


def _dgrad(src, src_off, src_stride, n_src, w_dgrad, n_out, dst, dst_stride, b, h, w, accumulate):
    """dst[:, :n_out] (+)= conv(src[:, src_off : src_off+n_src], w_dgrad)   (data gradient of one dense conv)."""
    wino = USE_WINOGRAD[0]
    kw = dict(in_=_vp(src, src_off), in_stride=src_stride, Cin=n_src, w=_vp(w_dgrad),
              Np=ops.pad32(n_out) if wino else ops.pad16(n_out), winograd=int(wino), B=b, H=h, W=w, ksize=3, out=_vp(dst), out_stride=dst_stride, N=n_out)
    if accumulate:
        kw.update(mode=CONV_ADD, addend=_vp(dst), addend_stride=dst_stride)
    else:
        kw.update(mode=CONV_LINEAR)
    ops.conv(**kw)


class DenseBlock(nn.Module):
    def __init__(self, channel_in, channel_out, init='xavier', gc=32, bias=True):
        super().__init__()
        assert gc == GC and bias, 'the HIP DenseBlock implements the reference configuration (gc=32, bias)'
        assert channel_in % 4 == 0 and channel_out % 4 == 0, 'channel counts must be multiples of 4'
        self.channel_in, self.channel_out = channel_in, channel_out
        self.cinp = _pad8(channel_in)
        self.conv1 = nn.Conv2d(channel_in, gc, 3, 1, 1, bias=bias)
        self.conv2 = nn.Conv2d(channel_in + gc, gc, 3, 1, 1, bias=bias)
        self.conv3 = nn.Conv2d(channel_in + 2 * gc, gc, 3, 1, 1, bias=bias)
        self.conv4 = nn.Conv2d(channel_in + 3 * gc, gc, 3, 1, 1, bias=bias)
        self.conv5 = nn.Conv2d(channel_in + 4 * gc, channel_out, 3, 1, 1, bias=bias)
        self.lrelu = nn.LeakyReLU(negative_slope=SLOPE, inplace=True)
        # reference initialisation (archs.py:84-86,100-132), same RNG draw order
        if init == 'xavier':
            for cv in (self.conv1, self.conv2, self.conv3, self.conv4):
                nn.init.xavier_normal_(cv.weight)
                cv.weight.data *= 0.1
                cv.bias.data.zero_()
        nn.init.kaiming_normal_(self.conv5.weight, a=0, mode='fan_in')
        self.conv5.weight.data *= 0
        self.conv5.bias.data.zero_()
        self._packs = _DensePacks()
        self._ar = {}

    def convs(self):
        return (self.conv1, self.conv2, self.conv3, self.conv4, self.conv5)

    def arange(self, dev):
        key = str(dev)
        if key not in self._ar:
            self._ar[key] = torch.arange(self.channel_in, dtype=torch.int32, device=dev)
        return self._ar[key]

    def arange_out(self, dev):
        key = 'o' + str(dev)
        if key not in self._ar:
            self._ar[key] = torch.arange(self.channel_out, dtype=torch.int32, device=dev)
        return self._ar[key]

    def run(self, x, mode='linear', aux1=None, aux2=None, clamp=1.0):
        params = [p for cv in self.convs() for p in (cv.weight, cv.bias)]
        return _DenseFn.apply(x, aux1, aux2, self, mode, float(clamp), *params)

    def forward(self, x):
        """NCHW-shaped in / out (the reference's call convention)."""
        return self.run(import_nchw(x)).permute(0, 3, 1, 2)


class InvBlockExp(nn.Module):
    def __init__(self, channel_num, channel_split_num, clamp=1.):
        super().__init__()
        self.split_len1 = channel_split_num
        self.split_len2 = channel_num - channel_split_num
        self.clamp = clamp
        self.F = DenseBlock(self.split_len2, self.split_len1)
        self.G = DenseBlock(self.split_len1, self.split_len2)
        self.H = DenseBlock(self.split_len1, self.split_len2)

    def apply_pixel_major(self, x, rev=False):
        x1, x2 = x[..., :self.split_len1], x[..., self.split_len1:]
        if not rev:
            y1 = self.F.run(x2, 'add', x1)                               # y1 = x1 + F(x2)
            hval = self.H.run(y1)                                        # s = clamp*(2*sigmoid(H(y1)) - 1)
            y2 = self.G.run(y1, 'irn_fwd', x2, hval, self.clamp)         # y2 = x2*exp(s) + G(y1)
        else:
            hval = self.H.run(x1)
            y2 = self.G.run(x1, 'irn_inv', x2, hval, self.clamp)         # y2 = (x2 - G(x1)) / exp(s)
            y1 = x1 - self.F.run(y2)                                     # y1 = x1 - F(y2)
        return torch.cat((y1, y2), dim=3)

    def forward(self, x, rev=False):
        return self.apply_pixel_major(import_nchw(x), rev).permute(0, 3, 1, 2)


class InvRescaleNet(nn.Module):
    def __init__(self, c, h, w, opt):
        super().__init__()
        channel_out = opt.lr_dims
        operations = [HaarDownsampling(c)]
        current = c * 4
        for _ in range((opt.scale - 1).bit_length()):
            operations.append(HaarDownsampling(current))
            current *= 4
            for _ in range(opt.num_coupling):
                operations.append(InvBlockExp(current, min(channel_out, current // 2)))
        self.operations = nn.ModuleList(operations)

    def prepare_packs(self):
        """Build every DenseBlock's packed weights on the CURRENT stream (they are keyed on the optimiser epoch, i.e.
        rebuilt once per step): called on the main stream before the two pass chains fork, see
        ReversibleGraphNet.prepare_packs."""
        for op in self.operations:
            if isinstance(op, InvBlockExp):
                for blk in (op.F, op.G, op.H):
                    blk._packs.get(blk)

    @property
    def concurrent_passes_safe(self):
        """Two pass chains may run on two streams at once: all parameter gradients accumulate on the side stream."""
        return bool(USE_SIDE_STREAM[0])

    def forward(self, x, rev=False):
        if not x.is_cuda:
            raise NotImplementedError('sin-inn_amd runs on the GPU only: move the module and its inputs to cuda')
        out = x
        for op in (reversed(self.operations) if rev else self.operations):
            if isinstance(op, HaarDownsampling):
                out = op(out, rev)
            else:
                out = op.apply_pixel_major(out.permute(0, 2, 3, 1), rev).permute(0, 3, 1, 2)
        return out
