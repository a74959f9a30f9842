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
import ctypes

import numpy as np
import torch
import torch.nn as nn

from . import _lib, ops
from ._lib import CONV_ADD, CONV_LINEAR, check
from .modules import GATE_TAP, USE_SIDE_STREAM, USE_WINOGRAD, WEIGHTS_EPOCH, _grad_buf, _side_stream, import_nchw

CONV_LRELU, CONV_IRN_FWD, CONV_IRN_INV = 6, 7, 8
GC = 32
SLOPE = 0.2
DEBUG_SYNC = [False]      # diagnostic: device-synchronise after every DenseBlock backward


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
    """Packed weights of the five convs for the padded feature-buffer channel order [x | pad | f1 | f2 | f3 | f4].

    The buffers are persistent and REGISTERED with the model-wide pack registry (sin_inn_amd.modules): the optimiser step
    refreshes every pack of the model -- GLOW and DenseBlock alike -- in one batched launch.  The channel padding (channel_in
    -> a multiple of 8 inside the feature buffer, conv5's outputs -> a multiple of 8) is expressed in the pack descriptor
    (sininn_pack_desc.src_n / gap_begin / gap_len), not by materialising a padded copy of the weight: round 2 re-packed all
    120 convs of an IRN model from Python every step (zero-filled padded weight + two slice copies + two pack launches per
    conv, ~700 tiny launches per step on the main stream in front of the pass chains)."""

    def __init__(self):
        self.entries, self.packs, self.sig = None, None, None

    def get(self, block):
        from .modules import _PACK_REGISTRY, _PackCache, _PackEntry
        convs = block.convs()
        wino = bool(USE_WINOGRAD[0])
        # one cheap signature for the block first (this runs ~100 times per step): optimiser epoch, every parameter's version
        # counter and storage address.  Equal to the signature the packs were last validated with -> nothing to check.
        sig = (WEIGHTS_EPOCH[0], wino) + tuple(v for cv in convs for q in (cv._parameters,)
                                               for v in (q['weight']._version, q['bias']._version, q['weight'].data_ptr()))
        if sig == self.sig:
            return self.packs
        keys = [_PackCache._key(cv, True, wino, wino, False) for cv in convs]
        if self.entries is None or any(e.key[0] != k[0] or e.key[4:] != k[4:] for e, k in zip(self.entries, keys)):
            # first use, another device / storage, or the Winograd switch flipped: (re)allocate and (re)register
            for e in self.entries or ():
                _PACK_REGISTRY.discard(e)
            cin, cinp = block.channel_in, block.cinp
            self.entries = []
            for i, cv in enumerate(convs):
                n = cv.weight.shape[0]
                pad = (_pad8(n), cinp + GC * i, cin, cinp - cin)
                packs = ops.alloc_packs(pad[0], pad[1], 3, None, True, wino, wino, cv.weight.device)
                e = _PackEntry(cv, None, None, packs)
                e.pad = pad
                self.entries.append(e)
                _PACK_REGISTRY.add(e)
            self.packs = [e.packs for e in self.entries]          # ONE list object per allocation: descriptor caches key on it
        if any(e.key != k for e, k in zip(self.entries, keys)):
            # stale (new block, weights changed outside the optimiser): refresh these five with one batched launch
            descs = [ops.pack_desc(cv.weight.detach(), cv.bias.detach(), None, e.packs, wino, wino, e.pad)
                     for e, cv in zip(self.entries, convs)]
            ops.pack_batch_run(ops.pack_batch(descs, convs[0].weight.device))
            for e, k in zip(self.entries, keys):
                e.key = k
        self.sig = sig
        return self.packs


_MODES = {'linear': 0, 'add': 1, 'irn_fwd': 2, 'irn_inv': 3}


def _dense_args(block, packs, b, h, w, mode, clamp):
    """A fresh sininn_dense_args with everything that does not change from call to call filled in: copied from a per-block
    template that is rebuilt only when the pack list (its identity) or the shape / mode changes."""
    key = (b, h, w, mode, float(clamp), bool(USE_WINOGRAD[0]))
    cache = block.__dict__.setdefault('_args_tpl', {})
    hit = cache.get(key)
    if hit is None or hit[0] is not packs:
        a = _lib.DenseArgs(B=b, H=h, W=w, cin=block.channel_in, cout=block.channel_out, mode=_MODES[mode],
                           winograd=int(USE_WINOGRAD[0]), clamp=float(clamp))
        for i, (wf, bf, wd) in enumerate(packs):
            a.w_fwd[i], a.b_fwd[i], a.w_dgrad[i] = wf.data_ptr(), bf.data_ptr(), wd.data_ptr()
        hit = (packs, bytes(a))
        cache[key] = hit
    return _lib.DenseArgs.from_buffer_copy(hit[1])


class _DenseFn(torch.autograd.Function):
    """out = tail(conv5(dense(x))) ; tail in {linear, add(aux1), irn_fwd(v=aux1, h=aux2), irn_inv(v=aux1, h=aux2)}.
    One C-ABI call per pass (sininn_dense_forward / sininn_dense_backward: the launch sequence is issued from C++; the five
    weight gradients of the block run as one grouped launch pair on the weight-gradient stream)."""

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
        buf = torch.empty((m, bw), device=dev, dtype=torch.float32)
        out = torch.empty((b, h, w, cout), device=dev, dtype=torch.float32)
        a = _dense_args(block, packs, b, h, w, mode, clamp)
        a.x, a.x_stride, a.buf, a.out = xd.data_ptr(), xs, buf.data_ptr(), out.data_ptr()
        a.buf_floats, a.out_floats = buf.numel(), out.numel()
        a1 = a2 = None
        if mode != 'linear':
            a1, s1 = _pixel_view(aux1.detach())
            a.aux1, a.aux1_stride = a1.data_ptr(), s1
        if mode in ('irn_fwd', 'irn_inv'):
            a2 = aux2.detach().contiguous()
            a.aux2, a.aux2_floats = a2.data_ptr(), a2.numel()
        check(_lib.lib().sininn_dense_forward(a, ops._stream()))
        if block.__dict__.get('_grad_mode', True) and any(ctx.needs_input_grad):      # the caller's grad mode (autograd is off inside forward)
            ctx.block, ctx.mode, ctx.clamp, ctx.shape = block, mode, clamp, (b, h, w, cin)
            ctx.save_for_backward(buf, out, a1 if a1 is not None else buf, a2 if a2 is not None else buf, xd)
            if GATE_TAP[0] is not None:          # parity tooling: the LeakyReLU gates of conv1-4 (feature slots of buf are > 0)
                feats = buf.view(b, h, w, bw)[..., cinp:]
                GATE_TAP[0].append((block, block.__dict__.get('_tap_rev'),
                                    [(feats[..., GC * i:GC * (i + 1)] > 0).permute(0, 3, 1, 2) for i in range(4)]))
        return out

    @staticmethod
    def backward(ctx, dout):
        block, mode, clamp = ctx.block, ctx.mode, ctx.clamp
        buf, out, a1, a2, xd = ctx.saved_tensors
        b, h, w, cin = ctx.shape
        m = b * h * w
        dev = buf.device
        cinp, cout = block.cinp, block.channel_out
        bw, coutp = cinp + 4 * GC, _pad8(cout)
        packs = block._packs.get(block)
        convs = block.convs()
        dout = dout.contiguous()
        dout.record_stream(torch.cuda.current_stream())     # may have been produced on another chain / helper stream
        lib = _lib.lib()
        irn = mode in ('irn_fwd', 'irn_inv')
        a = _dense_args(block, packs, b, h, w, mode, clamp)
        a.x, a.x_stride = xd.data_ptr(), xd.stride(2)
        a.buf, a.out, a.dout = buf.data_ptr(), out.data_ptr(), dout.data_ptr()
        a.buf_floats, a.out_floats, a.dout_floats = buf.numel(), out.numel(), dout.numel()
        dF = torch.empty((m, bw), device=dev, dtype=torch.float32)        # fully written by conv5's data gradient
        a.dF, a.dF_floats = dF.data_ptr(), dF.numel()
        dD = dh = dv = None
        if irn or coutp != cout:
            dD = torch.empty((m, coutp), device=dev, dtype=torch.float32)
            a.dD, a.dD_floats = dD.data_ptr(), dD.numel()
        if mode != 'linear':
            a.aux1, a.aux1_stride = a1.data_ptr(), a1.stride(2)
        if irn:
            dh = torch.empty((m, cout), device=dev, dtype=torch.float32)
            dv = torch.empty((b, h, w, cout), device=dev, dtype=torch.float32)
            a.aux2, a.dh, a.dv = a2.data_ptr(), dh.data_ptr(), dv.data_ptr()
            a.aux2_floats, a.dh_floats, a.dv_floats = a2.numel(), dh.numel(), dv.numel()
        for i, cv in enumerate(convs):
            if cv.weight.requires_grad:
                a.gw[i], a.gb[i] = _grad_buf(cv.weight).data_ptr(), _grad_buf(cv.bias).data_ptr()
        nbytes = lib.sininn_dense_workspace_bytes(b, h, w, block.channel_in, cout)
        ws = torch.empty((nbytes + 3) // 4, device=dev, dtype=torch.float32)
        a.workspace, a.workspace_bytes = ws.data_ptr(), nbytes
        # weight gradients go to the dedicated side stream (like the GLOW executor's): every `+=` into a parameter gradient
        # is issued on that ONE stream, so two pass chains may run concurrently
        main_h = ops._stream_handle()
        side = _side_stream(dev) if USE_SIDE_STREAM[0] else None
        side_h = side.cuda_stream if side is not None else main_h
        check(lib.sininn_dense_backward(a, ctypes.c_void_p(main_h), ctypes.c_void_p(side_h)))
        if DEBUG_SYNC[0]:
            torch.cuda.synchronize()
        if side is not None and side_h != main_h:
            for t in (buf, dF, dD, ws, dout):
                if t is not None:
                    t.record_stream(side)
        dx = dF.view(b, h, w, bw)[..., :cin]
        g_aux1 = dout if mode == 'add' else (dv if irn else None)
        g_aux2 = dh.view(b, h, w, cout) if irn else None
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
        m = self._modules                      # the plain dict behind self.convK (nn.Module.__getattr__ is slow; hot path)
        return (m['conv1'], m['conv2'], m['conv3'], m['conv4'], m['conv5'])

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
        params = [q[n] for cv in self.convs() for q in (cv._parameters,) for n in ('weight', 'bias')]
        self.__dict__['_grad_mode'] = torch.is_grad_enabled()
        return _DenseFn.apply(x, aux1, aux2, self, mode, float(clamp), *params)

    def forward(self, x):
        """NCHW-shaped in / out (the reference's call convention)."""
        return self.run(import_nchw(x)).permute(0, 3, 1, 2)


class _IrnTailFn(torch.autograd.Function):
    """y = v * exp(s) + g  (inverse: (v - g) / exp(s)),  s = clamp * (2 sigmoid(h) - 1): the InvBlockExp tail as a node of its
    own (sininn_irn_tail / sininn_irn_coupling_bwd).  With the tail outside G's DenseBlock call, H(y1) and G(y1) are
    independent autograd nodes in both directions of differentiation and can run on two streams."""

    @staticmethod
    def forward(ctx, v, h, g, clamp, inverse):
        vd, vs = _pixel_view(v.detach())
        hd, gd = h.detach().contiguous(), g.detach().contiguous()
        b, hh, ww, co = vd.shape
        out = torch.empty((b, hh, ww, co), device=vd.device, dtype=torch.float32)
        check(_lib.lib().sininn_irn_tail(vd.data_ptr(), vs, hd.data_ptr(), gd.data_ptr(), b * hh * ww, co, float(clamp), int(inverse),
                                         out.data_ptr(), co, ops._stream()))
        ctx.clamp, ctx.inverse = float(clamp), int(inverse)
        ctx.save_for_backward(out if inverse else vd, hd)
        return out

    @staticmethod
    def backward(ctx, dout):
        vy, hd = ctx.saved_tensors
        b, hh, ww, co = hd.shape
        m = b * hh * ww
        dout = dout.contiguous()
        vyv, vys = _pixel_view(vy)
        dg = torch.empty((b, hh, ww, co), device=hd.device, dtype=torch.float32)
        dh = torch.empty_like(dg)
        dv = torch.empty_like(dg)
        check(_lib.lib().sininn_irn_coupling_bwd(dout.data_ptr(), co, vyv.data_ptr(), vys, hd.data_ptr(), m, co, ctx.clamp, ctx.inverse,
                                                 dg.data_ptr(), dh.data_ptr(), dv.data_ptr(), co, ops._stream()))
        return dv, dh, dg, None, None


_AUX = {}
# H(y1) on a second stream beside G(y1) in no-grad passes: level-1 DenseBlock convs launch 64 blocks on 256 CUs, so a single pass
# chain leaves most of the chip idle (`bench.py --arch IRN --overlap none / wgrad / full` = 25.0 / 24.7 / 21.1 ms: chain-level
# concurrency is what this architecture responds to).  SININN_IRN_HG=0 switches it off.
import os as _os
HG_OVERLAP = [_os.environ.get('SININN_IRN_HG', '1') != '0']
HG_TRAIN = [_os.environ.get('SININN_IRN_HG_TRAIN', '0') == '1']     # diagnostic: the two-stream block in differentiated passes too (measured slower)


def _aux_stream(device):
    """the helper stream that belongs to the CURRENT stream (each pass chain of a training step gets its own)"""
    key = (str(device), ops._stream_handle())
    if key not in _AUX:
        from .modules import make_stream
        _AUX[key] = make_stream(device, 0, f'IRN H-beside-G helper of stream {key[1]:#x}')
    return _AUX[key]


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
        if GATE_TAP[0] is not None:            # parity tooling: the direction this pass runs in, recorded with the gates
            for blk in (self.F, self.G, self.H):
                blk.__dict__['_tap_rev'] = bool(rev)
        x1, x2 = x[..., :self.split_len1], x[..., self.split_len1:]
        # two streams only for passes that are not differentiated (validation / inference: ONE chain, which leaves the chip
        # underfilled -- IRN inverse pass at batch 40: 7.43 -> 6.46 ms).  A training step already runs two pass chains and the
        # weight-gradient stream; H beside G on top of that measured slower (21.0 -> 23.4 ms: two more streams, the tail as a
        # kernel of its own), so training keeps the single-chain block with the tail fused into G's conv5.
        # Never inside a stream capture: the fork / join below makes the helper stream wait for this stream AND this stream wait
        # for the helper; when this stream is not the capture's origin (the second pass chain of a captured training step, any
        # stream a caller forked), the HIP runtime bundled with torch 2.10+rocm7.0 links the two streams into each other's
        # parallelCaptureStreams_ and hip::Stream::EndCapture() recurses over that 2-cycle until the stack overflows -- the
        # capture_end abort of round 3 (tools/capture_diag.py dumps the lists, tools/capture_topology.hip reproduces it).
        if HG_OVERLAP[0] and x.is_cuda and (HG_TRAIN[0] or not torch.is_grad_enabled()) and \
                (HG_TRAIN[0] or not torch.cuda.is_current_stream_capturing()):
            return self._apply_two_streams(x1, x2, rev)
        if not rev:
            y1 = self.F.run(x2, 'add', x1)                               # y1 = x1 + F(x2)
            hval = self.H.run(y1)                                        # s = clamp*(2*sigmoid(H(y1)) - 1)
            y2 = self.G.run(y1, 'irn_fwd', x2, hval, self.clamp)         # y2 = x2*exp(s) + G(y1)
        else:
            hval = self.H.run(x1)
            y2 = self.G.run(x1, 'irn_inv', x2, hval, self.clamp)         # y2 = (x2 - G(x1)) / exp(s)
            y1 = x1 - self.F.run(y2)                                     # y1 = x1 - F(y2)
        return torch.cat((y1, y2), dim=3)

    def _apply_two_streams(self, x1, x2, rev):
        """The same block with H and G as independent nodes: H runs on the chain's helper stream beside G, the stand-alone tail
        joins them.  Backward mirrors it by itself: the tail's backward yields dh and dG, and autograd runs every node on the
        stream its forward ran on (with the event waits between them), so H's and G's backward overlap as well."""
        main = torch.cuda.current_stream()
        aux = _aux_stream(x1.device)
        cond = x1 if rev else self.F.run(x2, 'add', x1)                  # y1 = x1 + F(x2) in the forward direction
        aux.wait_stream(main)
        with torch.cuda.stream(aux):
            hval = self.H.run(cond)
        cond.record_stream(aux)
        gout = self.G.run(cond)
        main.wait_stream(aux)
        hval.record_stream(main)
        y2 = _IrnTailFn.apply(x2, hval, gout, self.clamp, 1 if rev else 0)
        y1 = (x1 - self.F.run(y2)) if rev else cond
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
                # import_nchw: zero-copy for the pixel-major tensors the ops hand each other, one layout pass for an
                # NCHW-contiguous tensor from outside (e.g. a latent built with torch.cat / loaded from disk, rev=True)
                out = op.apply_pixel_major(import_nchw(out), rev).permute(0, 3, 1, 2)
        return out
