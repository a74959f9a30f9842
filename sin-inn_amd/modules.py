"""HIP-backed equivalents of the FrEIA operators the reference wires together in archs.py:26-71.

Same operator protocol as the (absent, third-party) FrEIA modules the reference uses:
``Op(dims_in, **kwargs)``; ``op([x], rev=False) -> [y]``; ``op.jacobian(x, rev)``;
``op.output_dims(dims)``.  Semantics follow SURVEY.md Appendix A (GLOWCouplingBlock,
PermuteRandom, IRevNetDownsampling).  All arithmetic runs in libsininn.so; tensors are fp32 and
pixel-major (NHWC) inside the network, NCHW-shaped channels_last views at the boundary.
"""
import ctypes as C

import numpy as np
import weakref

import torch
import torch.nn as nn

from . import _lib, ops
from ._lib import GlowArgs, SubnetArgs

HIDDEN = ops.HIDDEN

# bumped by the fused optimiser (it updates weights through raw pointers, invisible to torch's
# tensor version counters) so cached packed weights are rebuilt.
WEIGHTS_EPOCH = [0]


def bump_weights_epoch():
    """Called by the fused optimiser after it has written new weights: invalidates every cached pack and refreshes the
    registered ones with one batched launch on the current stream."""
    WEIGHTS_EPOCH[0] += 1
    _PACK_REGISTRY.refresh()


def _vp(t, off=0, dtype=torch.float32):
    return ops.ptr(t, off, dtype)


def to_pixel_major(x):
    """(B,C,H,W)-shaped tensor of any strides -> contiguous (B,H,W,C) tensor (zero-copy when already channels_last)."""
    b, c, h, w = x.shape
    v = x.permute(0, 2, 3, 1)
    if v.is_contiguous():
        return v
    out = torch.empty((b, h, w, c), device=x.device, dtype=torch.float32)
    ops.squeeze(x, out.permute(0, 3, 1, 2), b, c, h, w, 0, False)
    return out


class _LayoutImport(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return to_pixel_major(x.detach())

    @staticmethod
    def backward(ctx, g):
        return g.permute(0, 3, 1, 2)


def import_nchw(x):
    """autograd-aware to_pixel_major"""
    v = x.permute(0, 2, 3, 1)
    if v.is_contiguous():
        return v
    return _LayoutImport.apply(x)


# ------------------------------------------------------------------------------------------------
# subnet inspection + packed-weight cache
# ------------------------------------------------------------------------------------------------
def inspect_subnet(seq):
    """The HIP path implements exactly archs.py:11-17: Conv(k) -> ReLU -> Conv(k), k in {1,3}, 'same' padding."""
    ok = (isinstance(seq, nn.Sequential) and len(seq) == 3 and isinstance(seq[0], nn.Conv2d)
          and isinstance(seq[1], nn.ReLU) and isinstance(seq[2], nn.Conv2d))
    if ok:
        k = seq[0].kernel_size[0]
        for cv in (seq[0], seq[2]):
            ok = ok and cv.kernel_size == (k, k) and cv.padding == (k // 2, k // 2) and cv.stride == (1, 1) \
                and cv.dilation == (1, 1) and cv.groups == 1 and cv.bias is not None and k in (1, 3)
        ok = ok and seq[0].out_channels == HIDDEN and seq[2].in_channels == HIDDEN
    if not ok:
        raise NotImplementedError('sin-inn_amd GLOWCouplingBlock supports subnet_conv / subnet_conv_1x1 style '
                                  'subnets only (Conv2d(k)->ReLU->Conv2d(k), 256 hidden channels, k in {1,3})')
    return seq[0], seq[2], k


# Winograd F(2x2,3x3) for the 3x3 convs whose packed column count is a multiple of 32 (SININN_WINOGRAD=0 disables)
import os as _os
USE_WINOGRAD = [_os.environ.get('SININN_WINOGRAD', '1') != '0']


class _PackCache:
    """Packed weights of the convs of one block.  Buffers are persistent: a stale entry is refreshed in place, and every
    live entry is registered so that the optimiser step can refresh ALL packs of the model with one launch."""

    def __init__(self):
        self.store = {}

    @staticmethod
    def _key(conv, want_dgrad, wino_fwd, wino_dgrad, bf16=False):
        # (conv._parameters[...]: the plain dict behind conv.weight -- nn.Module.__getattr__ costs more than the rest of the key,
        # and this runs a few hundred times per training step)
        prm = conv._parameters
        w, b = prm['weight'], prm['bias']
        return (w.data_ptr(), w._version, b._version, WEIGHTS_EPOCH[0], want_dgrad, wino_fwd, wino_dgrad, bf16)

    def get(self, conv, colmap, want_dgrad, wino_fwd=False, wino_dgrad=False, bf16=False):
        key = self._key(conv, want_dgrad, wino_fwd, wino_dgrad, bf16)
        hit = self.store.get((id(conv), bf16))
        if hit is None or hit.key != key:
            same_shape = hit is not None and hit.key[0] == key[0] and hit.key[4:] == key[4:]
            if bf16:
                packs = ops.pack_conv_bf16(conv.weight.detach(), conv.bias.detach(), colmap, want_dgrad,
                                           out=hit.packs if same_shape else None)
            else:
                packs = ops.pack_conv(conv.weight.detach(), conv.bias.detach(), colmap, want_dgrad, wino_fwd, wino_dgrad,
                                      out=hit.packs if same_shape else None)
            if same_shape:
                hit.key = key
            else:
                if hit is not None:
                    _PACK_REGISTRY.discard(hit)
                hit = _PackEntry(conv, colmap, key, packs)
                self.store[(id(conv), bf16)] = hit
                # every pack, fp32 (one batched launch) and bf16 (in-place repack per conv), is refreshed INSIDE the optimiser
                # step on the stream that runs it: after a step no pass can miss the cache and repack on its own stream while
                # another stream reads the same buffer
                _PACK_REGISTRY.add(hit)
        return hit.packs


class _PackEntry:
    def __init__(self, conv, colmap, key, packs):
        self.conv = weakref.ref(conv)
        self.colmap, self.key, self.packs = colmap, key, packs


class _PackRegistry:
    """All live pack entries; refresh() re-packs every one of them from the current weights in a single launch."""

    def __init__(self):
        self.entries = []
        self.table = None
        self.generation = 0          # bumped whenever a pack buffer is allocated or dropped: whatever baked pack ADDRESSES into
                                     # something replayable (lit_wrapper's hipGraph cache) keys on it

    def add(self, entry):
        self.entries.append(entry)
        self.table = None
        self.generation += 1

    def discard(self, entry):
        self.entries = [e for e in self.entries if e is not entry]
        self.table = None
        self.generation += 1

    def refresh(self):
        # strong references for the duration of the call: a weakref can die between two derefs (the cyclic GC may run at
        # any allocation, e.g. when a previous model goes away)
        pairs = [(e, e.conv()) for e in self.entries]
        live = [(e, c) for e, c in pairs if c is not None and c.weight.is_cuda]
        if len(live) != len(self.entries):
            self.entries, self.table = [e for e, _ in live], None
        # mixed-precision packs: repacked in place, one (tiny) launch pair per conv -- the launches the next pass's cache
        # miss would have issued, moved onto the optimiser's stream
        for e, c in live:
            if e.key[7]:
                ops.pack_conv_bf16(c.weight.detach(), c.bias.detach(), e.colmap, e.key[4], out=e.packs)
                e.key = _PackCache._key(c, *e.key[4:])
        live = [(e, c) for e, c in live if not e.key[7]]
        if not live:
            return
        ptrs = tuple(c.weight.data_ptr() for _, c in live)
        if self.table is None or self.table[0] != ptrs:
            descs = [ops.pack_desc(c.weight.detach(), c.bias.detach(), e.colmap, e.packs, e.key[5], e.key[6],
                                   getattr(e, 'pad', None)) for e, c in live]
            self.table = (ptrs, ops.pack_batch(descs, live[0][1].weight.device))
        ops.pack_batch_run(self.table[1])
        for e, c in live:
            e.key = _PackCache._key(c, *e.key[4:])


_PACK_REGISTRY = _PackRegistry()


# ------------------------------------------------------------------------------------------------
# side stream for the weight-gradient kernels: wgrad(conv2) / wgrad(conv1) of a half-coupling only depend on
# dr / dh, not on each other's results or on the data-gradient chain, so they run on a second HIP stream and fill
# the prologue / epilogue / tail bubbles of the dgrad convs (and vice versa).  Joined before the optimiser step.
# ------------------------------------------------------------------------------------------------
_SIDE = {}
USE_SIDE_STREAM = [True]
GATE_TAP = [None]       # parity tooling: set to a list -> every differentiable GLOW / DenseBlock forward appends its gates


HELPER_STREAMS = []      # (name, stream) of every stream this package creates: second pass chain, weight gradients, IRN helpers


def make_stream(device, priority, name='helper'):
    """A HIP stream at `priority` (lower = higher priority).  torch.cuda.Stream covers {-1, 0}; anything else (a LOW-priority
    stream, +1 on MI355X) is created through the library and wrapped.  Every stream is registered (HELPER_STREAMS) so that a
    stream capture can be checked for -- and closed over -- work that one of them still holds (join_capturing_helpers)."""
    if priority in (0, -1):
        st = torch.cuda.Stream(device=device, priority=priority)
    else:
        least, greatest = C.c_int(0), C.c_int(0)
        with torch.cuda.device(device):
            _lib.check(_lib.lib().sininn_stream_priority_range(C.byref(least), C.byref(greatest)))
            prio = max(min(priority, least.value), greatest.value)
            handle = C.c_void_p()
            _lib.check(_lib.lib().sininn_stream_create(prio, C.byref(handle)))
        st = torch.cuda.ExternalStream(handle.value, device=device)
    HELPER_STREAMS.append((name, st))
    return st


def join_capturing_helpers():
    """Inside a stream capture, right before it ends: every helper stream of this package that is part of the capture and still
    holds work the capturing (current) stream does not depend on is joined into it (hipStreamEndCapture refuses -- and on ROCm
    7.2 was seen to crash on -- a capture with an unjoined stream).  Returns the names of the streams that needed the join: the
    pass code is supposed to join what it forks, so a non-empty list is a finding, not a routine."""
    cur = torch.cuda.current_stream()
    mine = [(n, s) for n, s in HELPER_STREAMS if s.device == cur.device and s.cuda_stream != cur.cuda_stream]
    if not mine or not torch.cuda.is_current_stream_capturing():
        return []
    handles = (C.c_void_p * len(mine))(*[s.cuda_stream for _, s in mine])
    flags = (C.c_int * len(mine))()
    _lib.check(_lib.lib().sininn_capture_unjoined(C.c_void_p(cur.cuda_stream), handles, len(mine), flags))
    loose = []
    for (name, st), f in zip(mine, flags):
        if f == 2:
            cur.wait_stream(st)
            loose.append(name)
    return loose


def _side_stream(device):
    key = str(device)
    if key not in _SIDE:
        import os
        _SIDE[key] = make_stream(device, int(os.environ.get('SININN_WGRAD_PRIO', '0')), 'weight-gradient stream')
    return _SIDE[key]


def side_stream_if_any(device):
    """The weight-gradient stream of `device` if the executors are using one (every `+=` into a parameter gradient is then
    issued on it and nowhere else), else None (weight gradients run on the callers' streams)."""
    return _SIDE.get(str(device)) if USE_SIDE_STREAM[0] else None


def join_side_streams():
    """Make the current stream wait for all outstanding weight-gradient kernels (call before reading .grad)."""
    for st in _SIDE.values():
        torch.cuda.current_stream(st.device).wait_stream(st)


def _grad_buf(p):
    """The tensor parameter gradients are accumulated into (created on first use, like autograd would).  Only the block
    executors call this, and they issue every `+=` into it on the weight-gradient stream: the parameter is tagged so, which is
    what lets the data-parallel all-reduce order itself behind that stream alone (FusedAdam.flat_grad_buffers)."""
    if p.grad is None:
        p.grad = torch.zeros_like(p, memory_format=torch.contiguous_format)
    assert p.grad.is_contiguous()
    p.__dict__['_sininn_executor_grad'] = True
    return p.grad


# ------------------------------------------------------------------------------------------------
# GLOW coupling block: fused conv-subnet + affine + log-det
# ------------------------------------------------------------------------------------------------
def _pv(t, off=0):
    return None if t is None else t.data_ptr() + 4 * off


def _pd(t):
    """device pointer of a pack (fp32 or bf16)"""
    return None if t is None else t.data_ptr()


def _subnet_args(block, seq, co, dev, need_grad, with_grads):
    """sininn_subnet descriptor of one subnet of `block` + the pack tensors it points into.  Called four times per block pass
    (forward and backward, two subnets): the subnet's structure is inspected once per block, and the descriptor is rebuilt only
    when a pack tuple or a gradient buffer it points to has been replaced (the optimiser refreshes packs IN PLACE)."""
    which = 1 if seq is block._modules['s1'] else 2
    sub = block.__dict__.get('_subinfo')
    if sub is None or sub[0] is not block._modules['s1'] or sub[1] is not block._modules['s2']:
        sub = (block._modules['s1'], block._modules['s2'], inspect_subnet(block._modules['s1']), inspect_subnet(block._modules['s2']))
        block.__dict__['_subinfo'] = sub
        block.__dict__['_subargs'] = {}
    conv1, conv2, k = sub[1 + which]
    cmap = ops.coupling_colmap(co, dev)
    bf16 = block.precision == 'bf16'
    wino = USE_WINOGRAD[0] and k == 3 and not bf16
    wino_w2 = wino                                  # conv2 forward: (s|t) interleave of either width
    p1, p2 = conv1._parameters, conv2._parameters
    # The data-gradient packs are built whenever the conv is trainable, not only when this call needs them: the cache key
    # then does not flip between no_grad (validation) and training passes, so an entry is never REPLACED mid-training --
    # a replacement would be packed lazily on whichever stream first misses (see ReversibleGraphNet.prepare_packs).
    pk1 = block._packs.get(conv1, None, need_grad or p1['weight'].requires_grad, wino, wino, bf16)
    pk2 = block._packs.get(conv2, cmap, need_grad or p2['weight'].requires_grad, wino_w2, wino, bf16)
    grads = None
    if with_grads:
        grads = tuple((_grad_buf(q['weight']), _grad_buf(q['bias'])) if q['weight'].requires_grad else None for q in (p1, p2))
        gkey = tuple(None if g is None else (g[0].data_ptr(), g[1].data_ptr()) for g in grads)
    else:
        gkey = None
    cache = block.__dict__['_subargs']
    hit = cache.get((which, with_grads))
    if hit is not None and hit[0] is pk1 and hit[1] is pk2 and hit[2] == gkey:
        return hit[3], hit[4]
    w1, b1, wd1 = pk1
    w2, b2, wd2 = pk2
    a = SubnetArgs(w1=_pd(w1), b1=_pv(b1), w2=_pd(w2), b2=_pv(b2), w1_dgrad=_pd(wd1), w2_dgrad=_pd(wd2),
                   winograd=(1 if wino else 0) | (2 if wino_w2 else 0) | (12 if wino else 0))
    if with_grads:
        if grads[0] is not None:
            a.gw1, a.gb1 = _pv(grads[0][0]), _pv(grads[0][1])
        if grads[1] is not None:
            a.gw2, a.gb2 = _pv(grads[1][0]), _pv(grads[1][1])
    keep = (w1, b1, wd1, w2, b2, wd2)
    cache[(which, with_grads)] = (pk1, pk2, gkey, a, keep)
    return a, keep


class _GlowFn(torch.autograd.Function):
    """x (B,H,W,C) pixel-major -> (out (B,H,W,C) with channel c stored at dst[c], logdet (B,)).
    One C-ABI call per pass (sininn_glow_forward / sininn_glow_backward)."""

    @staticmethod
    def forward(ctx, x, block, rev, dst, *params):
        x = x.detach()
        assert x.is_contiguous() and x.dim() == 4 and x.dtype == torch.float32
        if not x.is_cuda:
            raise NotImplementedError('sin-inn_amd ops run on the GPU only (got a CPU tensor)')
        b, h, w, c = x.shape
        dev = x.device
        lib = _lib.lib()
        # nothing is kept under torch.no_grad().  ctx.needs_input_grad reports the parameters' requires_grad whatever the grad
        # mode, and inside Function.forward autograd is always off: the caller's grad mode is sampled by apply_pixel_major
        need_grad = block.__dict__.get('_grad_mode', True) and any(ctx.needs_input_grad)
        out = torch.empty_like(x)
        # zero-initialised log-det accumulator: a row of the buffer the graph executor zeroed once for the whole pass
        # (one fill instead of one per block), else a fresh tensor
        logdet, block._ld_row = block._ld_row, None
        if logdet is None or logdet.shape != (b,) or logdet.device != dev:
            logdet = torch.zeros(b, device=dev, dtype=torch.float32)
        saved = torch.empty(lib.sininn_glow_saved_floats_dtype(b, h, w, c, 1 if block.precision == 'bf16' else 0), device=dev,
                            dtype=torch.float32)
        s1, keep1 = _subnet_args(block, block.s1, block.split_len2, dev, need_grad, False)
        s2, keep2 = _subnet_args(block, block.s2, block.split_len1, dev, need_grad, False)
        a = GlowArgs(B=b, H=h, W=w, C=c, ksize=block.ksize, rev=1 if rev else 0, clamp=block.clamp, x=_pv(x),
                     out=_pv(out), dst_map=_pv(dst), logdet=_pv(logdet), s1=s1, s2=s2, saved=_pv(saved),
                     dtype=1 if block.precision == 'bf16' else 0, no_save=0 if need_grad else 1)
        _lib.check(lib.sininn_glow_forward(C.byref(a), ops._stream()))
        if need_grad:
            ctx.block, ctx.rev, ctx.dst = block, rev, dst
            ctx.save_for_backward(x, out, saved)
            if GATE_TAP[0] is not None:          # parity tooling: export the ReLU gates this pass took (tests/test_gpu_gates.py)
                gates = {}
                for which, name in enumerate(('s1', 's2') if rev else ('s2', 's1')):
                    g = torch.empty((b, h, w, HIDDEN), device=dev, dtype=torch.uint8)
                    _lib.check(lib.sininn_glow_hidden_gates(C.byref(a), which, C.c_void_p(g.data_ptr()), ops._stream()))
                    gates[name] = g.permute(0, 3, 1, 2)
                GATE_TAP[0].append((block, bool(rev), gates))
        ctx.set_materialize_grads(False)
        return out, logdet

    @staticmethod
    def backward(ctx, dout, gld):
        block, rev, dst = ctx.block, ctx.rev, ctx.dst
        x, out, saved = ctx.saved_tensors
        b, h, w, c = x.shape
        dev = x.device
        lib = _lib.lib()
        if dout is None:
            dout = torch.zeros_like(x)
        dout = dout.contiguous()
        if gld is not None:
            gld = gld.contiguous()
        dx = torch.empty_like(x)
        nbytes = lib.sininn_glow_scratch_bytes_dtype(b, h, w, c, block.ksize, 1 if block.precision == 'bf16' else 0)
        scratch = torch.empty((nbytes + 3) // 4, device=dev, dtype=torch.float32)
        s1, keep1 = _subnet_args(block, block.s1, block.split_len2, dev, True, True)
        s2, keep2 = _subnet_args(block, block.s2, block.split_len1, dev, True, True)
        a = GlowArgs(B=b, H=h, W=w, C=c, ksize=block.ksize, rev=1 if rev else 0, clamp=block.clamp, x=_pv(x),
                     out=_pv(out), dst_map=_pv(dst), s1=s1, s2=s2, saved=_pv(saved), scratch=_pv(scratch),
                     scratch_bytes=nbytes, dout=_pv(dout), gld=_pv(gld), dx=_pv(dx),
                     skip_dx=0 if ctx.needs_input_grad[0] else 1,     # first block of a pass: nobody consumes dx
                     dtype=1 if block.precision == 'bf16' else 0)
        main_h = ops._stream_handle()
        side = _side_stream(dev) if USE_SIDE_STREAM[0] else None
        side_h = side.cuda_stream if side is not None else main_h
        _lib.check(lib.sininn_glow_backward(C.byref(a), C.c_void_p(main_h), C.c_void_p(side_h)))
        if side is not None and side_h != main_h:                # the weight-gradient kernels on the side stream still read these
            for t in (scratch, saved, x) + keep1 + keep2:
                if t is not None:
                    t.record_stream(side)
        return (dx if ctx.needs_input_grad[0] else None, None, None, None) + (None,) * (len(ctx.needs_input_grad) - 4)


class GLOWCouplingBlock(nn.Module):
    """FrEIA GLOWCouplingBlock, as used at archs.py:61-64 with clamp=1.2."""

    def __init__(self, dims_in, dims_c=[], subnet_constructor=None, clamp=5.):
        super().__init__()
        assert not dims_c, 'conditional coupling is not on the sin-inn path'
        channels = dims_in[0][0]
        assert len(dims_in[0]) == 3, 'image tensors (C,H,W) only'
        self.split_len1 = channels // 2
        self.split_len2 = channels - channels // 2
        assert self.split_len1 % 8 == 0 and self.split_len2 % 8 == 0, \
            'the HIP coupling kernel needs both halves to be multiples of 8 channels'
        self.clamp = float(clamp)
        self.s1 = subnet_constructor(self.split_len1, self.split_len2 * 2)
        self.s2 = subnet_constructor(self.split_len2, self.split_len1 * 2)
        self.ksize = inspect_subnet(self.s1)[2]
        assert inspect_subnet(self.s2)[2] == self.ksize
        self._packs = _PackCache()
        self.last_jac = None
        # 'fp32': f32 MFMA everywhere (the reference's arithmetic).  'bf16': the conv subnets run on bf16 MFMA with fp32
        # accumulation and bf16 hidden tensors; the flow tensors, the coupling arithmetic and the gradients stay fp32.
        self.precision = 'fp32'

    def _params(self):
        ps = self.__dict__.get('_param_list')
        subs = (self._modules['s1'], self._modules['s2'])
        if ps is None or ps[0] is not subs[0] or ps[1] is not subs[1]:
            ps = (subs[0], subs[1], [p for s in subs for p in s.parameters()])
            self.__dict__['_param_list'] = ps
        return ps[2]

    _ld_row = None          # set by ReversibleGraphNet.forward for the next call only

    def apply_pixel_major(self, x, rev=False, dst=None):
        self.__dict__['_grad_mode'] = torch.is_grad_enabled()       # plain attribute: nn.Module.__setattr__ is slow (hot path)
        out, logdet = _GlowFn.apply(x, self, bool(rev), dst, *self._params())
        self.last_jac = logdet
        return out

    def forward(self, x, c=[], rev=False):
        y = self.apply_pixel_major(import_nchw(x[0]), rev=rev)
        return [y.permute(0, 3, 1, 2)]

    def jacobian(self, x, c=[], rev=False):
        return self.last_jac

    def output_dims(self, input_dims):
        return input_dims


# ------------------------------------------------------------------------------------------------
# index-map operators
# ------------------------------------------------------------------------------------------------
class _SqueezeFn(torch.autograd.Function):
    """levels x IRevNetDownsampling (or its inverse) between arbitrary-stride 4-D tensors, optional channel map."""

    @staticmethod
    def forward(ctx, x, levels, inverse, chan_map, map_on_out, out_pixel_major):
        x = x.detach()
        b, c, h, w = x.shape                       # logical NCHW shape of the input
        f = 1 << levels
        if not inverse:
            fc, fh, fw = c, h, w
            oc, oh, ow = c * f * f, h // f, w // f
        else:
            fc, fh, fw = c // (f * f), h * f, w * f
            oc, oh, ow = fc, fh, fw
        if out_pixel_major:
            out = torch.empty((b, oh, ow, oc), device=x.device, dtype=torch.float32).permute(0, 3, 1, 2)
        else:
            out = torch.empty((b, oc, oh, ow), device=x.device, dtype=torch.float32)
        ops.squeeze(x, out, b, fc, fh, fw, levels, inverse, chan_map, map_on_out)
        ctx.cfg = (levels, inverse, chan_map, map_on_out, x.permute(0, 2, 3, 1).is_contiguous())
        return out

    @staticmethod
    def backward(ctx, g):
        levels, inverse, chan_map, map_on_out, in_pm = ctx.cfg
        # the adjoint of a permutation is its inverse: swap direction, keep the map on the same tensor side
        gx = _SqueezeFn.apply(g, levels, not inverse, chan_map, not map_on_out, in_pm)
        return gx, None, None, None, None, None


def squeeze_op(x, levels, inverse=False, chan_map=None, map_on_out=False, out_pixel_major=True):
    """x: (B,C,H,W)-shaped (any strides) -> (B,C',H',W')-shaped; pixel-major storage unless told otherwise."""
    return _SqueezeFn.apply(x, levels, inverse, chan_map, map_on_out, out_pixel_major)


class IRevNetDownsampling(nn.Module):
    """FrEIA IRevNetDownsampling: out[b,(hb*2+wb)*C+c,i,j] = in[b,c,2i+hb,2j+wb] (archs.py:28-31,35-38)."""

    def __init__(self, dims_in):
        super().__init__()
        self.last_jac = 0.

    def forward(self, x, rev=False):
        return [squeeze_op(x[0], 1, inverse=rev)]

    def jacobian(self, x, rev=False):
        return 0.

    def output_dims(self, input_dims):
        c, h, w = input_dims[0]
        return [(c * 4, h // 2, w // 2)]


class PermuteRandom(nn.Module):
    """FrEIA PermuteRandom(seed): np.random.seed(seed); perm = np.random.permutation(C) (archs.py:65-68)."""

    def __init__(self, dims_in, seed):
        super().__init__()
        self.in_channels = dims_in[0][0]
        np.random.seed(seed)
        self.perm = np.random.permutation(self.in_channels)
        np.random.seed()
        self.perm_inv = np.zeros_like(self.perm)
        for i, p in enumerate(self.perm):
            self.perm_inv[p] = i
        self._dev = {}

    def maps(self, device):
        """(perm, perm_inv) as device int32 tensors."""
        key = str(device)
        if key not in self._dev:
            self._dev[key] = (torch.as_tensor(self.perm.astype(np.int32), device=device),
                              torch.as_tensor(self.perm_inv.astype(np.int32), device=device))
        return self._dev[key]

    def forward(self, x, rev=False):
        perm, perm_inv = self.maps(x[0].device)
        # out[:, j] = in[:, idx[j]]  ==  read-side channel map on a 0-level "squeeze"
        return [squeeze_op(x[0], 0, inverse=False, chan_map=perm_inv if rev else perm, map_on_out=False)]

    def jacobian(self, x, rev=False):
        return 0.

    def output_dims(self, input_dims):
        return input_dims
