"""Thin torch-tensor front end of the C ABI: argument checking + pointer plumbing only.

Every function launches asynchronously on torch's current HIP stream.  CPU tensors are refused
(``NotImplementedError``, like softsplat.py:331 in the reference) -- there is no fallback.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import ConvArgs, I64x4, check

HIDDEN = 256


def _stream_handle():
    """raw hipStream_t (int) of torch's current stream on the current device.  torch.cuda.current_stream() builds a Stream
    object through three layers of device-index helpers (~10 us, a hundred times per training step); the C binding is 0.3 us."""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def _stream():
    return C.c_void_p(_stream_handle())


def _chk(t, dtype=torch.float32):
    if not t.is_cuda:
        raise NotImplementedError('sin-inn_amd ops run on the GPU only (got a CPU tensor)')
    assert t.dtype == dtype, f'expected {dtype}, got {t.dtype}'
    return t


def ptr(t, offset=0, dtype=torch.float32):
    """Device pointer of element `offset` (in elements) of tensor t; None -> NULL."""
    if t is None:
        return None
    _chk(t, dtype)
    return C.c_void_p(t.data_ptr() + offset * t.element_size())


def strides4(t):
    assert t.dim() == 4
    return I64x4(*t.stride())


def pad16(n):
    return (n + 15) // 16 * 16


# ---- weights -------------------------------------------------------------------------------------
_colmap_cache = {}


def coupling_tile(co):
    """MFMA column-tile width the coupling conv of a `co`-channel half runs with (32 when co % 16 == 0)."""
    return 32 if co % 16 == 0 else 16


def coupling_colmap(co, device):
    """Device int32 [2*co]: packed column -> subnet output channel, (s|t) interleaved per MFMA column tile."""
    key = (co, str(device))
    if key not in _colmap_cache:
        host = (C.c_int * (2 * co))()
        _lib.lib().sininn_coupling_colmap(co, coupling_tile(co), host)
        _colmap_cache[key] = torch.tensor(list(host), dtype=torch.int32, device=device)
    return _colmap_cache[key]


def pad32(n):
    return (n + 31) // 32 * 32


def alloc_packs(n, cin, k, colmap, want_dgrad, wino_fwd, wino_dgrad, device):
    """Empty (w_fwd, b_fwd, w_dgrad) buffers of a conv with n outputs / cin inputs (layouts: pack_conv)."""
    npk = colmap.numel() if colmap is not None else pad16(n)
    taps = k * k
    b_fwd = torch.empty(npk, device=device, dtype=torch.float32)
    w_fwd = torch.empty((16 if wino_fwd else taps) * npk * cin, device=device, dtype=torch.float32)
    w_dg = None
    if want_dgrad:
        cdp = pad32(cin) if wino_dgrad else pad16(cin)
        w_dg = torch.empty((16 if wino_dgrad else taps) * cdp * n, device=device, dtype=torch.float32)
    return w_fwd, b_fwd, w_dg


def pack_conv(weight, bias, colmap=None, want_dgrad=True, wino_fwd=False, wino_dgrad=False, out=None):
    """OIHW conv weight -> (w_fwd, b_fwd, w_dgrad).
    tap-major packs: w_fwd [taps][Np][Cin], w_dgrad [taps][pad16(Cin)][N];
    Winograd packs (3x3 only): w_fwd [16][Cin/8][Np][8], w_dgrad [16][N/8][pad32(Cin)][8]  (U = G g G^T, chunk-major).
    `out` = a previous result of the same call to refresh in place."""
    _chk(weight)
    n, cin, k, _ = weight.shape
    assert weight.is_contiguous() and (bias is None or bias.is_contiguous())
    assert not (wino_fwd or wino_dgrad) or k == 3
    lib = _lib.lib()
    dev = weight.device
    npk = colmap.numel() if colmap is not None else pad16(n)
    taps = k * k
    cmap = ptr(colmap, dtype=torch.int32)
    if out is not None:
        w_fwd, b_fwd, w_dg = out
    else:
        b_fwd = torch.empty(npk, device=dev, dtype=torch.float32)
        w_fwd = torch.empty((16 if wino_fwd else taps) * npk * cin, device=dev, dtype=torch.float32)
        w_dg = None
        if want_dgrad:
            cdp = pad32(cin) if wino_dgrad else pad16(cin)
            w_dg = torch.empty((16 if wino_dgrad else taps) * cdp * n, device=dev, dtype=torch.float32)
    # tap-major parts (+ the packed bias) in one launch, Winograd parts in another
    check(lib.sininn_pack_conv_weights(ptr(weight), ptr(bias), n, cin, k, cmap, npk,
                                       None if wino_fwd else ptr(w_fwd), ptr(b_fwd),
                                       pad16(cin), None if (wino_dgrad or not want_dgrad) else ptr(w_dg), _stream()))
    if wino_fwd or (want_dgrad and wino_dgrad):
        check(lib.sininn_pack_winograd(ptr(weight), n, cin, cmap, npk, ptr(w_fwd) if wino_fwd else None,
                                       pad32(cin), ptr(w_dg) if (want_dgrad and wino_dgrad) else None, _stream()))
    return w_fwd, b_fwd, w_dg


def pack_conv_bf16(weight, bias, colmap=None, want_dgrad=True, out=None):
    """OIHW conv weight -> bf16 packs for the mixed-precision convs: (wb_fwd [taps][Np][Kp] bf16, b_fwd [Np] fp32,
    wb_dgrad [taps][pad16(Cin)][Kd] bf16); Kp / Kd = Cin / N rounded up to 16."""
    _chk(weight)
    n, cin, k, _ = weight.shape
    assert weight.is_contiguous() and (bias is None or bias.is_contiguous())
    dev = weight.device
    npk = colmap.numel() if colmap is not None else pad16(n)
    taps = k * k
    kp, kd, cdp = pad16(cin), pad16(n), pad16(cin)
    if out is not None:
        w_fwd, b_fwd, w_dg = out
    else:
        b_fwd = torch.empty(npk, device=dev, dtype=torch.float32)
        w_fwd = torch.empty(taps * npk * kp, device=dev, dtype=torch.bfloat16)
        w_dg = torch.empty(taps * cdp * kd, device=dev, dtype=torch.bfloat16) if want_dgrad else None
    check(_lib.lib().sininn_pack_conv_weights_bf16(ptr(weight), ptr(bias), n, cin, k, ptr(colmap, dtype=torch.int32), npk,
                                                   ptr(w_fwd, dtype=torch.bfloat16), ptr(b_fwd), cdp,
                                                   ptr(w_dg, dtype=torch.bfloat16), _stream()))
    return w_fwd, b_fwd, w_dg


def pack_desc(weight, bias, colmap, packs, wino_fwd, wino_dgrad, pad=None):
    """sininn_pack_desc refreshing `packs` (a pack_conv / alloc_packs result) from (weight, bias).  pad = (n_packed, cin_packed,
    gap_begin, gap_len): the packs describe a conv with zero-padded outputs / a zero input-channel gap the weight does not have."""
    n, cin, k, _ = weight.shape
    w_fwd, b_fwd, w_dg = packs
    d = _lib.PackDesc()
    if pad is not None:
        n_packed, cin_packed, gap_begin, gap_len = pad
        assert cin_packed - gap_len == cin and n_packed >= n
        d.src_n, d.gap_begin, d.gap_len = n, gap_begin, gap_len
        n, cin = n_packed, cin_packed
    d.w, d.bias, d.N, d.Cin, d.ksize = ptr(weight), ptr(bias), n, cin, k
    d.colmap = ptr(colmap, dtype=torch.int32)
    d.Np = colmap.numel() if colmap is not None else pad16(n)
    d.w_fwd, d.b_fwd = ptr(w_fwd), ptr(b_fwd)
    d.Cdp = (pad32(cin) if wino_dgrad else pad16(cin))
    d.w_dgrad = ptr(w_dg)
    d.wino_fwd, d.wino_dgrad = int(bool(wino_fwd)), int(bool(wino_dgrad and w_dg is not None))
    return d


def pack_batch(descs, device):
    """Upload a list of PackDesc and return (device table tensor, n, total_work) for pack_batch_run."""
    lib = _lib.lib()
    arr = (_lib.PackDesc * len(descs))(*descs)
    total = 0
    for d in arr:
        d.work_begin = total
        total += lib.sininn_pack_work_items(C.byref(d))
    raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)
    return raw, len(descs), total


def pack_batch_run(table):
    raw, n, total = table
    check(_lib.lib().sininn_pack_batch(C.c_void_p(raw.data_ptr()), n, total, _stream()))


# ---- conv engine ---------------------------------------------------------------------------------
def conv(**kw):
    """Launch sininn_conv; keyword names follow sininn_conv_args (tensors given as (tensor, offset) or c_void_p)."""
    a = ConvArgs()
    for k, v in kw.items():
        setattr(a, 'inp' if k == 'in_' else k, int(v) if isinstance(v, bool) else v)
    check(_lib.lib().sininn_conv(C.byref(a), _stream()))


def wgrad(in_t, in_off, in_stride, cin, dout, dout_stride, n, b, h, w, ksize, gw, gb, dout_off=0):
    """gw (OIHW, contiguous) += dW ; gb += db."""
    lib = _lib.lib()
    nbytes = lib.sininn_wgrad_workspace_bytes(n, cin, ksize, b, h, w)
    ws = torch.empty((nbytes + 3) // 4, device=dout.device, dtype=torch.float32)
    assert gw.is_contiguous() and (gb is None or gb.is_contiguous())
    check(lib.sininn_wgrad(ptr(in_t, in_off), in_stride, cin, ptr(dout, dout_off), dout_stride, n, b, h, w, ksize,
                           ptr(gw), ptr(gb), ptr(ws), nbytes, _stream()))


def wgrad_group(problems, b, h, w, ksize):
    """problems: list of (in_t, in_off, in_stride, cin, dout, dout_off, dout_stride, n, gw, gb[, in_bf16, dout_bf16]): all +=
    in two launches.  Operands flagged bf16 are torch.bfloat16 tensors (strides in elements)."""
    lib = _lib.lib()
    arr = (_lib.WgradItem * len(problems))()
    for it, prob in zip(arr, problems):
        it.struct_bytes = C.sizeof(_lib.WgradItem)       # array elements are zero-filled, not constructed
        in_t, in_off, in_stride, cin, dout, dout_off, dout_stride, n, gw, gb = prob[:10]
        in_b, dout_b = (bool(prob[10]), bool(prob[11])) if len(prob) > 10 else (False, False)
        assert gw.is_contiguous() and (gb is None or gb.is_contiguous())
        it.inp, it.in_stride, it.Cin = ptr(in_t, in_off, torch.bfloat16 if in_b else torch.float32), in_stride, cin
        it.dout, it.dout_stride, it.N = ptr(dout, dout_off, torch.bfloat16 if dout_b else torch.float32), dout_stride, n
        it.gw, it.gb = ptr(gw), ptr(gb)
        it.in_bf16, it.dout_bf16 = int(in_b), int(dout_b)
    nbytes = lib.sininn_wgrad_group_workspace_bytes(arr, len(problems), b, h, w, ksize)
    ws = torch.empty((nbytes + 3) // 4, device=problems[0][0].device, dtype=torch.float32)
    check(lib.sininn_wgrad_group(arr, len(problems), b, h, w, ksize, ptr(ws), nbytes, _stream()))


def coupling_bwd(dy, dy_off, dy_stride, dy_map, vy, vy_off, vy_stride, vy_map, s, gld, b, hw, co, clamp, inverse,
                 dr, dv, dv_off, dv_stride):
    check(_lib.lib().sininn_coupling_bwd(ptr(dy, dy_off), dy_stride, dy_map, ptr(vy, vy_off), vy_stride, vy_map,
                                         ptr(s), ptr(gld), b, hw, co, clamp, inverse, ptr(dr), ptr(dv, dv_off),
                                         dv_stride, _stream()))


# ---- index maps ----------------------------------------------------------------------------------
_inverse_maps = {}


def _inverse_map(m):
    """Inverse permutation of a device int32 map, cached per tensor (the maps are persistent module buffers)."""
    key = (m.data_ptr(), m.numel())
    hit = _inverse_maps.get(key)
    if hit is None or hit[0] is not m:
        inv = torch.empty_like(m)
        inv[m.long()] = torch.arange(m.numel(), dtype=m.dtype, device=m.device)
        hit = _inverse_maps[key] = (m, inv)
    return hit[1]


def _dense_pixel_major(t):
    b, c, h, w = t.shape
    return t.stride() == (h * w * c, 1, w * c, c) or (c == 1 and t.is_contiguous())


def squeeze(x, out, b, c, h, w, levels, inverse, chan_map=None, map_on_out=False):
    """x / out are 4-D (B,C,H,W)-shaped tensors of ANY strides (fine side has C,H,W; coarse side C*4^l,H/2^l,W/2^l)."""
    # fast path: both tensors dense pixel-major and the map (if any) on the fine tensor's channels -> gather-form kernel
    # with 16-byte stores.  Forward reads the fine side (map used as is); inverse writes it (scatter form -> inverse map).
    fine_map_side = (chan_map is None) or levels == 0 or (map_on_out == bool(inverse))
    if (fine_map_side and _dense_pixel_major(x) and _dense_pixel_major(out) and (b * c * h * w) % 4 == 0
            and (c << (2 * levels)) % 4 == 0 and x.data_ptr() % 16 == 0 and out.data_ptr() % 16 == 0):
        gmap = chan_map
        if chan_map is not None and map_on_out:
            gmap = _inverse_map(chan_map)               # out[map[j]] = in[j]  <=>  out[j'] = in[inv[j']]
        check(_lib.lib().sininn_squeeze_rows(ptr(x), ptr(out), b, c, h, w, levels, 1 if inverse else 0,
                                             ptr(gmap, dtype=torch.int32), _stream()))
        return
    check(_lib.lib().sininn_squeeze(ptr(x), strides4(x), ptr(out), strides4(out), b, c, h, w, levels,
                                    1 if inverse else 0, ptr(chan_map, dtype=torch.int32), 1 if map_on_out else 0,
                                    _stream()))


def permute_channels(x2d, out2d, idx):
    m, c = x2d.shape
    check(_lib.lib().sininn_permute_channels(ptr(x2d), x2d.stride(0), ptr(out2d), out2d.stride(0), m, c,
                                             ptr(idx, dtype=torch.int32), _stream()))


# ---- losses --------------------------------------------------------------------------------------
def sqdiff_sum(x, y, out):
    b, c, h, w = x.shape
    check(_lib.lib().sininn_sqdiff_sum(ptr(x), strides4(x), ptr(y), strides4(y) if y is not None else I64x4(),
                                       b, c, h, w, ptr(out), _stream()))


def sqdiff_bwd(x, y, scale, gscale, gx, gy):
    b, c, h, w = x.shape
    z = I64x4()
    check(_lib.lib().sininn_sqdiff_bwd(ptr(x), strides4(x), ptr(y), strides4(y) if y is not None else z, b, c, h, w,
                                       ptr(scale), gscale, ptr(gx), strides4(gx) if gx is not None else z,
                                       ptr(gy), strides4(gy) if gy is not None else z, _stream()))


def mmd_gram(x, y, g):
    b, c, h, w = x.shape
    check(_lib.lib().sininn_mmd_gram(ptr(x), strides4(x), ptr(y), strides4(y), b, c, h, w, ptr(g), _stream()))


def mmd_finish(g, b, rev, out, coef):
    check(_lib.lib().sininn_mmd_finish(ptr(g), b, 1 if rev else 0, ptr(out), ptr(coef), _stream()))


def mmd_bwd(x, y, coef, scale, gx, gy):
    b, c, h, w = x.shape
    z = I64x4()
    check(_lib.lib().sininn_mmd_bwd(ptr(x), strides4(x), ptr(y), strides4(y), b, c, h, w, ptr(coef), ptr(scale),
                                    ptr(gx), strides4(gx) if gx is not None else z,
                                    ptr(gy), strides4(gy) if gy is not None else z, _stream()))


# ---- warps ---------------------------------------------------------------------------------------
def affine_warp(img, theta, out, ref=None, sse=None):
    b, c, h, w = img.shape
    assert theta.shape == (b, 2, 3) and theta.is_contiguous()
    check(_lib.lib().sininn_affine_warp(ptr(img), strides4(img), ptr(theta), b, c, h, w, ptr(out), strides4(out),
                                        ptr(ref), strides4(ref) if ref is not None else I64x4(), ptr(sse), _stream()))


def affine_warp_bwd(gout, theta, gimg):
    b, c, h, w = gout.shape
    check(_lib.lib().sininn_affine_warp_bwd(ptr(gout), strides4(gout), ptr(theta), b, c, h, w, ptr(gimg),
                                            strides4(gimg), _stream()))


def flow_warp_l1(img, flow, target, warped, metric):
    """img / target / warped: all fp32 or all bf16 (BASELINE configs[3] arithmetic); flow, metric fp32."""
    b, c, h, w = img.shape
    for t in (img, flow, target, warped, metric):
        assert t is None or t.is_contiguous()
    if img.dtype == torch.bfloat16:
        bf = torch.bfloat16
        assert all(t is None or t.dtype == bf for t in (target, warped)), 'bf16 flow warp: img, target, warped must all be bf16'
        check(_lib.lib().sininn_flow_warp_l1_bf16(ptr(img, dtype=bf), ptr(flow), ptr(target, dtype=bf), b, c, h, w,
                                                  ptr(warped, dtype=bf), ptr(metric), _stream()))
        return
    check(_lib.lib().sininn_flow_warp_l1(ptr(img), ptr(flow), ptr(target), b, c, h, w, ptr(warped), ptr(metric),
                                         _stream()))


def flow_warp_l1_bwd(img, flow, target, warped, gwarped, gmetric, gimg, gflow):
    """image-like operands (img, target, warped, gwarped) fp32 or bf16 like the forward; gimg / gflow / gmetric fp32."""
    b, c, h, w = img.shape
    for t in (img, flow, target, warped, gwarped, gmetric, gimg, gflow):
        assert t is None or t.is_contiguous()
    if img.dtype == torch.bfloat16:
        bf = torch.bfloat16
        assert all(t is None or t.dtype == bf for t in (target, warped, gwarped))
        check(_lib.lib().sininn_flow_warp_l1_bwd_bf16(ptr(img, dtype=bf), ptr(flow), ptr(target, dtype=bf), ptr(warped, dtype=bf),
                                                      ptr(gwarped, dtype=bf), ptr(gmetric), b, c, h, w, ptr(gimg), ptr(gflow),
                                                      _stream()))
        return
    check(_lib.lib().sininn_flow_warp_l1_bwd(ptr(img), ptr(flow), ptr(target), ptr(warped), ptr(gwarped),
                                             ptr(gmetric), b, c, h, w, ptr(gimg), ptr(gflow), _stream()))


# ---- sampler / optimiser -------------------------------------------------------------------------
def sample_windows(hr_clip, lr_clip, idx, win, hr_out, lr_out):
    t, hh, ww, _ = hr_clip.shape
    _, h, w, _ = lr_clip.shape
    assert hr_clip.is_contiguous() and lr_clip.is_contiguous() and idx.dtype == torch.int32
    check(_lib.lib().sininn_sample_windows(ptr(hr_clip, dtype=torch.uint8), ptr(lr_clip, dtype=torch.uint8),
                                           ptr(idx, dtype=torch.int32), idx.numel(), t, hh, ww, h, w, win,
                                           ptr(hr_out), strides4(hr_out), ptr(lr_out), strides4(lr_out), _stream()))


def adam_step(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
    for t in (p, g, m, v):
        assert t.is_contiguous() and t.numel() == p.numel()
    check(_lib.lib().sininn_adam_step(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), lr, beta1, beta2, eps, weight_decay,
                                      step, grad_scale, _stream()))
