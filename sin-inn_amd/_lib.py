"""ctypes binding of libsininn.so (the C ABI declared in include/sininn.h).

The product path has NO CPU fallback: if the shared library is missing, or a tensor is not a
contiguous-enough fp32 CUDA(HIP) tensor, the call raises.  (Same convention as the reference's raw
pointer kernel call site, video-interpolation/my_utils/softsplat.py:239-331: asserts +
NotImplementedError for CPU tensors.)
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('SININN_LIB', os.path.join(_HERE, 'libsininn.so'))   # override: A/B two builds on one box

c_f = C.c_void_p      # device float*
c_i = C.c_void_p      # device int*
I64x4 = C.c_int64 * 4


class ConvArgs(C.Structure):
    """Mirror of sininn_conv_args (include/sininn.h)."""
    _fields_ = [
        ('inp', c_f), ('in_stride', C.c_int), ('Cin', C.c_int),
        ('w', c_f), ('bias', c_f), ('Np', C.c_int), ('winograd', C.c_int),
        ('B', C.c_int), ('H', C.c_int), ('W', C.c_int), ('ksize', C.c_int),
        ('mode', C.c_int),
        ('out', c_f), ('out_stride', C.c_int), ('N', C.c_int),
        ('out_map', c_i),
        ('v', c_f), ('v_stride', C.c_int),
        ('out2', c_f), ('out2_stride', C.c_int),
        ('sbuf', c_f),
        ('logdet', c_f),
        ('Co', C.c_int), ('clamp', C.c_float), ('col_tile', C.c_int),
        ('mask', c_f), ('mask_stride', C.c_int),
        ('addend', c_f), ('addend_stride', C.c_int), ('addend_map', c_i),
        ('stamp', C.c_void_p),
        ('w_bf16', C.c_int), ('in_bf16', C.c_int), ('out_bf16', C.c_int), ('mask_bf16', C.c_int), ('in_group_stride', C.c_int),
        ('out_group_stride', C.c_int), ('mask_group_stride', C.c_int),
    ]


class SubnetArgs(C.Structure):
    """Mirror of sininn_subnet."""
    _fields_ = [('w1', c_f), ('b1', c_f), ('w2', c_f), ('b2', c_f), ('w1_dgrad', c_f), ('w2_dgrad', c_f),
                ('gw1', c_f), ('gb1', c_f), ('gw2', c_f), ('gb2', c_f), ('winograd', C.c_int)]


class GlowArgs(C.Structure):
    """Mirror of sininn_glow_args."""
    _fields_ = [('B', C.c_int), ('H', C.c_int), ('W', C.c_int), ('C', C.c_int), ('ksize', C.c_int), ('rev', C.c_int),
                ('clamp', C.c_float),
                ('x', c_f), ('out', c_f), ('dst_map', c_i), ('logdet', c_f),
                ('s1', SubnetArgs), ('s2', SubnetArgs),
                ('saved', c_f), ('scratch', C.c_void_p), ('scratch_bytes', C.c_size_t),
                ('dout', c_f), ('gld', c_f), ('dx', c_f), ('skip_dx', C.c_int), ('dtype', C.c_int), ('no_save', C.c_int)]


class WgradItem(C.Structure):
    """Mirror of sininn_wgrad_item."""
    _fields_ = [('struct_bytes', C.c_size_t),
                ('inp', c_f), ('in_stride', C.c_int), ('Cin', C.c_int), ('dout', c_f), ('dout_stride', C.c_int),
                ('N', C.c_int), ('gw', c_f), ('gb', c_f), ('in_bf16', C.c_int), ('dout_bf16', C.c_int),
                ('in_group_stride', C.c_int), ('dout_group_stride', C.c_int), ('gap_begin', C.c_int), ('gap_len', C.c_int)]

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.struct_bytes = C.sizeof(WgradItem)


class DenseArgs(C.Structure):
    """Mirror of sininn_dense_args."""
    _fields_ = [('struct_bytes', C.c_size_t),
                ('B', C.c_int), ('H', C.c_int), ('W', C.c_int), ('cin', C.c_int), ('cout', C.c_int), ('mode', C.c_int),
                ('winograd', C.c_int), ('clamp', C.c_float),
                ('x', c_f), ('x_stride', C.c_int), ('aux1', c_f), ('aux1_stride', C.c_int), ('aux2', c_f),
                ('buf', c_f), ('out', c_f),
                ('w_fwd', c_f * 5), ('b_fwd', c_f * 5), ('w_dgrad', c_f * 5),
                ('dout', c_f), ('dF', c_f), ('dD', c_f), ('dh', c_f), ('dv', c_f),
                ('gw', c_f * 5), ('gb', c_f * 5),
                ('workspace', C.c_void_p), ('workspace_bytes', C.c_size_t),
                ('buf_floats', C.c_size_t), ('out_floats', C.c_size_t), ('aux2_floats', C.c_size_t),
                ('dout_floats', C.c_size_t), ('dF_floats', C.c_size_t), ('dD_floats', C.c_size_t),
                ('dh_floats', C.c_size_t), ('dv_floats', C.c_size_t)]

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.struct_bytes = C.sizeof(DenseArgs)


class PackDesc(C.Structure):
    """Mirror of sininn_pack_desc."""
    _fields_ = [('w', c_f), ('bias', c_f), ('N', C.c_int), ('Cin', C.c_int), ('ksize', C.c_int), ('colmap', c_i),
                ('Np', C.c_int), ('w_fwd', c_f), ('b_fwd', c_f), ('Cdp', C.c_int), ('w_dgrad', c_f),
                ('wino_fwd', C.c_int), ('wino_dgrad', C.c_int), ('work_begin', C.c_int),
                ('src_n', C.c_int), ('gap_begin', C.c_int), ('gap_len', C.c_int)]


CONV_RELU, CONV_COUPLE_FWD, CONV_COUPLE_INV, CONV_MASK, CONV_ADD, CONV_LINEAR = range(6)

_SIGS = {
    'sininn_version': (C.c_int, []),
    'sininn_last_error': (C.c_char_p, []),
    'sininn_sizeof': (C.c_size_t, [C.c_int]),
    'sininn_stream_priority_range': (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    'sininn_stream_create': (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    'sininn_pack_conv_weights': (C.c_int, [c_f, c_f, C.c_int, C.c_int, C.c_int, c_i, C.c_int, c_f, c_f, C.c_int, c_f, C.c_void_p]),
    'sininn_pack_conv_weights_bf16': (C.c_int, [c_f, c_f, C.c_int, C.c_int, C.c_int, c_i, C.c_int, C.c_void_p, c_f, C.c_int,
                                               C.c_void_p, C.c_void_p]),
    'sininn_pack_winograd': (C.c_int, [c_f, C.c_int, C.c_int, c_i, C.c_int, c_f, C.c_int, c_f, C.c_void_p]),
    'sininn_pack_work_items': (C.c_int, [C.POINTER(PackDesc)]),
    'sininn_pack_batch': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    'sininn_softsplat': (C.c_int, [c_f, c_f, C.c_int, C.c_int, C.c_int, C.c_int, c_f, C.c_void_p]),
    'sininn_softsplat_bwd': (C.c_int, [c_f, c_f, c_f, C.c_int, C.c_int, C.c_int, C.c_int, c_f, c_f, C.c_void_p]),
    'sininn_occlusion_wang': (C.c_int, [c_f, C.c_int, C.c_int, C.c_int, C.c_float, c_f, c_f, C.c_void_p]),
    'sininn_census': (C.c_int, [c_f, c_f, c_f, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, c_f, c_f, C.c_void_p]),
    'sininn_census_bwd': (C.c_int, [c_f, c_f, c_f, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, c_f, c_f, c_f, c_f, C.c_void_p]),
    'sininn_masked_l1': (C.c_int, [c_f, c_f, c_f, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, c_f, c_f, C.c_void_p]),
    'sininn_masked_l1_bwd': (C.c_int, [c_f, c_f, c_f, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, c_f, c_f, c_f, c_f, C.c_void_p]),
    'sininn_occlusion_brox': (C.c_int, [c_f, c_f, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'sininn_ssim': (C.c_int, [c_f, c_f, c_f, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, c_f, c_f, C.c_void_p]),
    'sininn_ssim_bwd': (C.c_int, [c_f, c_f, c_f, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, c_f, c_f, c_f, c_f, C.c_void_p]),
    'sininn_bilateral_smooth': (C.c_int, [c_f, c_f, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, c_f, c_f, C.c_void_p]),
    'sininn_bilateral_smooth_bwd': (C.c_int, [c_f, c_f, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, c_f, c_f, C.c_void_p]),
    'sininn_coupling_colmap': (None, [C.c_int, C.c_int, C.POINTER(C.c_int)]),
    'sininn_conv': (C.c_int, [C.POINTER(ConvArgs), C.c_void_p]),
    'sininn_conv_test_hooks': (None, [C.c_int, C.c_int]),
    'sininn_wgrad_test_hooks': (None, [C.c_int]),
    'sininn_pair_k1_test_hook': (None, [C.c_int]),
    'sininn_sub1_bwd_test_hook': (None, [C.c_int]),
    'sininn_wgrad_workspace_bytes': (C.c_size_t, [C.c_int] * 6),
    'sininn_wgrad': (C.c_int, [c_f, C.c_int, C.c_int, c_f, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                               c_f, c_f, C.c_void_p, C.c_size_t, C.c_void_p]),
    'sininn_wgrad_group_workspace_bytes': (C.c_size_t, [C.POINTER(WgradItem), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    'sininn_wgrad_group': (C.c_int, [C.POINTER(WgradItem), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                     C.c_size_t, C.c_void_p]),
    'sininn_coupling_bwd': (C.c_int, [c_f, C.c_int, c_i, c_f, C.c_int, c_i, c_f, c_f, C.c_int, C.c_int, C.c_int,
                                      C.c_float, C.c_int, c_f, c_f, C.c_int, C.c_void_p]),
    'sininn_profile_begin': (None, [C.c_int, C.c_void_p, C.c_int]),
    'sininn_wall_clock_khz': (C.c_int, []),
    'sininn_profile_end': (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_float)]),
    'sininn_profile_classes_begin': (None, []),
    'sininn_profile_classes_end': (C.c_int, [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    'sininn_profile_classes_bytes': (C.c_int, [C.c_int, C.POINTER(C.c_double)]),
    'sininn_glow_saved_floats': (C.c_size_t, [C.c_int] * 4),
    'sininn_glow_saved_floats_dtype': (C.c_size_t, [C.c_int] * 5),
    'sininn_glow_scratch_bytes': (C.c_size_t, [C.c_int] * 5),
    'sininn_glow_scratch_bytes_dtype': (C.c_size_t, [C.c_int] * 6),
    'sininn_glow_forward': (C.c_int, [C.POINTER(GlowArgs), C.c_void_p]),
    'sininn_glow_backward': (C.c_int, [C.POINTER(GlowArgs), C.c_void_p, C.c_void_p]),
    'sininn_glow_hidden_gates': (C.c_int, [C.POINTER(GlowArgs), C.c_int, C.c_void_p, C.c_void_p]),
    'sininn_glow_group_major_fits': (C.c_int, [C.c_int, C.c_int, C.c_int]),
    'sininn_capture_unjoined': (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int)]),
    'sininn_conv_pair_k1_supported': (C.c_int, [C.POINTER(ConvArgs), C.POINTER(ConvArgs)]),
    'sininn_conv_sub3_supported': (C.c_int, [C.POINTER(ConvArgs), C.POINTER(ConvArgs)]),
    'sininn_conv_sub3': (C.c_int, [C.POINTER(ConvArgs), C.POINTER(ConvArgs), C.c_void_p]),
    'sininn_conv_pair_k1': (C.c_int, [C.POINTER(ConvArgs), C.POINTER(ConvArgs), C.c_void_p]),
    'sininn_conv_sub1_fwd_supported': (C.c_int, [C.POINTER(ConvArgs), C.POINTER(ConvArgs)]),
    'sininn_conv_sub1_fwd': (C.c_int, [C.POINTER(ConvArgs), C.POINTER(ConvArgs), C.c_void_p]),
    'sininn_conv_sub1_bwd_workspace_bytes': (C.c_size_t, [C.c_int, C.c_int]),
    'sininn_conv_sub1_bwd': (C.c_int, [C.POINTER(ConvArgs), C.POINTER(ConvArgs), C.POINTER(ConvArgs), C.c_int, c_f, c_f, c_f, c_f,
                                       C.c_void_p, C.c_size_t, C.c_void_p]),
    'sininn_dense_workspace_bytes': (C.c_size_t, [C.c_int] * 5),
    'sininn_dense_forward': (C.c_int, [C.POINTER(DenseArgs), C.c_void_p]),
    'sininn_dense_backward': (C.c_int, [C.POINTER(DenseArgs), C.c_void_p, C.c_void_p]),
    'sininn_haar': (C.c_int, [c_f, I64x4, c_f, I64x4, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'sininn_lrelu_bwd': (C.c_int, [c_f, C.c_int, c_f, C.c_int, C.c_int64, C.c_int, C.c_float, C.c_void_p]),
    'sininn_irn_tail': (C.c_int, [c_f, C.c_int, c_f, c_f, C.c_int64, C.c_int, C.c_float, C.c_int, c_f, C.c_int, C.c_void_p]),
    'sininn_irn_coupling_bwd': (C.c_int, [c_f, C.c_int, c_f, C.c_int, c_f, C.c_int64, C.c_int, C.c_float, C.c_int,
                                          c_f, c_f, c_f, C.c_int, C.c_void_p]),
    'sininn_squeeze': (C.c_int, [c_f, I64x4, c_f, I64x4, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_i,
                                 C.c_int, C.c_void_p]),
    'sininn_squeeze_rows': (C.c_int, [c_f, c_f, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_i, C.c_void_p]),
    'sininn_permute_channels': (C.c_int, [c_f, C.c_int, c_f, C.c_int, C.c_int64, C.c_int, c_i, C.c_void_p]),
    'sininn_sqdiff_sum': (C.c_int, [c_f, I64x4, c_f, I64x4, C.c_int, C.c_int, C.c_int, C.c_int, c_f, C.c_void_p]),
    'sininn_sqdiff_bwd': (C.c_int, [c_f, I64x4, c_f, I64x4, C.c_int, C.c_int, C.c_int, C.c_int, c_f, C.c_float,
                                    c_f, I64x4, c_f, I64x4, C.c_void_p]),
    'sininn_mmd_gram': (C.c_int, [c_f, I64x4, c_f, I64x4, C.c_int, C.c_int, C.c_int, C.c_int, c_f, C.c_void_p]),
    'sininn_mmd_finish': (C.c_int, [c_f, C.c_int, C.c_int, c_f, c_f, C.c_void_p]),
    'sininn_mmd_bwd': (C.c_int, [c_f, I64x4, c_f, I64x4, C.c_int, C.c_int, C.c_int, C.c_int, c_f, c_f,
                                 c_f, I64x4, c_f, I64x4, C.c_void_p]),
    'sininn_affine_warp': (C.c_int, [c_f, I64x4, c_f, C.c_int, C.c_int, C.c_int, C.c_int, c_f, I64x4, c_f, I64x4,
                                     c_f, C.c_void_p]),
    'sininn_affine_warp_bwd': (C.c_int, [c_f, I64x4, c_f, C.c_int, C.c_int, C.c_int, C.c_int, c_f, I64x4, C.c_void_p]),
    'sininn_flow_warp_l1': (C.c_int, [c_f, c_f, c_f, C.c_int, C.c_int, C.c_int, C.c_int, c_f, c_f, C.c_void_p]),
    'sininn_flow_warp_l1_bwd': (C.c_int, [c_f, c_f, c_f, c_f, c_f, c_f, C.c_int, C.c_int, C.c_int, C.c_int, c_f, c_f,
                                          C.c_void_p]),
    'sininn_flow_warp_l1_bf16': (C.c_int, [C.c_void_p, c_f, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, c_f,
                                           C.c_void_p]),
    'sininn_flow_warp_l1_bwd_bf16': (C.c_int, [C.c_void_p, c_f, C.c_void_p, C.c_void_p, C.c_void_p, c_f, C.c_int, C.c_int, C.c_int,
                                               C.c_int, c_f, c_f, C.c_void_p]),
    'sininn_sample_windows': (C.c_int, [C.c_void_p, C.c_void_p, c_i, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_int, C.c_int, c_f, I64x4, c_f, I64x4, C.c_void_p]),
    'sininn_sample_pairs': (C.c_int, [C.c_void_p, c_i, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                      C.c_void_p]),
    'sininn_bayer_bin': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'sininn_bayer_demosaic': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'sininn_frames_to_u8': (C.c_int, [c_f, I64x4, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'sininn_adam_step': (C.c_int, [c_f, c_f, c_f, c_f, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float,
                                   C.c_float, C.c_int, C.c_float, C.c_void_p]),
}

EXPORTED = tuple(_SIGS)
_lib = None


def lib():
    """Load libsininn.so (once).  Raises if the HIP extension has not been built: no fallback."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise ImportError(f'{LIB_PATH} is missing: build it with `make -C sin-inn_amd/csrc` '
                              '(or __graft_entry__.build()); there is no CPU fallback for the HIP path')
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(handle, name)          # AttributeError if the ABI lost a symbol
            fn.restype, fn.argtypes = res, args
        if handle.sininn_version() != 4:
            raise ImportError('libsininn.so ABI version mismatch')
        for which, mirror in enumerate((ConvArgs, WgradItem, DenseArgs, GlowArgs, SubnetArgs, PackDesc)):
            if handle.sininn_sizeof(which) != C.sizeof(mirror):
                raise ImportError(f'{mirror.__name__}: the ctypes mirror has {C.sizeof(mirror)} bytes, libsininn.so was built '
                                  f'with {handle.sininn_sizeof(which)} (include/sininn.h changed without _lib.py)')
        _lib = handle
    return _lib


def check(rc):
    if rc != 0:
        raise RuntimeError('libsininn: ' + lib().sininn_last_error().decode())
