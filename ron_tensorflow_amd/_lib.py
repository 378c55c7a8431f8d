"""ctypes binding of libron_hip.so (the C ABI declared in include/ron_hip.h).

There is no CPU fallback: if the library is missing or a call fails this module raises.
torch tensors are used only as device containers; every call receives raw pointers.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# tooling (tools/sweep_conv.py --exp / --diag) may point RON_HIP_LIB at the experimental / diagnostic build of the same ABI
LIB_PATH = os.environ.get('RON_HIP_LIB') or os.path.join(_HERE, 'libron_hip.so')

# the ABI this binding was written against (include/ron_hip.h, ron_abi_version): a stale libron_hip.so (git-ignored, but shipped by
# gpurun) with a shorter ron_conv_desc would silently ignore the new fields
EXPECTED_ABI = 2

RON_MAX_LAYERS = 8
RON_MAX_TOPK = 512

RON_IN_CLS_IS_PROB = 1
RON_IN_OBJ_IS_PROB = 2
RON_IN_LOC_DECODED = 4
RON_CFG_FUSE_POOLS = 1
RON_CFG_MULTI_STREAM = 2
RON_CFG_NO_STEM2 = 4
RON_CFG_NO_GROUPS = 8
RON_CFG_NO_HALO_SKIP = 16
RON_CFG_LEVEL_GROUPS = 32
RON_CFG_BATCH_GROUPS = 64

DTYPES = {'fp32': 0, 'f32': 0, 'float32': 0, 'bf16': 1, 'bfloat16': 1, 'fp16': 2, 'f16': 2, 'float16': 2,
          'f16x3': 3, 'fp16x3': 3}     # f16x3: split precision (two f16 planes per value, 3 MFMAs per product; include/ron_hip.h)
VARIANTS = {'reducedfc': 0, 'full': 1, 'ssd512': 2}


class RonError(RuntimeError):
    pass


class Heads(C.Structure):
    _fields_ = [('num_layers', C.c_int32), ('num_classes', C.c_int32),
                ('feat_h', C.c_int32 * RON_MAX_LAYERS), ('feat_w', C.c_int32 * RON_MAX_LAYERS),
                ('num_anchors', C.c_int32 * RON_MAX_LAYERS),
                ('cls', C.c_void_p * RON_MAX_LAYERS), ('obj', C.c_void_p * RON_MAX_LAYERS),
                ('loc', C.c_void_p * RON_MAX_LAYERS),
                ('anchor_y', C.c_void_p * RON_MAX_LAYERS), ('anchor_x', C.c_void_p * RON_MAX_LAYERS),
                ('anchor_h', C.c_void_p * RON_MAX_LAYERS), ('anchor_w', C.c_void_p * RON_MAX_LAYERS)]


class PostCfg(C.Structure):
    _fields_ = [('objectness_thres', C.c_float), ('select_threshold', C.c_float), ('nms_threshold', C.c_float),
                ('top_k', C.c_int32), ('bbox_img', C.c_float * 4), ('prior_scaling', C.c_float * 4),
                ('input_flags', C.c_uint32)]


class EvalCfg(C.Structure):
    _fields_ = [('objectness_thres', C.c_float), ('select_threshold', C.c_float), ('nms_threshold', C.c_float),
                ('keep_top_k', C.c_int32), ('nms_mode', C.c_int32), ('bbox_img', C.c_float * 4),
                ('prior_scaling', C.c_float * 4), ('input_flags', C.c_uint32)]


class TfeCfg(C.Structure):
    _fields_ = [('objectness_thres', C.c_float), ('select_threshold', C.c_float), ('nms_threshold', C.c_float),
                ('top_k', C.c_int32), ('keep_top_k', C.c_int32), ('nms_mode', C.c_int32), ('clip', C.c_int32),
                ('clipping_bbox', C.c_float * 4), ('min_size', C.c_float), ('prior_scaling', C.c_float * 4),
                ('input_flags', C.c_uint32)]


class Detections(C.Structure):
    _fields_ = [('capacity', C.c_int32), ('classes', C.c_void_p), ('scores', C.c_void_p), ('bboxes', C.c_void_p),
                ('anchor_index', C.c_void_p), ('count', C.c_void_p)]


class Config(C.Structure):
    _fields_ = [('variant', C.c_int32), ('dtype', C.c_int32), ('img_h', C.c_int32), ('img_w', C.c_int32),
                ('num_classes', C.c_int32), ('max_batch', C.c_int32), ('device', C.c_int32), ('flags', C.c_uint32)]


class ConvDesc(C.Structure):
    _fields_ = [('n', C.c_int32), ('h', C.c_int32), ('w', C.c_int32), ('cin', C.c_int32), ('cout', C.c_int32),
                ('kh', C.c_int32), ('kw', C.c_int32), ('stride', C.c_int32), ('dilation', C.c_int32),
                ('relu', C.c_int32), ('transpose', C.c_int32), ('dtype', C.c_int32), ('tile_cfg', C.c_int32),
                ('in_cstride', C.c_int32), ('in_coff', C.c_int32), ('pool', C.c_int32), ('splitk', C.c_int32),
                ('center_from', C.c_int32)]


# every symbol include/ron_hip.h declares: (restype, argtypes)
_P = C.c_void_p
SIGNATURES = {
    'ron_last_error': (C.c_char_p, []),
    'ron_abi_version': (C.c_int, []),
    'ron_crc32c': (C.c_uint32, [_P, C.c_uint64, C.c_uint32]),
    'ron_anchor_one_layer': (C.c_int, [C.c_int] * 4 + [C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double), C.c_int,
                                                       C.c_double, C.c_double, _P, _P, _P, _P]),
    'ron_ssd_anchor_one_layer': (C.c_int, [C.c_int] * 4 + [C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double), C.c_int,
                                                           C.c_double, C.c_double, _P, _P, _P, _P]),
    'ron_post_np_workspace_bytes': (C.c_int64, [C.POINTER(Heads), C.c_int]),
    'ron_post_np': (C.c_int, [C.POINTER(Heads), C.c_int, C.POINTER(PostCfg), _P, C.c_int64, C.POINTER(Detections),
                              C.POINTER(Detections), _P, _P]),
    'ron_np_sort_nms': (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_float, _P, C.c_int64,
                                  C.POINTER(Detections), C.POINTER(Detections), _P]),
    'ron_np_sort_nms_workspace_bytes': (C.c_int64, [C.c_int, C.c_int]),
    'ron_bboxes_decode_layer': (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P,
                                          C.POINTER(C.c_float), _P, _P]),
    'ron_softmax_last': (C.c_int, [_P, C.c_int64, C.c_int, C.c_int, _P, _P]),
    'ron_post_eval_workspace_bytes': (C.c_int64, [C.POINTER(Heads), C.c_int]),
    'ron_post_eval_workspace_bytes_mode': (C.c_int64, [C.POINTER(Heads), C.c_int, C.c_int]),
    'ron_bboxes_filter_min': (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_float, _P, _P, C.c_int, _P, _P]),
    'ron_post_eval': (C.c_int, [C.POINTER(Heads), C.c_int, _P, C.POINTER(EvalCfg), _P, C.c_int64, C.POINTER(Detections), _P]),
    'ron_post_tfe_workspace_bytes': (C.c_int64, [C.POINTER(Heads), C.c_int]),
    'ron_preprocess_eval': (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), _P, _P]),
    'ron_preprocess_eval_geom': (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), _P, _P]),
    'ron_pack_records': (C.c_int, [C.POINTER(Detections), C.c_int, _P, _P]),
    'ron_gather_records': (C.c_int, [_P, C.c_int, C.c_int, _P, _P, _P]),
    'ron_bboxes_matching': (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P, _P, C.c_int, C.c_float, _P, _P, _P, _P]),
    'ron_post_tfe': (C.c_int, [C.POINTER(Heads), C.c_int, C.POINTER(TfeCfg), _P, C.c_int64, _P, _P, _P]),
    'ron_create': (C.c_int, [C.POINTER(_P), C.POINTER(Config)]),
    'ron_destroy': (C.c_int, [_P]),
    'ron_clone': (C.c_int, [_P, C.POINTER(_P)]),
    'ron_num_variables': (C.c_int, [_P]),
    'ron_variable_info': (C.c_int, [_P, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
    'ron_load_weight': (C.c_int, [_P, C.c_char_p, _P, C.POINTER(C.c_int64), C.c_int]),
    'ron_finalize_weights': (C.c_int, [_P]),
    'ron_heads_describe': (C.c_int, [_P, C.POINTER(Heads)]),
    'ron_forward': (C.c_int, [_P, _P, C.c_int, C.POINTER(Heads), _P]),
    'ron_end_point_shape': (C.c_int, [_P, C.c_char_p, C.c_int, C.POINTER(C.c_int64)]),
    'ron_end_point_copy': (C.c_int, [_P, C.c_char_p, C.c_int, _P, _P]),
    'ron_detect': (C.c_int, [_P, _P, C.c_int, C.POINTER(PostCfg), C.POINTER(Detections), _P]),
    'ron_flops_per_image': (C.c_double, [_P]),
    'ron_profile_enable': (C.c_int, [_P, C.c_int]),
    'ron_num_grouped_launches': (C.c_int, [_P]),
    'ron_profile_num_ops': (C.c_int, [_P]),
    'ron_profile_get': (C.c_int, [_P, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.POINTER(C.c_double),
                                  C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    'ron_profile_reset': (C.c_int, [_P]),
    'ron_conv2d_nhwc': (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P]),
    'ron_conv_plan': (C.c_int, [C.POINTER(ConvDesc), C.POINTER(C.c_int32)]),
    'ron_conv2d_heads_nhwc': (C.c_int, [C.POINTER(ConvDesc), C.c_int, _P, _P, _P, _P, _P, _P]),
    'ron_maxpool2x2_nhwc': (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    'ron_conv2d_bench': (C.c_int, [C.POINTER(ConvDesc), C.c_int, C.c_int, C.POINTER(C.c_float)]),
    'ron_conv_num_tile_cfgs': (C.c_int, []),
}

_lib = None


def lib():
    """The loaded library.  Raises RonError when it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RonError('%s not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                           '(or `make -C ron_tensorflow_amd/csrc`). There is no CPU fallback.' % LIB_PATH)
        # torch wheels bundle their own libamdhip64 (same soname as the system's): whichever is loaded first serves both.  Loaded
        # after libron_hip.so, torch would bring a second HIP runtime into the process and this library's calls would land in the one
        # that sees no device ("no ROCm-capable device is detected").  torch tensors are the device containers of every call, so
        # torch comes first.
        import torch  # noqa: F401
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            if os.environ.get('RON_HIP_LIB') and not hasattr(handle, name):
                continue                        # A/B tooling against an older build of the library (tools/experiments/libron_hip_r05.so):
                                                # entry points added since are absent there; calling one raises AttributeError
            fn = getattr(handle, name)          # AttributeError if the symbol is missing: fail loudly
            fn.restype = res
            fn.argtypes = args
        have = handle.ron_abi_version()
        if have != EXPECTED_ABI:
            raise RonError('%s has ABI version %d, this package needs %d: rebuild libron_hip.so (`make -C ron_tensorflow_amd/csrc`)'
                           % (LIB_PATH, have, EXPECTED_ABI))
        _lib = handle
    return _lib


def check(rc):
    if rc != 0:
        raise RonError('libron_hip: status %d: %s' % (rc, lib().ron_last_error().decode()))


def ptr(t):
    """Device/host pointer of a contiguous torch tensor or numpy array (None -> NULL)."""
    if t is None:
        return None
    if hasattr(t, 'data_ptr'):
        assert t.is_contiguous()
        return C.c_void_p(t.data_ptr())
    assert t.flags['C_CONTIGUOUS']
    return C.c_void_p(t.ctypes.data)


def current_stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
