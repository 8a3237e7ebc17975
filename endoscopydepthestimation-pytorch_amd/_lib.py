"""ctypes binding of libendo_hip.so (C ABI: include/endo_hip.h).

The product path has no fallback: if the library is missing or a call fails, a RuntimeError is
raised.  Build it with ``python -c 'import __graft_entry__ as g; g.build()'`` (hipcc, gfx950).
"""

import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# ENDO_HIP_LIB points at another build of the same ABI (A/B runs of kernel variants inside one benchmark job)
LIB_PATH = os.environ.get("ENDO_HIP_LIB") or os.path.join(_HERE, "lib", "libendo_hip.so")

_P = ctypes.c_void_p
_I = ctypes.c_int
_F = ctypes.c_float
_L = ctypes.c_int64

# name -> (restype, argtypes); every symbol include/endo_hip.h declares
SIGNATURES = {
    "endo_abi_version": (_I, []),
    "endo_error_string": (ctypes.c_char_p, [_I]),
    "endo_depth_scale_fwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _F, _P]),
    "endo_depth_scale_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _P]),
    "endo_flow_from_depth_fwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "endo_flow_from_depth_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "endo_depth_warp_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _P]),
    "endo_depth_warp_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _P]),
    "endo_depth_warp_fwd_tiled": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _I, _I, _P]),
    "endo_depth_warp_bwd_tiled": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _I, _I, _P]),
    "endo_sparse_l1_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _F, _P]),
    "endo_sparse_l1_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _P]),
    "endo_norm_dist_fwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _P]),
    "endo_norm_dist_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _P]),
    "endo_scale_inv_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _F, _P]),
    "endo_scale_inv_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _P]),
    "endo_loss_head_workspace_floats": (_L, [_I, _I, _I]),
    "endo_loss_head": (_I, [_P] * 16 + [_F, _F, _F] + [_P] * 4 + [_I, _I, _I, _P]),
    "endo_warp_consistency_workspace_floats": (_L, [_I, _I, _I]),
    "endo_warp_consistency_bytes": (_L, [_I, _I, _I]),
    "endo_warp_consistency": (_I, [_P] * 8 + [_F, _F] + [_P] * 4 + [_I, _I, _I, _P]),
    "endo_warp_fallback_blocks": (_I, [_P, _P, _I]),
    "endo_bf16_pack_nhwc": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "endo_f16_pack_nhwc": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "endo_bf16_unpack_nhwc": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "endo_f16_unpack_nhwc": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "endo_bf16_conv_weight_elems": (_L, [_I, _I, _I]),
    "endo_bf16_conv_weights": (_I, [_P, _I, _I, _I, _P, _P]),
    "endo_bf16_conv": (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _P, _I, _I, _I, _I, _P, _I, _I, _I, _I, _I, _P]),
    "endo_net16_create": (_I, [ctypes.POINTER(_P), _I, _I, _I, _I]),
    "endo_net16h_create": (_I, [ctypes.POINTER(_P), _I, _I, _I, _I]),
    "endo_net16_destroy": (None, [_P]),
    "endo_net16h_destroy": (None, [_P]),
    "endo_net16_tape_bytes": (_L, [_P]),
    "endo_net16h_tape_bytes": (_L, [_P]),
    "endo_net16_fwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _P]),
    "endo_net16h_fwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _P]),
    "endo_net16_bwd_workspace_bytes": (_L, [_P]),
    "endo_net16h_bwd_workspace_bytes": (_L, [_P]),
    "endo_net16_set_wgrad_overlap": (_I, [_P, _I]),
    "endo_net16h_set_wgrad_overlap": (_I, [_P, _I]),
    "endo_net16_offset": (_L, [_P, _I, _I]),
    "endo_net16h_offset": (_L, [_P, _I, _I]),
    "endo_net16_bwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _P]),
    "endo_net16h_bwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _P]),
    "endo_mask_mul": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "endo_net_create": (_I, [ctypes.POINTER(_P), _I, _I, _I]),
    "endo_net_create_grouped": (_I, [ctypes.POINTER(_P), _I, _I, _I, _I]),
    "endo_net_groups": (_I, [_P]),
    "endo_net_set_option": (_I, [_P, _I, _I]),
    "endo_net_get_option": (_I, [_P, _I]),
    "endo_net_group_stride": (_L, [_P]),
    "endo_net_destroy": (None, [_P]),
    "endo_net_param_floats": (_L, []),
    "endo_net_bn_floats": (_L, []),
    "endo_net_tape_floats": (_L, [_P]),
    "endo_net_gradws_floats": (_L, [_P]),
    "endo_net_param_offset": (_L, [_I]),
    "endo_net_bn_offset": (_L, [_I, _I]),
    "endo_net_level_channels": (_I, [_I]),
    "endo_net_act_offset": (_L, [_P, _I]),
    "endo_net_tape_offset": (_L, [_P, _I, _I]),
    "endo_net_fwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _P]),
    "endo_net_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _P]),
    "endo_sgd_clip_step": (_I, [_P, _P, _P, _P, _L, _F, _F, _F, _F, _I, _P, _P]),
    "endo_sparse_scatter": (_I, [_P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _F, _P, _P, _P, _P, _P, _P]),
    "endo_relative_poses": (_I, [_P, _I, ctypes.c_double, _P, _P, _P, _P, _P]),
    "endo_point_cloud": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _F, _F, _P, _P, _P, _P]),
    "endo_jpeg_info": (_I, [_P, _L, _P]),
    "endo_jpeg_entropy_decode": (_I, [_P, _L, _P, _L, _P]),
    "endo_jpeg_workspace_bytes": (_L, [_P, _L]),
    "endo_jpeg_decode_crop": (_I, [_P, _L, ctypes.c_double, _I, _I, _I, _I, _I, _P, _P, _P, _P, _L, _P]),
    "endo_hsv_full": (_I, [_P, _L, _I, _P, _P, _P]),
    "endo_point_brightness": (_I, [_P, _I, _I, _I, _P, _I, _P, _P, _P, _P, _I, ctypes.c_double, ctypes.c_double, _P, _P, _P, _P]),
    "endo_prof_enable": (_I, [_I]),
    "endo_prof_sample": (_I, [_I]),
    "endo_prof_seen": (_I, [_I, ctypes.POINTER(_L)]),
    "endo_prof_read": (_I, [_I, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_L),
                            ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "endo_prof_family_name": (ctypes.c_char_p, [_I]),
}

_lib = None


def load():
    """Load the shared library (once).  Raises RuntimeError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libendo_hip.so not found at %s -- the HIP extension is required (no CPU fallback); "
                "build it with __graft_entry__.build()" % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)      # AttributeError = ABI mismatch, fail loudly
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(code, what):
    if code != 0:
        msg = load().endo_error_string(int(code))
        raise RuntimeError("%s failed: %s (code %d)" % (what, msg.decode() if msg else "?", code))


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def dev_f32(t, name):
    """Validate the boundary contract: contiguous fp32 on a HIP device."""
    if not t.is_cuda:
        raise RuntimeError("%s must live on the GPU: the MI355X path has no CPU fallback" % name)
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()
