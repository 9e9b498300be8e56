"""ctypes binding of libsrk.so (the C ABI declared in include/srk.h).

There is deliberately NO fallback: if the library is missing or a call fails, a RuntimeError is
raised.  torch is used only for device memory, streams and autograd bookkeeping; every number on
the hot path is produced by a kernel in csrc/.
"""
import ctypes
import os
import re

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SRK_LIB_PATH") or os.path.join(_HERE, "libsrk.so")  # env override: A/B kernel builds
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "srk.h")

# enums mirrored from include/srk.h
ACT_NONE, ACT_RELU, ACT_PRELU, ACT_LRELU, ACT_TANH, ACT_SIGMOID = range(6)
ALGO_AUTO, ALGO_GENERIC, ALGO_MFMA, ALGO_DIRECT, ALGO_MFMA_BF16X3, ALGO_MFMA_BF16X6, ALGO_MFMA_F16X3 = range(7)
AMAX_FLOATS = 256   # SRK_AMAX_FLOATS: 16 slots, one per 64-byte line
LOSS_MSE, LOSS_L1, LOSS_CHARBONNIER, LOSS_BCE = range(4)
ACT_BY_NAME = {None: ACT_NONE, "relu": ACT_RELU, "prelu": ACT_PRELU, "lrelu": ACT_LRELU, "tanh": ACT_TANH,
               "sigmoid": ACT_SIGMOID}

c_f = ctypes.c_void_p  # device float*
c_vp = ctypes.c_void_p
c_int = ctypes.c_int
c_float = ctypes.c_float
c_size = ctypes.c_size_t


class ConvDesc(ctypes.Structure):
    """struct srk_conv_desc"""
    _fields_ = [(n, ctypes.c_int32) for n in
                ("N", "H", "W", "Cin", "OH", "OW", "Cout", "KH", "KW", "stride", "pad", "transposed", "out_pad",
                 "algo", "x_nchw", "dy_ps_r")]


class Epilogue(ctypes.Structure):
    """struct srk_epilogue"""
    _fields_ = [("bias", c_vp), ("prelu_weight", c_vp), ("residual", c_vp), ("slope", c_float),
                ("act", ctypes.c_int32), ("prelu_n", ctypes.c_int32), ("ps_r", ctypes.c_int32),
                ("x_amax", c_vp), ("y_amax", c_vp), ("bn_partial", c_vp)]


class ConvResult(ctypes.Structure):
    """struct srk_conv_result (host out-struct of srk_conv2d_forward_ex; struct_size is the caller's sizeof)"""
    _fields_ = [("struct_size", ctypes.c_uint32), ("wrote_amax", ctypes.c_int32), ("bn_partial_rows", ctypes.c_int32)]

    def __init__(self):
        super().__init__(ctypes.sizeof(ConvResult), 0, 0)


class BwdMask(ctypes.Structure):
    """struct srk_bwd_mask"""
    _fields_ = [("y", c_vp), ("slope", c_float)]


_PROTOTYPES = {
    "srk_version": (c_int, []),
    "srk_status_string": (ctypes.c_char_p, [c_int]),
    "srk_last_error_string": (ctypes.c_char_p, []),
    "srk_last_kernel_name": (ctypes.c_char_p, []),
    "srk_last_conv_wrote_amax": (c_int, []),
    "srk_last_conv_bn_partial_rows": (c_int, []),
    "srk_ring_timeouts": (c_int, [c_int]),
    "srk_conv_out_dim": (c_int, [c_int] * 6),
    "srk_nchw_to_nhwc": (c_int, [c_f, c_f, c_int, c_int, c_int, c_int, c_vp]),
    "srk_nhwc_to_nchw": (c_int, [c_f, c_f, c_int, c_int, c_int, c_int, c_vp]),
    "srk_pack_weight_fwd": (c_int, [c_f, c_f, c_int, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "srk_pack_weight_bwd": (c_int, [c_f, c_f, c_int, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "srk_pack_bias_ps": (c_int, [c_f, c_f, c_int, c_int, c_vp]),
    "srk_wgrad_reduce_defer": (c_int, [c_int]),
    "srk_wgrad_reduce_flush": (c_int, [c_vp]),
    "srk_pack_weights_batched": (c_int, [c_f, c_vp, c_vp, c_int, c_int, c_vp, c_int, c_vp]),
    "srk_packed_weight_bytes": (c_size, [c_int, c_int, c_int, c_int, c_int]),
    "srk_conv2d_forward": (c_int, [ctypes.POINTER(ConvDesc), c_f, c_f, c_f, ctypes.POINTER(Epilogue), c_vp]),
    "srk_conv2d_forward_ex": (c_int, [ctypes.POINTER(ConvDesc), c_f, c_f, c_f, ctypes.POINTER(Epilogue),
                                      ctypes.POINTER(ConvResult), c_vp]),
    "srk_conv2d_backward_data": (c_int, [ctypes.POINTER(ConvDesc), c_f, c_f, c_f, ctypes.POINTER(BwdMask), c_f,
                                         c_vp]),
    "srk_conv2d_backward_data_relu_supported": (c_int, [ctypes.POINTER(ConvDesc), c_f, c_f, ctypes.POINTER(BwdMask)]),
    "srk_conv2d_backward_data_relu": (c_int, [ctypes.POINTER(ConvDesc), c_f, c_f, c_f, ctypes.POINTER(BwdMask), c_f,
                                              c_vp]),
    "srk_conv2d_backward_weight_workspace_bytes": (c_size, [ctypes.POINTER(ConvDesc)]),
    "srk_conv2d_backward_weight": (c_int, [ctypes.POINTER(ConvDesc), c_f, c_f, ctypes.POINTER(BwdMask), c_f, c_f,
                                           c_float, c_vp, c_size, c_vp]),
    "srk_conv2d_backward_weight_grouped_workspace_bytes": (c_size, [ctypes.POINTER(ConvDesc), c_int]),
    "srk_conv2d_backward_weight_grouped": (c_int, [ctypes.POINTER(ConvDesc), c_int, ctypes.POINTER(c_vp),
                                                   ctypes.POINTER(c_vp), ctypes.POINTER(BwdMask), ctypes.POINTER(c_vp),
                                                   ctypes.POINTER(c_vp), c_float, c_vp, c_size, c_vp]),
    "srk_resblock2_supported": (c_int, [c_int, c_int, c_int, c_int]),
    "srk_resblock2_forward": (c_int, [c_int, c_int, c_int, c_int, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_int, c_f, c_f,
                                      c_vp]),
    "srk_absmax": (c_int, [c_f, c_size, c_f, c_vp]),
    "srk_conv2d_f16x3_supported": (c_int, [ctypes.POINTER(ConvDesc), ctypes.POINTER(Epilogue), c_f]),
    "srk_resblock2_backward_data": (c_int, [c_int, c_int, c_int, c_int, c_f, c_f, c_f, c_f, c_f, c_f, c_int, c_vp]),
    "srk_pixel_shuffle_forward": (c_int, [c_f, c_f, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "srk_pixel_shuffle_backward": (c_int, [c_f, c_f, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "srk_act_forward": (c_int, [c_f, c_f, c_size, c_int, c_int, c_float, c_f, c_int, c_vp]),
    "srk_act_backward": (c_int, [c_f, c_f, c_f, c_size, c_int, c_int, c_float, c_f, c_int, c_f, c_vp]),
    "srk_axpby": (c_int, [c_f, c_f, c_f, c_size, c_float, c_float, c_vp]),
    "srk_loss_workspace_bytes": (c_size, []),
    "srk_loss_forward_backward": (c_int, [c_int, c_f, c_f, ctypes.POINTER(ctypes.c_int64), c_int, c_int, c_int,
                                          c_int, c_float, c_float, c_f, c_f, c_vp, c_vp]),
    "srk_sgd_step": (c_int, [c_f, c_f, c_f, c_size, c_float, c_float, c_float, c_int, c_int, c_f, c_f, c_vp]),
    "srk_adam_step": (c_int, [c_f, c_f, c_f, c_f, c_size, c_float, c_float, c_float, c_float, c_float, c_vp, c_f,
                              c_f, c_vp]),
    "srk_grad_norm_workspace_bytes": (c_size, []),
    "srk_grad_norm_clip": (c_int, [c_f, c_size, c_float, c_f, c_f, c_vp, c_vp]),
    "srk_bn_stats": (c_int, [c_f, c_vp, c_size, c_int, c_vp, c_vp]),
    "srk_bn_workspace_bytes": (c_size, [c_int]),
    "srk_bn_finalize": (c_int, [c_vp, ctypes.c_double, c_f, c_f, c_f, c_f, c_float, c_float, c_int, c_vp, c_vp]),
    "srk_bn_stats_finalize": (c_int, [c_f, c_vp, c_size, c_int, c_f, c_f, c_f, c_f, c_float, c_float, c_vp, c_vp, c_vp]),
    "srk_bn_finalize_partials": (c_int, [c_vp, c_int, c_vp, c_size, c_int, c_f, c_f, c_f, c_f, c_float, c_float, c_vp, c_vp]),
    "srk_bn_backward_stats_grads": (c_int, [c_f, c_f, c_f, c_f, c_vp, c_size, c_int, c_f, c_f, c_vp, c_vp]),
    "srk_bn_apply": (c_int, [c_f, c_f, c_f, c_f, c_f, c_f, c_size, c_int, c_int, c_float, c_f, c_vp]),
    "srk_bn_apply_act": (c_int, [c_f, c_f, c_f, c_f, c_f, c_f, c_size, c_int, c_int, c_float, c_f, c_int, c_f, c_f, c_vp]),
    "srk_bn_backward_stats_grads_act": (c_int, [c_f, c_f, c_f, c_f, c_f, c_f, c_vp, c_size, c_int, c_f, c_f, c_int,
                                                 c_float, c_f, c_int, c_f, c_vp, c_vp]),
    "srk_bn_backward_apply_act": (c_int, [c_f, c_f, c_f, c_f, c_f, c_f, c_vp, ctypes.c_double, c_f, c_size, c_int, c_int,
                                           c_float, c_f, c_int, c_vp]),
    "srk_bn_fused_supported": (c_int, [c_int]),
    "srk_bn_stats_partials": (c_int, [c_f, c_size, c_int, c_vp, ctypes.POINTER(c_int), c_vp]),
    "srk_bn_finalize_apply_act": (c_int, [c_vp, c_int, c_vp, c_size, c_int, c_f, c_f, c_f, c_f, c_float, c_float, c_vp,
                                          c_f, c_f, c_f, c_f, c_int, c_float, c_f, c_int, c_f, c_f, c_vp]),
    "srk_bn_backward_partials_act": (c_int, [c_f, c_f, c_f, c_f, c_f, c_f, c_size, c_int, c_int, c_float, c_f, c_int, c_vp,
                                             ctypes.POINTER(c_int), c_vp]),
    "srk_bn_backward_finalize_apply_act": (c_int, [c_vp, c_int, c_vp, ctypes.c_double, c_f, c_f, c_f, c_f, c_f, c_f, c_f,
                                                   c_size, c_int, c_f, c_f, c_int, c_float, c_f, c_int, c_f, c_vp]),
    "srk_bn_eval_params": (c_int, [c_f, c_f, c_float, c_f, c_f, c_int, c_vp]),
    "srk_rownorm_forward": (c_int, [c_f, c_f, c_f, c_f, c_int, c_int, c_float, c_vp]),
    "srk_rownorm_backward": (c_int, [c_f, c_f, c_f, c_f, c_f, c_int, c_int, c_vp]),
    "srk_bn_backward_stats": (c_int, [c_f, c_f, c_f, c_f, c_vp, c_size, c_int, c_vp, c_vp]),
    "srk_bn_backward_apply": (c_int, [c_f, c_f, c_f, c_f, c_f, c_vp, ctypes.c_double, c_f, c_size, c_int, c_vp]),
    "srk_bn_param_grads": (c_int, [c_vp, c_f, c_f, c_int, c_vp]),
    "srk_scale_dev": (c_int, [c_f, c_f, c_f, c_size, c_vp]),
    "srk_linear_forward": (c_int, [c_f, c_f, c_f, c_f, c_int, c_int, c_int, c_int, c_float, c_vp]),
    "srk_linear_backward": (c_int, [c_f, c_f, c_f, c_f, c_f, c_f, c_int, c_int, c_int, c_float, c_vp]),
    "srk_img_resize_u8_workspace_bytes": (c_size, [c_int] * 6),
    "srk_img_resize_u8": (c_int, [c_vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, c_vp, c_int, c_int, c_int, c_int,
                                  c_int, c_int, c_int, c_vp, c_size, c_vp]),
    "srk_patch_augment_u8": (c_int, [c_vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, c_vp, c_int, c_int, c_int,
                                     c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "srk_patch_from_image_u8_workspace_bytes": (c_size, [c_int, c_int, c_int, c_int, c_int]),
    "srk_patch_from_image_u8": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                        c_int, c_vp, c_vp, c_size, c_vp]),
    "srk_psnr_workspace_bytes": (c_size, []),
    "srk_psnr": (c_int, [c_f, ctypes.POINTER(ctypes.c_int64), c_f, ctypes.POINTER(ctypes.c_int64), c_int, c_int, c_int,
                         c_int, c_f, c_f, c_vp, c_vp]),
    "srk_channel_affine": (c_int, [c_f, c_f, c_size, c_int, c_size, ctypes.POINTER(c_float), ctypes.POINTER(c_float),
                                   c_int, c_vp]),
    "srk_upsample_nearest_forward": (c_int, [c_f, c_f, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "srk_upsample_nearest_backward": (c_int, [c_f, c_f, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "srk_maxpool2x2_forward": (c_int, [c_f, c_f, c_int, c_int, c_int, c_int, c_vp]),
    "srk_img_interp_workspace_bytes": (ctypes.c_size_t, [c_int] * 7),
    "srk_img_interp": (c_int, [c_f, c_f, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, ctypes.c_size_t, c_vp]),
}

_lib = None


def header_symbols():
    """Every function name declared in include/srk.h (used by the CPU-side export test)."""
    with open(HEADER_PATH) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(srk_[a-z0-9_]+)\s*\(", text)))


def load():
    """Load libsrk.so (building it is __graft_entry__.build()'s job). Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libsrk.so not found at %s — run `python __graft_entry__.py build` (hipcc --offload-arch=gfx950). "
            "There is no CPU/eager fallback for the hot path." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


ERR_UNSUPPORTED = -2   # SRK_ERR_UNSUPPORTED (include/srk.h)


def check(rc, what):
    if rc != 0:
        lib = load()
        raise RuntimeError("%s failed: %s (%s)" % (what, lib.srk_status_string(rc).decode(),
                                                   lib.srk_last_error_string().decode()))


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "the MI355X hot path only runs on GPU tensors (got a %s tensor); there is no CPU fallback — "
                "use oracle/ for a CPU reference" % t.device)
        if t is not None and t.device.index is not None and t.device.index != torch.cuda.current_device():
            # kernels are enqueued on the CURRENT device's stream (one process per GPU): a tensor of another device
            # would be dereferenced by the wrong GPU
            raise RuntimeError("tensor on %s but the current device is cuda:%d — call torch.cuda.set_device(%d) (one "
                               "process per GPU) or wrap the call in torch.cuda.device(...)"
                               % (t.device, torch.cuda.current_device(), t.device.index))
        if t is not None and t.dtype != torch.float32:
            raise RuntimeError("the hot path is fp32 (got %s)" % t.dtype)
