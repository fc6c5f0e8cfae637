"""ctypes binding of libnind_hip.so (C ABI declared in include/nind_hip.h).

There is NO CPU fallback: if the shared library is missing or a call fails, an exception is raised.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libnind_hip.so")

ND_F32, ND_BF16, ND_F16 = 0, 1, 2
DTYPE = {"f32": 0, "fp32": 0, "float32": 0, "bf16": 1, "bfloat16": 1, "f16": 2, "fp16": 2, "float16": 2}
ACT = {"none": 0, "PReLU": 1, "ELU": 2, "Hardswish": 3}
KIND = {"conv3": 0, "convT3": 1, "convT2s2": 2, "conv1": 3, "conv2s2": 4}
# nd_flags (include/nind_hip.h): per-call arithmetic switches
FLAG_NO_SPLITK, FLAG_DIRECT_CONV, FLAG_W1D_REGS, FLAG_FULL_TILES, FLAG_UNFUSED_POOL = 1, 2, 4, 8, 16


class StepProfile(ctypes.Structure):
    """nd_step_profile of include/nind_hip.h"""
    _fields_ = [("ms", c_float), ("ms_xform_in", c_float), ("ms_gemm", c_float), ("ms_xform_out", c_float),
                ("form", c_int), ("kind", c_int), ("flops", c_double), ("mfma_flops", c_double), ("bytes", c_double),
                ("xform_bytes_in", c_double), ("xform_bytes_out", c_double)]


FORM_NAMES = {-1: "pool", 0: "direct", 1: "w1d_f43", 2: "w1d_f23", 3: "wino3p_f6x6"}


class NindHipError(RuntimeError):
    pass


class NindHipMissing(ImportError):
    pass


_lib = None

_SIGNATURES = {
    "nd_version": (c_int, []),
    "nd_last_error": (c_char_p, []),
    "nd_tile_grid": (c_int, [c_int] * 5 + [POINTER(c_int)] * 3),
    "nd_tile_geom": (c_int, [c_int] * 6 + [POINTER(c_int)] * 4),
    "nd_tile_gather": (c_int, [c_void_p] + [c_int] * 7 + [c_void_p, c_void_p]),
    "nd_stitch_add": (c_int, [c_void_p] + [c_int] * 5 + [c_void_p, c_int, c_int, c_void_p]),
    "nd_utnet_num_tensors": (c_int, []),
    "nd_utnet_tensor_name": (c_char_p, [c_int]),
    "nd_utnet_packed_bytes": (c_size_t, [c_int, c_int]),
    "nd_utnet_pack_weights": (c_int, [c_int, c_int, POINTER(c_void_p), c_int, c_void_p, c_size_t]),
    "nd_utnet_pack_weights_device": (c_int, [c_int, c_int, POINTER(c_void_p), c_int, c_void_p, c_size_t, c_void_p]),
    "nd_utnet_workspace_bytes": (c_size_t, [c_int] * 4),
    "nd_utnet_workspace_init": (c_int, [c_void_p, c_size_t] + [c_int] * 4 + [c_void_p]),
    "nd_utnet_workspace_bytes_hw": (c_size_t, [c_int] * 5),
    "nd_utnet_workspace_init_hw": (c_int, [c_void_p, c_size_t] + [c_int] * 5 + [c_void_p]),
    "nd_utnet_forward_hw": (c_int, [c_int] * 4 + [c_void_p] * 3 + [c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "nd_utnet_forward": (c_int, [c_int] * 4 + [c_void_p] * 3 + [c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "nd_utnet_denoise_tiles": (c_int, [c_int] * 4 + [c_void_p] * 3 + [c_int] * 8 + [c_void_p, c_size_t, c_void_p]),
    "nd_unet_num_tensors": (c_int, []),
    "nd_unet_tensor_name": (c_char_p, [c_int]),
    "nd_unet_packed_bytes": (c_size_t, [c_int]),
    "nd_unet_pack_weights": (c_int, [c_int, POINTER(c_void_p), c_int, c_void_p, c_size_t]),
    "nd_unet_workspace_bytes": (c_size_t, [c_int] * 4),
    "nd_unet_workspace_init": (c_int, [c_void_p, c_size_t] + [c_int] * 4 + [c_void_p]),
    "nd_unet_forward": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "nd_utnet_flops": (c_double, [c_int, c_int]),
    "nd_utnet_profile_stack": (c_int, [c_int] * 4 + [c_void_p, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p, c_void_p, c_int]),
    "nd_utnet_step_name": (c_char_p, [c_int]),
    "nd_utnet_useful_region": (c_int, [c_int, c_int, c_int, c_int, POINTER(c_int)]),
    "nd_layer_packed_bytes": (c_size_t, [c_int] * 4),
    "nd_layer_pack": (c_int, [c_int] * 4 + [c_void_p, c_void_p, c_void_p, c_size_t]),
    "nd_layer_workspace_bytes": (c_size_t, [c_int] * 7),
    "nd_layer_forward": (c_int, [c_int, c_int, c_float, c_int, c_void_p, c_void_p] + [c_int] * 5
                         + [c_void_p, c_void_p, c_size_t, c_int, c_int, c_void_p]),
    "nd_winograd_packed_bytes": (c_size_t, [c_int] * 3),
    "nd_winograd_pack": (c_int, [c_int] * 4 + [c_void_p, c_void_p, c_void_p, c_size_t]),
    "nd_layer_winograd_workspace_bytes": (c_size_t, [c_int] * 7),
    "nd_layer_forward_winograd": (c_int, [c_int, c_int, c_int, c_float, c_void_p, c_void_p] + [c_int] * 5
                                  + [c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "nd_maxpool2_forward": (c_int, [c_void_p] + [c_int] * 4 + [c_void_p, c_void_p, c_size_t, c_void_p]),
    "nd_layer_wgrad_workspace_bytes": (c_size_t, [c_int] * 6),
    "nd_layer_wgrad": (c_int, [c_int, c_void_p, c_void_p] + [c_int] * 5 + [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "nd_utnet_param_count": (c_size_t, [c_int]),
    "nd_utnet_param_range": (c_int, [c_int, c_int, POINTER(c_size_t), POINTER(c_size_t)]),
    "nd_utnet_train_blob_bytes": (c_size_t, [c_int]),
    "nd_utnet_train_workspace_bytes": (c_size_t, [c_int] * 3),
    "nd_utnet_train_workspace_init": (c_int, [c_void_p, c_size_t, c_int, c_int, c_int, c_void_p]),
    "nd_utnet_train_step": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float,
                                    c_float, c_float, c_void_p, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "nd_utnet_train_forward": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "nd_utnet_train_backward": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_size_t,
                                        c_void_p, POINTER(c_void_p), c_int]),
    "nd_utnet_grad_buckets": (c_int, [c_int, POINTER(c_size_t), POINTER(c_size_t), c_int]),
    "nd_utnet_train_step_ev": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float,
                                       c_float, c_float, c_void_p, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p, POINTER(c_void_p),
                                       c_int]),
    "nd_adam_step": (c_int, [c_void_p] * 5 + [c_size_t, c_float, c_float, c_float, c_float, c_int, c_int, c_void_p]),
    "nd_ssim_workspace_bytes": (c_size_t, [c_int] * 4),
    "nd_ssim": (c_int, [c_void_p, c_void_p] + [c_int] * 4 + [c_void_p, c_void_p, c_size_t, c_void_p]),
    "nd_ms_ssim": (c_int, [c_void_p, c_void_p] + [c_int] * 4 + [c_void_p, c_void_p, c_size_t, c_void_p]),
    "nd_ssim_loss_workspace_bytes": (c_size_t, [c_int] * 4),
    "nd_ssim_loss_grad": (c_int, [c_void_p, c_void_p] + [c_int] * 5 + [c_float, c_void_p, c_void_p, c_int, c_void_p, c_size_t,
                                  c_void_p]),
    "nd_mse": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_size_t, c_void_p]),
    "nd_conv_bench": (c_int, [c_int] * 9 + [c_void_p, c_size_t, c_void_p, POINTER(c_float)]),
    "nd_winograd_bench": (c_int, [c_int] * 8 + [c_void_p, c_size_t, c_void_p, POINTER(c_float)]),
    "nd_num_conv_variants": (c_int, []),
    "nd_conv_variant_name": (c_char_p, [c_int]),
}

EXPORTS = tuple(_SIGNATURES)


def load():
    """Load (once) and return the ctypes handle; raises NindHipMissing when the library was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise NindHipMissing(
            f"{LIB_PATH} not found: build it with `make -C nind_denoise_amd/csrc` (or __graft_entry__.build()). "
            "nind_denoise_amd has no CPU fallback for the denoise hot path.")
    # torch first: it ships its own HIP runtime, and the process must hold ONE copy -- a libamdhip64 pulled in by this library
    # before torch loads its own sees no device ("no ROCm-capable device is detected" from the first launch)
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().nd_last_error().decode(errors="replace")
        exc = ValueError if rc == -1 else (MemoryError if rc == -2 else NindHipError)
        raise exc(f"{what}: {msg}" if what else msg)


def stream_ptr(device=None):
    import torch
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t):
    """Raw data pointer of a torch tensor (must be contiguous)."""
    assert t.is_contiguous(), "tensor must be contiguous"
    return c_void_p(t.data_ptr())


# ---------------------------------------------------------------------------- host-only helpers

def tile_grid(width, height, cs, ucs, ol):
    cols, rows, pad = c_int(), c_int(), c_int()
    check(load().nd_tile_grid(width, height, cs, ucs, ol, cols, rows, pad), "nd_tile_grid")
    return cols.value, rows.value, pad.value


def tile_geom(i, width, height, cs, ucs, ol):
    x0, y0 = c_int(), c_int()
    ud = (c_int * 4)()
    us = (c_int * 2)()
    check(load().nd_tile_geom(i, width, height, cs, ucs, ol, x0, y0, ud, us), "nd_tile_geom")
    return x0.value, y0.value, tuple(ud), tuple(us)


def utnet_tensor_names():
    lib = load()
    return [lib.nd_utnet_tensor_name(i).decode() for i in range(lib.nd_utnet_num_tensors())]
