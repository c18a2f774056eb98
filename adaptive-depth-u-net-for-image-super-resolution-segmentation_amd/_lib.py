"""ctypes binding of the C ABI declared in include/adunet.h.

The HIP library is the product: if ``csrc/libadunet_hip.so`` is missing or fails to load, every
compute entry point raises -- there is no CPU or PyTorch fallback anywhere in this package.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# ADUNET_LIB overrides the path (diagnostic / ablation builds of the same C ABI)
LIB_PATH = os.environ.get("ADUNET_LIB") or os.path.join(_HERE, "csrc", "libadunet_hip.so")

AD_F32, AD_BF16, AD_F16 = 0, 1, 2
EPI_NONE, EPI_RELU = 0, 1

_vp, _i, _i64, _f, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t

# name -> (restype, argtypes); must list every symbol of include/adunet.h
SIGNATURES = {
    "ad_version": (_i, []),
    "ad_last_error": (C.c_char_p, []),
    "ad_device_cus": (_i, []),
    "ad_set_option": (_i, [C.c_char_p, _i]),
    "ad_conv3x3_mosaic": (_i, [_i, _i, _i, _i, _i, _i, _i, _i]),
    "ad_get_option": (_i, [C.c_char_p]),
    "ad_cin_granule": (_i, [_i]),
    "ad_pad_channels": (_i, [_vp, _vp, _i64, _i, _i, _i, _vp]),
    "ad_conv3x3_pack": (_i, [_vp, _i, _i, _i, _vp, _vp, _i, _vp]),
    "ad_conv3x3_pack_elems": (_sz, [_i, _i, _i]),
    "ad_conv3x3_pack_job_bytes": (_sz, []),
    "ad_conv3x3_pack_job_blocks": (_i, [_i, _i]),
    "ad_conv3x3_pack_batch": (_i, [_vp, _i, _i, _i, _vp]),
    "ad_conv3x3_fwd_ws_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "ad_conv3x3_fwd": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _sz, _i, _vp]),
    "ad_conv3x3_ln_relu_is_fused": (_i, [_i, _i, _i, _i, _i, _i, _i]),
    "ad_conv3x3_ln_stats_is_fused": (_i, [_i, _i, _i, _i, _i, _i, _i]),
    "ad_conv3x3_ln_relu_fwd": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _sz,
                                    _i, _vp]),
    "ad_conv3x3_dgrad_relu_is_fused": (_i, [_i, _i, _i, _i, _i, _i, _i]),
    "ad_conv3x3_dgrad_relu_ws_bytes": (_sz, []),
    "ad_conv3x3_dgrad_relu": (_i, [_vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp, _sz, _i, _vp]),
    "ad_conv3x3_dgrad_ln_bwd_is_fused": (_i, [_i, _i, _i, _i, _i, _i]),
    "ad_conv3x3_dgrad_ln_bwd_ws_bytes": (_sz, []),
    "ad_conv3x3_dgrad_ln_bwd": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _sz, _i, _vp]),
    "ad_conv3x3_c3_supported": (_i, [_i, _i, _i, _i, _i]),
    "ad_conv3x3_c3_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "ad_conv3x3_c3_ln_relu_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "ad_conv3x3_c3_wgrad_ws_bytes": (_sz, [_i, _i, _i]),
    "ad_conv3x3_c3_wgrad": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _sz, _i, _vp]),
    "ad_conv3x3_wgrad_ws_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "ad_conv3x3_wgrad": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _i, _vp]),
    "ad_layernorm_relu_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _f, _i, _i, _vp]),
    "ad_layernorm_bwd_ws_bytes": (_sz, [_i64, _i]),
    "ad_layernorm_relu_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _vp, _sz, _i, _vp]),
    "ad_relu_bwd": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _vp, _sz, _i, _vp]),
    "ad_resample": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "ad_resample_ln_bwd_supported": (_i, [_i, _i, _i, _i, _i, _i]),
    "ad_resample_ln_bwd_ws_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "ad_resample_ln_bwd": (_i, [_vp] * 11 + [_vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _sz, _i, _vp]),
    "ad_head_ws_bytes": (_sz, [_i, _i]),
    "ad_head_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _i, _f, _vp, _sz, _i, _vp]),
    "ad_head_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _i, _f, _f, _vp, _vp, _sz, _i, _vp]),
    "ad_head_ln_bwd_ws_bytes": (_sz, [_i, _i]),
    "ad_head_ln_bwd": (_i, [_vp] * 16 + [_i, _i64, _i, _i, _f, _f, _vp, _vp, _vp, _vp, _sz, _i, _vp]),
    "ad_adam_step": (_i, [_vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _i, _f, _vp]),
    "ad_adam_alpha": (_f, [_f, _f, _f, _i]),
    "ad_adam_step_dev": (_i, [_vp, _vp, _vp, _vp, _i64, _vp, _f, _f, _f, _f, _vp]),
    "ad_cast": (_i, [_vp, _i, _vp, _i, _i64, _vp]),
    "ad_batchnorm_ws_bytes": (_sz, [_i]),
    "ad_batchnorm_relu_fwd_train": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _i64, _i, _f, _i, _vp, _sz, _i, _vp]),
    "ad_batchnorm_relu_pool_fwd_train": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _i, _i, _i, _i, _f, _i, _vp, _sz,
                                              _i, _vp]),
    "ad_batchnorm_relu_bwd_dbias": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _vp, _sz, _i, _vp]),
    "ad_batchnorm_relu_fwd_infer": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _f, _i, _i, _vp]),
    "ad_batchnorm_relu_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _vp, _sz, _i, _vp]),
    "ad_colsum": (_i, [_vp, _vp, _i64, _i, _vp, _sz, _i, _vp]),
    "ad_maxpool2_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ad_maxpool2_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ad_maxpool2_bwd_add": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ad_pixel_shuffle2": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ad_seg_head_ws_bytes": (_sz, [_i, _i]),
    "ad_seg_head_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _vp, _sz, _i, _vp]),
    "ad_pw_gemm_tile_channels": (_i, [_i64, _i, _i, _i]),
    "ad_pw_gemm_tile_order": (_i, [_i, _i, _i, _vp, _i]),
    "ad_seg_metrics": (_i, [_vp, _i, _f, _f, _f, _f, _vp, _vp]),
    "ad_seg_head_fwd_counts": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _vp, _sz, _i, _vp]),
    "ad_seg_head_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _f, _f, _f, _vp, _vp, _sz, _i, _vp]),
    "ad_softmax_head_fwd": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp]),
    "ad_pw_supported": (_i, [_i64, _i, _i, _i]),
    "ad_pw_bank_elems": (_sz, [_i, _i]),
    "ad_pw_bank_pack": (_i, [_vp, _i, _i, _vp, _vp, _i, _vp]),
    "ad_pw_gemm": (_i, [_vp, _vp, _vp, _i64, _i, _i, _i, _vp]),
    "ad_pw_gemm_variant": (_i, [_i64, _i, _i, _i]),
    "ad_pw_bank_grad": (_i, [_vp, _i, _i, _vp, _vp]),
    "ad_pw_wgrad_supported": (_i, [_i64, _i, _i, _i]),
    "ad_pw_wgrad_ws_bytes": (_sz, [_i64, _i, _i]),
    "ad_pw_wgrad": (_i, [_vp, _vp, _vp, _i64, _i, _i, _vp, _sz, _i, _vp]),
    "ad_upconv_gather_fwd_supported": (_i, [_i, _i, _i]),
    "ad_upconv_slab_cols": (_i, [_vp, _i, _i, _i, _i]),
    "ad_upconv_gather_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "ad_upconv_gather_bwd_supported": (_i, [_i]),
    "ad_upconv_gather_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "ad_comm_unique_id": (_i, [_vp]),
    "ad_comm_create": (_i, [_vp, _i, _i, C.POINTER(C.c_void_p)]),
    "ad_comm_destroy": (_i, [_vp]),
    "ad_allreduce_bucket": (_i, [_vp, _vp, _i64, _vp]),
    "ad_u8_to_float_pad": (_i, [_vp, _vp, _vp, _i64, _vp]),
    "ad_pad_clip_f32": (_i, [_vp, _vp, _i64, _i, _i, _vp]),
    "ad_take_channels": (_i, [_vp, _vp, _i64, _i, _i, _vp]),
    "ad_luma_bt601": (_i, [_vp, _vp, _i64, _vp]),
    "ad_metrics_ws_bytes": (_sz, [_i, _i, _i]),
    "ad_mse_per_image": (_i, [_vp, _vp, _i, _i, _i, _i64, _i, _vp, _vp, _sz, _vp]),
    "ad_ssim_per_image": (_i, [_vp, _vp, _i, _i, _i, _i64, _i, _f, _vp, _vp, _sz, _vp]),
    "ad_avgpool2_plane": (_i, [_vp, _i, _i, _i, _i64, _i, _vp, _vp]),
    "ad_loss_scale_check": (_i, [_vp, _i64, _vp, _vp]),
    "ad_loss_scale_update": (_i, [_vp, _i, _vp]),
    "ad_adam_step_scaled": (_i, [_vp, _vp, _vp, _vp, _i64, _vp, _f, _f, _f, _f, _vp, _vp]),
}

_lib = None


class AdunetError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle; raise loudly if the HIP library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AdunetError(
            f"HIP extension not built: {LIB_PATH} is missing. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    import torch  # noqa: F401  -- load torch's HIP runtime first so both share one libamdhip64
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    # A/B switches: the HOST reads the environment and sets the library's explicit options (ad_set_option)
    for env, opt in (("ADUNET_NO_MAP1", b"no_map1"), ("ADUNET_NO_MAP4", b"no_map4"), ("ADUNET_NO_DGRAD_LN", b"no_dgrad_ln"),
                     ("ADUNET_NO_MOSAIC", b"no_mosaic"), ("ADUNET_NO_PW_WIDE", b"no_pw_wide")):
        if os.environ.get(env):
            lib.ad_set_option(opt, 1)
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().ad_last_error().decode("utf-8", "replace")
        if rc == -1:
            raise ValueError(f"{what}: {msg}")
        raise AdunetError(f"{what}: rc={rc}: {msg}")
