"""ctypes binding of libwmar_hip.so (declared in include/wmar_hip.h).

There is no CPU fallback: if the library is missing the import of any product
module that needs it fails loudly, and every entry point raises on a non-zero
status with the library's own message.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libwmar_hip.so")

# Every symbol include/wmar_hip.h declares (tests check the .so exports all of them).
SYMBOLS = [
    "wmar_last_error", "wmar_version", "wmar_key_row_words", "wmar_key_table_rows", "wmar_key_table_build",
    "wmar_key_greenlist", "wmar_wm_process_logits", "wmar_sample_fused", "wmar_detect", "wmar_detect_num_ngrams",
    "wmar_gpt_create", "wmar_gpt_destroy", "wmar_gpt_device_bytes", "wmar_gpt_decode_step", "wmar_gpt_generate",
    "wmar_gpt_set_timing", "wmar_gpt_set_attention_phases", "wmar_gpt_get_timing", "wmar_gpt_profile_role", "wmar_gpt_plan_info", "wmar_gpt_check", "wmar_rar_create", "wmar_rar_destroy",
    "wmar_rar_device_bytes", "wmar_rar_forward_position", "wmar_rar_generate", "wmar_vq_create", "wmar_vq_destroy", "wmar_vq_device_bytes",
    "wmar_vq_decode", "wmar_vq_encode", "wmar_mvq_create", "wmar_mvq_destroy", "wmar_mvq_device_bytes", "wmar_mvq_decode",
    "wmar_mvq_encode", "wmar_gumbel_key_build", "wmar_gumbel_sample", "wmar_gumbel_score", "wmar_rar_generate_gumbel", "wmar_rar_check", "wmar_rar_launch_status",
    "wmar_cham_create", "wmar_cham_destroy", "wmar_cham_device_bytes", "wmar_cham_forward_tokens", "wmar_cham_generate_image",
    "wmar_cham_sample",
    "wmar_augment",
    "wmar_comm_unique_id", "wmar_comm_init", "wmar_comm_bcast", "wmar_comm_allgather", "wmar_comm_rank", "wmar_comm_world", "wmar_comm_destroy",
]

WMAR_ESHORT = -3


class KeyParams(C.Structure):
    _fields_ = [("salt_key", C.c_uint64), ("alive_ids", C.POINTER(C.c_int64)), ("n_alive", C.c_int64),
                ("dead_ids", C.POINTER(C.c_int64)), ("n_dead", C.c_int64), ("vocab_size", C.c_int64),
                ("gamma", C.c_double), ("split_strategy", C.c_int32), ("seed_strategy", C.c_int32)]


class WmCtx(C.Structure):
    _fields_ = [("table_dev", C.c_void_p), ("n_rows", C.c_int64), ("vocab_size", C.c_int64),
                ("seed_strategy", C.c_int32), ("context_size", C.c_int32), ("spatial_dim", C.c_int32),
                ("delta", C.c_float)]


class GptConfig(C.Structure):
    _fields_ = [("vocab_size", C.c_int32), ("block_size", C.c_int32), ("n_layer", C.c_int32),
                ("n_head", C.c_int32), ("n_embd", C.c_int32), ("max_batch", C.c_int32)]


class RarConfig(C.Structure):
    _fields_ = [("hidden_size", C.c_int32), ("num_hidden_layers", C.c_int32), ("num_attention_heads", C.c_int32),
                ("intermediate_size", C.c_int32), ("image_seq_len", C.c_int32), ("codebook_size", C.c_int32),
                ("condition_num_classes", C.c_int32), ("max_batch", C.c_int32)]


class SampleParams(C.Structure):
    _fields_ = [("temperature", C.c_float), ("top_k", C.c_int32), ("top_p", C.c_double), ("use_graph", C.c_int32)]


class VqConfig(C.Structure):
    _fields_ = [("ch", C.c_int32), ("num_res_blocks", C.c_int32), ("resolution", C.c_int32),
                ("in_channels", C.c_int32), ("out_ch", C.c_int32), ("z_channels", C.c_int32),
                ("embed_dim", C.c_int32), ("n_embed", C.c_int32), ("n_levels", C.c_int32),
                ("ch_mult", C.c_int32 * 8), ("n_attn_res", C.c_int32), ("attn_resolutions", C.c_int32 * 8),
                ("max_batch", C.c_int32)]


class MvqConfig(C.Structure):
    _fields_ = [("hidden_channels", C.c_int32), ("num_res_blocks", C.c_int32), ("resolution", C.c_int32),
                ("num_channels", C.c_int32), ("z_channels", C.c_int32), ("num_embeddings", C.c_int32),
                ("n_levels", C.c_int32), ("channel_mult", C.c_int32 * 8), ("max_batch", C.c_int32)]


class ChamConfig(C.Structure):
    _fields_ = [("dim", C.c_int32), ("n_layers", C.c_int32), ("n_heads", C.c_int32), ("n_kv_heads", C.c_int32),
                ("vocab_size", C.c_int32), ("ffn_hidden", C.c_int32), ("norm_eps", C.c_float), ("rope_theta", C.c_float),
                ("qk_normalization", C.c_int32), ("swin_norm", C.c_int32), ("max_rows", C.c_int32),
                ("max_seq_len", C.c_int32), ("tensors_bf16", C.c_int32)]


class ChamSampleParams(C.Structure):
    _fields_ = [("temperature", C.c_float), ("top_p", C.c_double), ("guidance_scale_text", C.c_float),
                ("guidance_scale_image", C.c_float), ("use_graph", C.c_int32), ("pad_id", C.c_int32)]


class WmarError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libwmar_hip: {msg} (status {code})")
        self.code = code


_LIB = None


def load():
    """Load libwmar_hip.so.  torch must be imported first so that the HIP runtime the
    library binds to (libamdhip64.so.7) is the one torch already mapped."""
    global _LIB
    if _LIB is not None:
        return _LIB
    import torch  # noqa: F401  (maps torch's libamdhip64 before ours is resolved)

    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m wmar_amd.build` (hipcc, gfx950). "
            "wmar_amd has no CPU fallback.")
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    vp, i64, i32, f32, f64 = C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_double
    L.wmar_last_error.restype = C.c_char_p
    L.wmar_key_row_words.restype = i64
    L.wmar_key_row_words.argtypes = [i64]
    L.wmar_key_table_rows.restype = i64
    L.wmar_key_table_rows.argtypes = [i32, i32, i64]
    L.wmar_key_table_build.argtypes = [C.POINTER(KeyParams), i64, i64, vp, i32]
    L.wmar_key_greenlist.restype = i64
    L.wmar_key_greenlist.argtypes = [C.POINTER(KeyParams), C.c_uint64, vp]
    L.wmar_wm_process_logits.argtypes = [C.POINTER(WmCtx), vp, i64, vp, i64, i64, vp]
    L.wmar_sample_fused.argtypes = [C.POINTER(WmCtx), vp, i64, i64, vp, i64, i64, f32, i32, f64, vp, vp, vp, vp]
    L.wmar_detect.argtypes = [C.POINTER(WmCtx), f64, vp, i64, i64, vp, vp, vp, vp, i64, vp]
    L.wmar_detect_num_ngrams.restype = i64
    L.wmar_detect_num_ngrams.argtypes = [i32, i32, i64]
    L.wmar_augment.argtypes = [i32, vp, vp, vp, i64, i32, i32, i32, i32, f64, f64, vp]
    L.wmar_comm_unique_id.argtypes = [vp, i64]
    L.wmar_comm_init.argtypes = [vp, i64, i32, i32, C.POINTER(vp)]
    L.wmar_comm_bcast.argtypes = [vp, vp, i64, i32, vp]
    L.wmar_comm_allgather.argtypes = [vp, vp, vp, i64, vp]
    L.wmar_comm_rank.argtypes = [vp]
    L.wmar_comm_rank.restype = i32
    L.wmar_comm_world.argtypes = [vp]
    L.wmar_comm_world.restype = i32
    L.wmar_comm_destroy.argtypes = [vp]
    L.wmar_comm_destroy.restype = None
    L.wmar_gpt_create.argtypes = [C.POINTER(GptConfig), C.POINTER(C.c_char_p), C.POINTER(vp), i32, vp, C.POINTER(vp)]
    L.wmar_gpt_destroy.argtypes = [vp]
    L.wmar_gpt_destroy.restype = None
    L.wmar_gpt_device_bytes.restype = i64
    L.wmar_gpt_device_bytes.argtypes = [vp]
    L.wmar_gpt_decode_step.argtypes = [vp, vp, i64, i32, vp, vp]
    L.wmar_gpt_generate.argtypes = [vp, C.POINTER(WmCtx), C.POINTER(SampleParams), vp, i64, i32, vp, vp, vp, vp]
    L.wmar_gpt_set_timing.argtypes = [vp, i32]
    L.wmar_gpt_set_attention_phases.argtypes = [vp, i32, i32]
    L.wmar_gpt_profile_role.argtypes = [vp, i32, i64, i32, i32, vp, C.POINTER(f64)]
    L.wmar_gpt_get_timing.argtypes = [vp, C.POINTER(f64), C.POINTER(i64), C.POINTER(f64)]
    L.wmar_gpt_plan_info.argtypes = [vp, i64, C.c_char_p, i64]
    L.wmar_gpt_check.argtypes = [vp, vp]
    L.wmar_rar_create.argtypes = [C.POINTER(RarConfig), C.POINTER(C.c_char_p), C.POINTER(vp), i32, vp, C.POINTER(vp)]
    L.wmar_rar_destroy.argtypes = [vp]
    L.wmar_rar_destroy.restype = None
    L.wmar_rar_device_bytes.restype = i64
    L.wmar_rar_device_bytes.argtypes = [vp]
    L.wmar_rar_forward_position.argtypes = [vp, vp, vp, i64, i32, vp, vp]
    L.wmar_rar_generate.argtypes = [vp, C.POINTER(WmCtx), vp, i64, C.POINTER(f32), i32, f32, vp, vp, i32, vp]
    L.wmar_rar_generate_gumbel.argtypes = [vp, vp, i64, C.POINTER(f32), i32, f32, f32, i32, vp, vp, i32, vp]
    L.wmar_rar_check.argtypes = [vp, vp]
    L.wmar_rar_launch_status.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    L.wmar_gumbel_key_build.argtypes = [C.c_uint64, i64, vp, vp, vp]
    L.wmar_gumbel_sample.argtypes = [vp, i64, i64, vp, i64, i32, f32, f32, i32, vp, vp]
    L.wmar_gumbel_score.argtypes = [vp, i64, i64, i64, vp, i64, vp, vp, vp]
    if hasattr(L, "wmar_vq_create"):
        L.wmar_vq_create.argtypes = [C.POINTER(VqConfig), C.POINTER(C.c_char_p), C.POINTER(vp), i32, vp, C.POINTER(vp)]
        L.wmar_vq_destroy.argtypes = [vp]
        L.wmar_vq_destroy.restype = None
        L.wmar_vq_device_bytes.restype = i64
        L.wmar_vq_device_bytes.argtypes = [vp]
        L.wmar_vq_decode.argtypes = [vp, vp, i64, vp, vp]
        L.wmar_vq_encode.argtypes = [vp, vp, i64, vp, vp, vp]
    L.wmar_cham_create.argtypes = [C.POINTER(ChamConfig), C.POINTER(C.c_char_p), C.POINTER(vp), i32, vp, C.POINTER(vp)]
    L.wmar_cham_destroy.argtypes = [vp]
    L.wmar_cham_destroy.restype = None
    L.wmar_cham_device_bytes.restype = i64
    L.wmar_cham_device_bytes.argtypes = [vp]
    L.wmar_cham_forward_tokens.argtypes = [vp, vp, vp, i64, vp, vp]
    L.wmar_cham_generate_image.argtypes = [vp, C.POINTER(WmCtx), vp, vp, i64, C.POINTER(ChamSampleParams), vp, vp, i32, vp, i32, vp, vp]
    L.wmar_cham_sample.argtypes = [C.POINTER(WmCtx), vp, i64, i64, vp, i64, i64, f32, f64, f32, f32, vp, vp, i32, vp, vp, vp, vp]
    L.wmar_mvq_create.argtypes = [C.POINTER(MvqConfig), C.POINTER(C.c_char_p), C.POINTER(vp), i32, vp, C.POINTER(vp)]
    L.wmar_mvq_destroy.argtypes = [vp]
    L.wmar_mvq_destroy.restype = None
    L.wmar_mvq_device_bytes.restype = i64
    L.wmar_mvq_device_bytes.argtypes = [vp]
    L.wmar_mvq_decode.argtypes = [vp, vp, i64, vp, vp]
    L.wmar_mvq_encode.argtypes = [vp, vp, i64, vp, vp, vp]
    _LIB = L
    return L


def check(rc: int):
    if rc != 0:
        msg = load().wmar_last_error().decode("utf-8", "replace")
        if rc == WMAR_ESHORT:
            raise ValueError(msg)
        raise WmarError(rc, msg)


def stream_ptr(device=None) -> int:
    """hipStream_t of torch's current stream on `device` as an integer handle."""
    import torch

    return int(torch.cuda.current_stream(device).cuda_stream)


def tensor_table(tensors: dict):
    """(names, pointers, n, keepalive) arrays for the *_create calls."""
    names = list(tensors.keys())
    arr_n = (C.c_char_p * len(names))(*[n.encode() for n in names])
    arr_p = (C.c_void_p * len(names))(*[tensors[n].data_ptr() for n in names])
    return arr_n, arr_p, len(names)
