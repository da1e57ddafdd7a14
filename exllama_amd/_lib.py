"""ctypes binding of libexl_amd.so (the C ABI declared in include/exl_amd.h).

The library is the product: if it is missing or fails to load this module raises -- there is no
CPU / PyTorch fallback anywhere in the package.  Build it with `python __graft_entry__.py build`
(or `make -C exllama_amd/csrc`).
"""

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libexl_amd.so")

c_void_p, c_int, c_float, c_size_t = C.c_void_p, C.c_int, C.c_float, C.c_size_t


class ExlTuning(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "matmul_recons_thd", "fused_mlp_thd", "sdp_thd", "matmul_fused_remap", "rmsnorm_no_half2",
        "rope_no_half2", "matmul_no_half2", "silu_no_half2", "concurrent_streams")]


class ExlSampler(C.Structure):
    """include/exl_amd.h: ExlSampler (defaults = the reference's ExLlamaGenerator.Settings, generator.py:9-22)."""
    _fields_ = [("temperature", C.c_float), ("top_k", C.c_int32), ("top_p", C.c_float), ("min_p", C.c_float), ("typical", C.c_float),
                ("rep_penalty_max", C.c_float), ("rep_sustain", C.c_int32), ("rep_decay", C.c_int32), ("banned_token", C.c_int32),
                ("reserved", C.c_int32), ("seed", C.c_uint64)]

    def __init__(self, temperature=0.95, top_k=40, top_p=0.65, min_p=0.0, typical=0.0, rep_penalty_max=1.15, rep_sustain=256,
                 rep_decay=128, banned_token=-1, seed=0):
        super().__init__(float(temperature), int(top_k), float(top_p), float(min_p), float(typical), float(rep_penalty_max),
                         int(rep_sustain), int(rep_decay), int(banned_token), 0, int(seed))


# name -> (restype, argtypes); every symbol include/exl_amd.h declares
SIGNATURES = {
    "exl_last_error": (C.c_char_p, []),
    "exl_version": (c_int, []),
    "exl_set_tuning": (c_int, [C.POINTER(ExlTuning)]),
    "exl_get_tuning": (c_int, [C.POINTER(ExlTuning)]),
    "exl_prepare_buffers": (c_int, [c_int, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_size_t]),
    "exl_cleanup": (c_int, []),
    "exl_make_q4": (c_int, [c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, C.POINTER(c_void_p)]),
    "exl_free_q4": (c_int, [c_void_p]),
    "exl_q4_info": (c_int, [c_void_p] + [C.POINTER(c_int)] * 5 + [C.POINTER(c_void_p)]),
    "exl_q4_layout": (c_int, [c_void_p, C.POINTER(c_int)]),
    "exl_q4_matmul": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "exl_q4_matmul_gemv": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "exl_q4_matmul_gemm": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "exl_q4_matmul_dual": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, C.POINTER(c_int)]),
    "exl_q4_qkv_rope_cache": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                              c_int, c_int, c_int, c_int, c_int, c_void_p, C.POINTER(c_int)]),
    "exl_q4_attn_prompt": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, C.c_float, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                           c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, C.POINTER(c_int)]),
    "exl_q4_mlp_prompt": (c_int, [c_void_p, c_void_p, C.c_float, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, C.POINTER(c_int)]),
    "exl_q4_layer_prompt": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_float] + [c_void_p] * 7 + [c_void_p] * 4 + [c_int] * 4 +
                            [c_void_p, c_void_p, c_size_t, c_int, C.POINTER(c_int), C.POINTER(c_int)]),
    "exl_frag_bytes": (C.c_size_t, [c_int, c_int]),
    "exl_q4_matmul_frag": (c_int, [C.POINTER(c_void_p), c_int, c_void_p, c_int, c_void_p, c_float, C.POINTER(c_void_p), c_int, c_int, c_void_p, c_int,
                           c_void_p, c_void_p, c_int, c_void_p, C.POINTER(c_int), C.POINTER(c_int)]),
    "exl_q4_matmul_lora": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "exl_q4_reconstruct": (c_int, [c_void_p, c_void_p, c_void_p]),
    "exl_column_remap": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "exl_half_matmul": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "exl_rms_norm": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_int, c_int, c_void_p]),
    "exl_embedding": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "exl_head_matmul": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "exl_rope": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "exl_silu_mul": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "exl_update_cache": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "exl_attention": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "exl_q4_attn": (c_int, [c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                            c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                            c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int,
                            c_void_p, c_void_p]),
    "exl_q4_attn_2": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "exl_q4_mlp": (c_int, [c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_int, c_int,
                           c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int,
                           c_void_p, c_void_p]),
    "exl_decoder_create": (c_int, [c_int] * 9 + [c_float] + [c_void_p] * 5 + [C.POINTER(c_void_p)]),
    "exl_decoder_set_layer": (c_int, [c_void_p, c_int] + [c_void_p] * 11),
    "exl_decoder_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "exl_decoder_set_kv_splits": (c_int, [c_void_p, c_int, C.POINTER(c_int)]),
    "exl_decoder_step_greedy": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "exl_decoder_step_timed": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, C.POINTER(c_float)]),
    "exl_decoder_free": (c_int, [c_void_p]),
    "exl_sample": (c_int, [c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "exl_decoder_step_sample": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "exl_decoder_hidden": (c_int, [c_void_p, C.POINTER(c_void_p)]),
    "exl_decoder_set_hidden": (c_int, [c_void_p, c_void_p]),
    "exl_decoder_plan": (c_int, [c_void_p, c_int, C.POINTER(c_int)]),
    "exl_decoder_step_part": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "exl_decoder_set_tp": (c_int, [c_void_p, c_int]),
    "exl_decoder_set_lora": (c_int, [c_void_p, c_int, C.POINTER(c_void_p), C.POINTER(c_void_p), C.POINTER(c_int)]),
    "exl_decoder_set_option": (c_int, [c_void_p, c_int, c_int]),
    "exl_rep_penalty": (c_int, [c_int, c_void_p, c_void_p, c_float, c_int, c_int, c_int]),
    "exl_apply_rep_penalty": (c_int, [c_int, c_void_p, c_float, c_int, c_int, c_int, c_int, c_void_p]),
}

# extended (beyond the reference surface) entry points, bound when present
OPTIONAL_SIGNATURES = {}

_lib = None


def load():
    """dlopen the library once; raises RuntimeError with build instructions when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"exllama_amd: native library not found at {LIB_PATH}. This package has no fallback path: "
            f"build the HIP extension first (python __graft_entry__.py build, or make -C exllama_amd/csrc).")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # e.g. libamdhip64 missing
        raise RuntimeError(f"exllama_amd: failed to load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    for name, (res, args) in OPTIONAL_SIGNATURES.items():
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    """Turn a non-zero return code into RuntimeError (the reference's TORCH_CHECK -> RuntimeError convention)."""
    if rc != 0:
        msg = load().exl_last_error()
        raise RuntimeError(f"{what}: {msg.decode() if msg else 'error'} (code {rc})")
