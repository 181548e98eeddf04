"""ctypes binding of libmaskbit_hip.so (include/maskbit_hip.h).

There is deliberately NO fallback: if the shared library is missing or fails to load the
product path raises.  (CPU restatements live in ``oracle/`` and are test infrastructure.)
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmaskbit_hip.so")
ABI_VERSION = 8


class GenCfg(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("bits", "splits", "hidden", "heads", "depth", "mlp", "seq", "nclass", "prenorm", "embed_tables", "precision")]


class DecCfg(C.Structure):
    _fields_ = [("token_size", C.c_int), ("hidden_channels", C.c_int), ("num_resolutions", C.c_int),
                ("num_res_blocks", C.c_int), ("num_channels", C.c_int), ("channel_mult", C.c_int * 8),
                ("latent_size", C.c_int), ("build_encoder", C.c_int), ("sample_with_conv", C.c_int),
                ("enc_res_blocks", C.c_int)]


class SamplePlan(C.Structure):
    _fields_ = [("num_steps", C.c_int), ("use_guidance", C.c_int), ("scale", C.POINTER(C.c_float)),
                ("temperature", C.POINTER(C.c_float)), ("mask_len", C.POINTER(C.c_int)), ("step_begin", C.c_int), ("step_end", C.c_int)]


# name -> (restype, argtypes); every symbol include/maskbit_hip.h (the ABI) and include/maskbit_hip_diag.h (single-kernel test entries) declare
SIGNATURES = {
    "mb_gen_saturation_count": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint), C.c_int, C.c_void_p]),
    "mb_abi_version": (C.c_int, []),
    "mb_last_error": (C.c_char_p, []),
    "mb_gen_create": (C.c_int, [C.POINTER(GenCfg), C.c_int, C.POINTER(C.c_void_p)]),
    "mb_gen_destroy": (None, [C.c_void_p]),
    "mb_gen_load": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int, C.c_void_p]),
    "mb_gen_set_wcorr": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "mb_gen_set_alo": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "mb_gen_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "mb_gen_forward_cfg": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "mb_gen_forward_attn": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "mb_sample_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_int,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "mb_dec_create": (C.c_int, [C.POINTER(DecCfg), C.c_int, C.POINTER(C.c_void_p)]),
    "mb_dec_destroy": (None, [C.c_void_p]),
    "mb_dec_load": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int, C.c_void_p]),
    "mb_dec_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "mb_dec_saturation_count": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint), C.c_int, C.c_void_p]),
    "mb_enc_encode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "mb_sample": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(SamplePlan), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mb_gemm_ex": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "mb_layernorm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "mb_w4_from_f32": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mb_w4lo_from_f32": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mb_layernorm_f4": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_int, C.c_int, C.c_void_p]),
    "mb_attention_pair": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "mb_attention_pair_f4": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "mb_gemm_mini": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                               C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.c_void_p]),
    "mb_gemm_mini_seq": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                   C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.c_void_p]),
    "mb_gemm_mini_split": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                     C.POINTER(C.c_void_p), C.c_void_p]),
    "mb_gemm_act_split": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                    C.c_int, C.c_void_p]),
    "mb_gemm": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                          C.c_int, C.c_int, C.c_void_p]),
    "mb_set_cu_count": (C.c_int, [C.c_int]),
    "mb_prof_enable": (C.c_int, [C.c_int]),
    "mb_prof_read": (C.c_int, [C.c_char_p, C.c_int]),
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load (once) and type the shared library.  torch is imported first so that the HIP runtime
    the library binds to is the one torch already loaded (same SONAME libamdhip64.so.7)."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  (loads libamdhip64 before our DT_NEEDED entry is resolved)
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the MaskBit HIP kernels are not built. Run `python -c 'import __graft_entry__ as g; "
            f"g.build()'` (needs hipcc, gfx950). There is no CPU fallback in maskbit_amd.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header and library out of sync
        fn.restype = res
        fn.argtypes = args
    v = lib.mb_abi_version()
    if v != ABI_VERSION:
        raise RuntimeError(f"libmaskbit_hip.so ABI {v} != binding ABI {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().mb_last_error()
        raise RuntimeError(f"{what or 'libmaskbit_hip'} failed (code {rc}): {msg.decode() if msg else '?'}")


def prof_enable(on, every: int = 1) -> None:
    """HIP-event kernel timing on the launch stream; ``every`` = n times the kernels of every n-th generator forward only."""
    load().mb_prof_enable(max(1, int(every)) if on else 0)


def prof_read() -> dict:
    """-> {kernel name: (calls, total_ms)} measured with HIP events on the launch stream."""
    buf = C.create_string_buffer(1 << 16)
    n = load().mb_prof_read(buf, len(buf))
    if n < 0:
        check(n, "mb_prof_read")
    out = {}
    for line in buf.value.decode().splitlines():
        name, calls, ms = line.split()
        out[name] = (int(calls), float(ms))
    return out
