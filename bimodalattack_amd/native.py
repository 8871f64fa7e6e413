"""ctypes binding of ``libbma_hip.so`` -- the C ABI declared in ``include/bma.h``.

This is the only way the engine reaches its kernels: raw device pointers, sizes
and the HIP stream cross the boundary; no torch type does.  There is NO fallback:
if the library is missing or a symbol is absent, importing this module raises --
the product path must never silently run on something else.

torch is imported first on purpose: the PyTorch-ROCm wheel ships its own
``libamdhip64.so.7`` and the library here must bind to THAT runtime (same SONAME,
so the loader reuses it); two HIP runtimes in one process would not share
streams or allocations.  ``check_single_hip_runtime`` verifies it.
"""

from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

import torch  # noqa: F401  (loads the HIP runtime the library must share)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BMA_LIB", os.path.join(_HERE, "lib", "libbma_hip.so"))

BMA_F32, BMA_BF16, BMA_F16 = 0, 1, 2
BMA_SEG_SHARED, BMA_SEG_PERCAND, BMA_SEG_GATHER = 0, 1, 2
BMA_MAX_SEGS = 8
ABI_VERSION = 112


class BmaSegment(Structure):
    _fields_ = [("ptr", c_void_p), ("len", c_int32), ("kind", c_int32)]


class BmaError(RuntimeError):
    def __init__(self, fn: str, code: int, text: str):
        super().__init__(f"{fn} failed: {text} (code {code})")
        self.code = code


# name -> (restype, argtypes): exactly the prototypes of include/bma.h
PROTOTYPES = {
    "bma_version": (c_int, []),
    "bma_strerror": (c_char_p, [c_int]),
    "bma_quick_gelu": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "bma_quick_gelu_bwd": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "bma_allgather_f32": (c_int, [c_void_p, c_int64, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "bma_linf_step": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_void_p, c_void_p]),
    "bma_ce_target_ws_bytes": (c_size_t, [c_int, c_int]),
    "bma_ce_target": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                              c_void_p, c_void_p, c_void_p, c_float, c_void_p]),
    "bma_mask_topk_ws_bytes": (c_size_t, [c_int, c_int, c_int]),
    "bma_mask_topk": (c_int, [c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "bma_rand_positions": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "bma_sample_scatter": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                   c_void_p, c_void_p]),
    "bma_splice": (c_int, [POINTER(BmaSegment), c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int,
                           c_float, c_void_p, c_void_p]),
    "bma_splice_rows": (c_int, [POINTER(BmaSegment), c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int,
                                c_float, c_void_p, c_int64, c_void_p, c_void_p]),
    "bma_add_rmsnorm": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_float, c_int64, c_int, c_int, c_int,
                                c_void_p, c_void_p, c_void_p]),
    "bma_add_rmsnorm_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int64, c_int, c_int, c_int, c_void_p,
                                    c_void_p]),
    "bma_add_layernorm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                  c_void_p]),
    "bma_add_layernorm_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "bma_rope2": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int,
                          c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int,
                          c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_float, c_int, c_void_p]),
    "bma_qknorm_rope2": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int,
                                 c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int,
                                 c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "bma_rmsnorm": (c_int, [c_void_p, c_void_p, c_float, c_int64, c_int, c_int, c_int, c_void_p, c_void_p]),
    "bma_swiglu": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "bma_gated_act": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "bma_gated_act_il": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "bma_gated_act_il_bwd": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "bma_gated_act_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "bma_rope": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int, c_int, c_int, c_int,
                         c_void_p, c_void_p, c_int, c_float, c_int, c_void_p]),
    "bma_rope_inplace": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                 c_int, c_int, c_void_p]),
    "bma_rmsnorm_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_int64, c_int, c_int, c_int, c_void_p, c_void_p]),
    "bma_swiglu_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "bma_attn_merge": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p,
                               c_void_p]),
    "bma_attn_merge_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int,
                                    c_int, c_int, c_void_p, c_void_p]),
    "bma_gather_rows": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p]),
    "bma_ragged_attention": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64,
                                     c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p,
                                     c_void_p, c_int, c_int, c_int64, c_int, c_int, c_int, c_int, c_float, c_void_p,
                                     c_void_p, c_void_p, c_void_p]),
    "bma_ragged_attention_set_long": (None, [c_int, c_int]),
    "bma_prefix_attention": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int,
                                     c_int64, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "bma_b1_attention": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_int64, c_void_p,
                                 c_void_p]),
    "bma_b1_attention_bwd": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int,
                                     c_int, c_int, c_float, c_void_p, c_int64, c_void_p]),
    "bma_gemm_nt_ws_bytes": (c_size_t, [c_int, c_int, c_int]),
    "bma_gemm_nt_tiles": (c_int, [c_int, c_int, c_int]),
    "bma_gemm_nt_plan": (c_int, [c_int, c_int, c_int, POINTER(c_int)]),
    "bma_gemm_nt_set_plan": (None, [c_int, c_int, c_int, c_int]),
    "bma_gemm_nt": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p,
                            c_size_t, c_void_p, c_int, c_void_p]),
    "bma_gemm_nt_next": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p,
                                 c_size_t, c_void_p, c_int, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "bma_causal_attention": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int64,
                                     c_int64, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "bma_causal_attention_bwd": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p,
                                         c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p,
                                         c_void_p, c_int64, c_void_p, c_void_p]),
    "bma_causal_attention_set_plan": (None, [c_int64]),
    "bma_prefix_attention_set_plan": (None, [c_int]),
    "bma_causal_attention_gqa": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int64,
                                         c_int64, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "bma_causal_attention_bwd_gqa": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p,
                                             c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p,
                                             c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p]),
    "bma_gemm_mid_ws_bytes": (c_size_t, [c_int, c_int, c_int]),
    "bma_gemm_mid_plan": (c_int, [c_int, c_int, c_int, POINTER(c_int)]),
    "bma_gemm_mid_set_plan": (None, [c_int, c_int, c_int, c_int]),
    "bma_gemm_mid": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p,
                             c_size_t, c_void_p]),
    "bma_profile_enable": (c_int, [c_int]),
    "bma_profile_read": (c_int, [c_int, POINTER(c_int64), POINTER(ctypes.c_double), POINTER(ctypes.c_double)]),
    "bma_profile_kernel_name": (c_char_p, [c_int]),
}

KERNEL_IDS = {"linf": 0, "ce_rows": 1, "ce_dlogits": 2, "mask_topk": 3, "sample_scatter": 4, "splice": 5,
              "ce_rows_grad": 6, "rmsnorm": 7, "swiglu": 8, "rope": 9, "attn_merge": 10, "gather_rows": 11,
              "ragged_attn": 12, "prefix_attn": 13, "add_rmsnorm": 14, "gemm_nt": 15, "b1_attn": 16, "gemm_mid": 17, "causal_attn": 18}


def profile_enable(on: bool) -> None:
    check("bma_profile_enable", lib.bma_profile_enable(1 if on else 0))


def profile_read() -> dict:
    """{kernel: dict(symbol, launches, ms, bytes)} for every kernel launched since profile_enable(True)."""
    out = {}
    for name, k in KERNEL_IDS.items():
        n, ms, by = c_int64(0), ctypes.c_double(0.0), ctypes.c_double(0.0)
        check("bma_profile_read", lib.bma_profile_read(k, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(by)))
        out[name] = dict(symbol=lib.bma_profile_kernel_name(k).decode(), launches=n.value, ms=ms.value, bytes=by.value)
    return out


def _load() -> ctypes.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"bimodalattack_amd: HIP library not found at {LIB_PATH}. Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C bimodalattack_amd/csrc`). "
            "There is no CPU or PyTorch fallback for the attack kernels."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:  # pragma: no cover
            raise ImportError(f"bimodalattack_amd: {LIB_PATH} lacks symbol {name}; rebuild it") from e
        fn.restype, fn.argtypes = res, args
    got = lib.bma_version()
    if got != ABI_VERSION:
        raise ImportError(f"bimodalattack_amd: {LIB_PATH} has ABI version {got}, host expects {ABI_VERSION}; rebuild")
    return lib


lib = _load()


def strerror(code: int) -> str:
    return lib.bma_strerror(code).decode()


def check(fn: str, code: int) -> None:
    if code != 0:
        raise BmaError(fn, code, strerror(code))


def loaded_hip_runtimes():
    """Paths of every libamdhip64 mapped into this process."""
    paths = set()
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                if "libamdhip64" in line:
                    paths.add(line.split()[-1])
    except OSError:  # pragma: no cover
        pass
    return sorted(paths)


def check_single_hip_runtime() -> None:
    rts = loaded_hip_runtimes()
    if len(rts) > 1:
        raise RuntimeError(f"two HIP runtimes are loaded ({rts}); the kernels and torch would not share streams")
