"""ctypes binding of libkuiper_hip.so (include/kuiper_hip.h).

The product path has NO CPU fallback: if the HIP library is missing this module raises, and
any op called without a GPU returns the library's error code as an exception.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KH_LIB") or os.path.join(_PKG, "lib", "libkuiper_hip.so")  # KH_LIB: experiment builds

# every symbol include/kuiper_hip.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "kh_error_string", "kh_version", "kh_device_count",
    "kh_add_f32", "kh_matmul_f32", "kh_matmul_q8", "kh_embedding_f32", "kh_embedding_f32_host", "kh_swiglu_f32",
    "kh_rmsnorm_f32", "kh_rope_f32", "kh_sincos_cache_f32", "kh_mha_f32", "kh_mha_decode_f32", "kh_mha_prefill_f32",
    "kh_mha_decode_workspace_bytes", "kh_argmax_f32",
    "kh_argmax_f32_host", "kh_softmax_f32", "kh_scale_f32", "kh_scale_sum_f32",
    "kh_model_create_from_file", "kh_model_create_from_host_image",
    "kh_model_create_from_device_weights", "kh_model_destroy", "kh_model_get_config",
    "kh_model_stream", "kh_model_get_load_ms", "kh_model_predict", "kh_model_get_logits", "kh_model_get_kv", "kh_model_kv_bytes", "kh_model_read_kv", "kh_model_write_kv",
    "kh_spm_create_from_file", "kh_spm_create_from_memory", "kh_spm_destroy", "kh_spm_vocab_size",
    "kh_spm_bos_id", "kh_spm_eos_id", "kh_spm_unk_id", "kh_spm_encode", "kh_spm_decode",
    "kh_bpe_create_from_file", "kh_bpe_create_from_memory", "kh_bpe_destroy", "kh_bpe_vocab_size",
    "kh_bpe_bos_id", "kh_bpe_eos_id", "kh_bpe_stop_id", "kh_bpe_encode", "kh_bpe_decode",
    "kh_model_generate", "kh_model_generate_until", "kh_model_first_sample", "kh_model_time_step", "kh_model_prefill", "kh_model_prefill_gemm", "kh_model_time_prefill", "kh_model_profile_kernel", "kh_model_profile_step", "kh_kclass_name",
    "kh_plan_decode_shapes", "kh_plan_decode_ring", "kh_plan_prefill_shape", "kh_plan_attention",
    "kh_debug_set", "kh_debug_get", "kh_debug_list",
]

KH_EXEC_GRAPH, KH_EXEC_FUSED, KH_EXEC_UNFUSED = 0, 1, 2
KH_NUM_KCLASS = 7
KH_ERR_RANGE = -6
KH_ERR_INTERNAL = -7
KH_FLAG_ATTN_MERGE_IN_LAUNCH = 1
KH_FLAG_ATTN_MERGE_FENCED = 2
KH_FLAG_PREFILL_EXACT = 4


class KhError(RuntimeError):
    def __init__(self, code: int, what: str):
        self.code = code
        super().__init__(f"{what}: kh error {code} ({error_string(code)})")


class ModelOpts(C.Structure):
    _fields_ = [("family", C.c_int32), ("is_quant", C.c_int32), ("rope_mode", C.c_int32),
                ("rope_theta", C.c_float), ("rms_eps", C.c_float), ("max_seq_len", C.c_int32),
                ("device", C.c_int32), ("flags", C.c_int32)]


class FirstSample(C.Structure):
    _fields_ = [("pos", C.c_int32), ("prefill_mode", C.c_int32), ("top1_id", C.c_int32), ("top2_id", C.c_int32),
                ("top1", C.c_float), ("top2", C.c_float)]


class Config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "dim", "hidden_dim", "layer_num", "head_num", "kv_head_num", "vocab_size", "seq_len",
        "kv_dim", "kv_mul", "head_size", "is_shared_weight", "is_quant", "group_size", "family",
        "rope_mode", "cache_len")] + [("rope_theta", C.c_float), ("rms_eps", C.c_float),
                                      ("weight_bytes", C.c_int64), ("launches_per_token", C.c_int32),
                                      ("ring_selftest", C.c_int32), ("attn_merge_selftest", C.c_int32)]


_lib: Optional[C.CDLL] = None
_vp, _i32, _i64, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float


def lib() -> C.CDLL:
    """Load the HIP library; raises (loudly) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: its wheel bundles the HIP runtime (libamdhip64.so.7); loading it before our
    # library makes both bind to ONE runtime, so torch-owned device pointers/streams are valid
    # inside libkuiper_hip.so.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m kuiperllama_amd.build` "
            "(there is no CPU fallback for the HIP path)")
    L = C.CDLL(LIB_PATH)
    L.kh_error_string.argtypes = [C.c_int]
    L.kh_error_string.restype = C.c_char_p
    L.kh_version.restype = C.c_int
    L.kh_device_count.restype = C.c_int
    L.kh_add_f32.argtypes = [_vp, _vp, _vp, _i32, _vp]
    L.kh_matmul_f32.argtypes = [_vp, _vp, _vp, _i32, _i32, _f32, _vp]
    L.kh_matmul_q8.argtypes = [_vp, _vp, _vp, _i32, _vp, _i32, _i32, _vp]
    L.kh_embedding_f32.argtypes = [_vp, _i32, _vp, _vp, _i32, _i32, _vp]
    L.kh_embedding_f32_host.argtypes = [_vp, _i32, _vp, _vp, _i32, _i32, _vp]
    L.kh_swiglu_f32.argtypes = [_vp, _vp, _vp, _i32, _vp]
    L.kh_rmsnorm_f32.argtypes = [_vp, _vp, _vp, _i32, _f32, _vp]
    L.kh_rope_f32.argtypes = [_i32, _i32, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _i32, _vp]
    L.kh_sincos_cache_f32.argtypes = [_i32, _i32, _f32, _vp, _vp, _vp]
    L.kh_mha_f32.argtypes = [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp,
                             _vp, _vp]
    L.kh_mha_decode_workspace_bytes.argtypes = [_i32, _i32, _i32]
    L.kh_mha_decode_workspace_bytes.restype = _i64
    L.kh_mha_prefill_f32.argtypes = [_i32] * 8 + [_vp, _vp, _vp, _vp, _vp]
    L.kh_mha_decode_f32.argtypes = [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp,
                                    _vp, _vp, _i64, _vp]
    L.kh_argmax_f32.argtypes = [_vp, _i64, _vp, _vp]
    L.kh_argmax_f32_host.argtypes = [_vp, _i64, C.POINTER(_i64), _vp]
    L.kh_softmax_f32.argtypes = [_vp, _i32, _vp]
    L.kh_scale_f32.argtypes = [_f32, _vp, _i32, _vp]
    L.kh_scale_sum_f32.argtypes = [_vp, _vp, _vp, _i32, _i32, _i32, _vp]
    L.kh_model_create_from_file.argtypes = [C.c_char_p, C.POINTER(ModelOpts), C.POINTER(_vp)]
    L.kh_model_create_from_host_image.argtypes = [_vp, C.c_size_t, C.POINTER(ModelOpts),
                                                  C.POINTER(_vp)]
    L.kh_model_create_from_device_weights.argtypes = [C.POINTER(_i32), _vp, C.c_size_t,
                                                      C.POINTER(ModelOpts), C.POINTER(_vp)]
    L.kh_model_destroy.argtypes = [_vp]
    L.kh_model_destroy.restype = None
    L.kh_model_get_config.argtypes = [_vp, C.POINTER(Config)]
    L.kh_model_stream.argtypes = [_vp]
    L.kh_model_stream.restype = _vp
    L.kh_model_get_load_ms.argtypes = [_vp]
    L.kh_model_get_load_ms.restype = _f32
    L.kh_model_predict.argtypes = [_vp, _i32, _i32, _i32, _i32, C.POINTER(_i32)]
    L.kh_model_get_logits.argtypes = [_vp, _vp]
    L.kh_model_get_kv.argtypes = [_vp, C.POINTER(_vp), C.POINTER(_vp)]
    L.kh_model_kv_bytes.argtypes = [_vp, C.POINTER(_i64), C.POINTER(_i64)]
    L.kh_model_read_kv.argtypes = [_vp, _i32, _i32, _i32, _vp, _vp]
    L.kh_model_write_kv.argtypes = [_vp, _i32, _i32, _i32, _vp, _vp]
    L.kh_model_generate.argtypes = [_vp, C.POINTER(_i32), _i32, _i32, _i32, C.POINTER(_i32),
                                    C.POINTER(_i32), C.POINTER(_f32)]
    L.kh_model_generate_until.argtypes = [_vp, C.POINTER(_i32), _i32, _i32, _i32, C.POINTER(_i32),
                                          _i32, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_f32)]
    L.kh_model_first_sample.argtypes = [_vp, C.POINTER(FirstSample)]
    L.kh_model_time_step.argtypes = [_vp, _i32, _i32, C.POINTER(_f32)]
    L.kh_plan_decode_shapes.argtypes = [_i32, _i32, _i32, _i32, _i32, C.POINTER(_i32)]
    L.kh_plan_decode_ring.argtypes = [_i32, _i32, _i32, _i32, _i32, C.POINTER(_i32)]
    L.kh_plan_prefill_shape.argtypes = [_i32, _i32, _i32, _i32, _i32, _i32, C.POINTER(_i32)]
    L.kh_plan_attention.argtypes = [_i32, _i32, _i32, _i32, _i32, C.POINTER(_i32)]
    L.kh_spm_create_from_file.argtypes = [C.c_char_p, C.POINTER(_vp)]
    L.kh_spm_create_from_memory.argtypes = [_vp, C.c_int64, C.POINTER(_vp)]
    L.kh_spm_destroy.argtypes = [_vp]
    L.kh_spm_destroy.restype = None
    for fn in (L.kh_spm_vocab_size, L.kh_spm_bos_id, L.kh_spm_eos_id, L.kh_spm_unk_id):
        fn.argtypes = [_vp]
        fn.restype = _i32
    L.kh_spm_encode.argtypes = [_vp, C.c_char_p, C.c_int64, _i32, _i32, C.POINTER(_i32), _i32,
                                C.POINTER(_i32)]
    L.kh_spm_decode.argtypes = [_vp, C.POINTER(_i32), _i32, C.c_char_p, C.c_int64,
                                C.POINTER(C.c_int64)]
    L.kh_bpe_create_from_file.argtypes = [C.c_char_p, _i32, C.POINTER(_vp)]
    L.kh_bpe_create_from_memory.argtypes = [_vp, C.c_int64, _i32, C.POINTER(_vp)]
    L.kh_bpe_destroy.argtypes = [_vp]
    L.kh_bpe_destroy.restype = None
    for fn in (L.kh_bpe_vocab_size, L.kh_bpe_bos_id, L.kh_bpe_eos_id):
        fn.argtypes = [_vp]
    L.kh_bpe_stop_id.argtypes = [_vp, _i32]
    L.kh_bpe_encode.argtypes = [_vp, C.c_char_p, C.c_int64, _i32, _i32, _i32, C.POINTER(_i32), _i32,
                                C.POINTER(_i32)]
    L.kh_bpe_decode.argtypes = [_vp, C.POINTER(_i32), _i32, _i32, C.c_char_p, C.c_int64,
                                C.POINTER(C.c_int64)]
    L.kh_model_prefill.argtypes = [_vp, C.POINTER(_i32), _i32, _i32]
    L.kh_model_prefill_gemm.argtypes = [_vp, C.POINTER(_i32), _i32, _i32]
    L.kh_model_time_prefill.argtypes = [_vp, C.POINTER(_i32), _i32, _i32, _i32, C.POINTER(_f32)]
    L.kh_model_profile_kernel.argtypes = [_vp, _i32, _i32, _i32, C.POINTER(_f32)]
    L.kh_model_profile_step.argtypes = [_vp, _i32, _i32, C.POINTER(_f32), C.POINTER(_i32)]
    L.kh_kclass_name.argtypes = [C.c_int]
    L.kh_kclass_name.restype = C.c_char_p
    L.kh_debug_set.argtypes = [C.c_char_p, C.c_char_p]
    L.kh_debug_get.argtypes = [C.c_char_p]
    L.kh_debug_get.restype = C.c_char_p
    L.kh_debug_list.argtypes = [C.c_char_p, _i64]
    L.kh_debug_list.restype = _i64
    for name in EXPORTS:  # fail at load time, not at first call, if a symbol is missing
        getattr(L, name)
    _lib = L
    return L


def debug_set(key: str, value: Optional[str]) -> None:
    """Set (or with None: clear) one tuning / test hook of the library (kh_debug_set).  A hook set here stays
    until it is cleared here: sync_env() only manages the hooks it mirrored from os.environ itself."""
    _ENV_MIRRORED.pop(key, None)  # from now on the key belongs to the caller, not to the environment mirror
    check(lib().kh_debug_set(key.encode(), None if value is None else str(value).encode()), "kh_debug_set")


def debug_get(key: str) -> Optional[str]:
    v = lib().kh_debug_get(key.encode())
    return None if v is None else v.decode()


_HOOK_PREFIXES = ("KH_SHAPE_", "KH_ATTN_", "KH_PG_", "KH_PREFILL", "KH_RING", "KH_SELFTEST", "KH_KV_")
_ENV_MIRRORED = {}  # hook -> value, as last written into the table FROM os.environ by sync_env()
_ENV_SEEDED = False


def _seed_mirror(L) -> None:
    """First sync: the library seeded its table from the KH_* environment when it was loaded.  Hooks that are in
    the table AND were in the environment then are environment-owned from the start (value as in the table), so
    that a variable removed from os.environ before the first sync_env() - monkeypatch.delenv - is cleared like
    any other; a key that is in the table only (kh_debug_set before the first sync) stays the caller's."""
    global _ENV_SEEDED
    _ENV_SEEDED = True
    n = L.kh_debug_list(None, 0)
    if n <= 0:
        return
    buf = C.create_string_buffer(int(n) + 1)
    L.kh_debug_list(buf, int(n) + 1)
    for k in buf.value.decode().split("\n"):
        if k and k.startswith(_HOOK_PREFIXES) and k in _ENV_AT_IMPORT:
            v = L.kh_debug_get(k.encode())
            if v is not None and v.decode() == _ENV_AT_IMPORT[k]:
                _ENV_MIRRORED.setdefault(k, _ENV_AT_IMPORT[k])


_ENV_AT_IMPORT = {k: v for k, v in os.environ.items() if k.startswith(_HOOK_PREFIXES)}


def sync_env() -> None:
    """Mirror the KH_* hook variables of os.environ into the library's hook table.

    The library reads the environment once, when it is loaded; afterwards hooks change only through
    kh_debug_set.  The Python binding calls this before every entry point that reads a hook, so
    `os.environ[...] = ...` / `monkeypatch.setenv` keep working in tests and tools.  Only keys that came from
    the environment are managed: a variable that disappears from os.environ is cleared in the table, a hook
    set through debug_set() is left alone (an environment variable of the same name overrides it)."""
    if not _ENV_SEEDED:
        _seed_mirror(lib())
    want = {k: v for k, v in os.environ.items() if k.startswith(_HOOK_PREFIXES)}
    if want == _ENV_MIRRORED:  # nothing changed since the last call: no table walk per generate()
        return
    L = lib()
    for k in set(_ENV_MIRRORED) - set(want):
        L.kh_debug_set(k.encode(), None)
        del _ENV_MIRRORED[k]
    for k, v in want.items():
        if _ENV_MIRRORED.get(k) != v:
            L.kh_debug_set(k.encode(), v.encode())
            _ENV_MIRRORED[k] = v


def error_string(code: int) -> str:
    return lib().kh_error_string(int(code)).decode()


def check(code: int, what: str) -> None:
    if code != 0:
        raise KhError(int(code), what)


def plan_decode_shapes(dim: int, hidden_dim: int, kv_dim: int, vocab_size: int, quant: bool) -> dict:
    """{kernel: {split, u, grid, wg}} of the five decode GEMV kernels for a geometry (host-only)."""
    sync_env()
    out = (_i32 * 20)()
    rc = lib().kh_plan_decode_shapes(dim, hidden_dim, kv_dim, vocab_size, int(quant), out)
    if rc != 0:
        raise KhError(rc, "kh_plan_decode_shapes")
    names = ("qkv", "wo", "ffn13", "w2", "cls")
    return {n: dict(zip(("split", "u", "grid", "wg"), out[4 * i:4 * i + 4])) for i, n in enumerate(names)}


def plan_decode_ring(dim: int, hidden_dim: int, vocab_size: int, quant: bool, group_size: int = 64) -> dict:
    """Which int8 decode GEMVs run on the LDS-DMA ring kernels: {ffn13: {slots, grid}, cls: {slots, grid}};
    slots 0 = the register-tile kernel of plan_decode_shapes (host-only, kh_plan_decode_ring)."""
    sync_env()
    out = (_i32 * 4)()
    rc = lib().kh_plan_decode_ring(dim, hidden_dim, vocab_size, int(quant), group_size, out)
    if rc != 0:
        raise KhError(rc, "kh_plan_decode_ring")
    return {"ffn13": {"slots": out[0], "grid": out[1]}, "cls": {"slots": out[2], "grid": out[3]}}


def plan_attention(head_num: int, kv_mul: int, head_size: int, seq_len: int, pos: int) -> dict:
    """Decode-attention geometry for a cache of seq_len rows at position pos (host-only, kh_plan_attention)."""
    sync_env()
    out = (_i32 * 8)()
    rc = lib().kh_plan_attention(head_num, kv_mul, head_size, seq_len, pos, out)
    if rc != 0:
        raise KhError(rc, "kh_plan_attention")
    return dict(zip(("ns", "ns_g", "stride", "t_long", "group_path", "active_splits", "split_len", "workgroups"),
                    list(out)))


def plan_prefill_shape(epi: str, T: int, rows: int, K: int, quant: bool, r2_ok: bool = True) -> dict:
    """Launch plan of one GEMM of a T-token prefill pass (epi: qkv | resid | swiglu); host-only."""
    sync_env()
    out = (_i32 * 7)()
    rc = lib().kh_plan_prefill_shape({"qkv": 0, "resid": 1, "swiglu": 2}[epi], T, rows, K, int(quant),
                                     int(r2_ok), out)
    if rc != 0:
        raise KhError(rc, "kh_plan_prefill_shape")
    return dict(zip(("R", "NT", "ks", "slices", "solo", "kz", "workgroups"), out[:]))
