"""ctypes binding of oracle/libkuiper_oracle.so — the CPU restatement of the reference's CPU
decode path.

TEST INFRASTRUCTURE ONLY: import this from tests/, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` — never from ``kuiperllama_amd``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libkuiper_oracle.so")

ROPE_INTERLEAVED, ROPE_HALF = 0, 1
ACC_F32, ACC_F64 = 0, 1
FAMILY_LLAMA, FAMILY_QWEN2 = 0, 1


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "kuiper_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or (
            os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(_LIB_PATH)):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


class _Config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "dim", "hidden_dim", "layer_num", "head_num", "kv_head_num", "vocab_size", "seq_len",
        "kv_dim", "kv_mul", "head_size", "is_shared_weight", "is_quant", "group_size",
        "family", "rope_mode")] + [("rope_theta", C.c_float), ("rms_eps", C.c_float),
                                   ("cache_len", C.c_int32)]


_lib: Optional[C.CDLL] = None
_fp = C.POINTER(C.c_float)
_i8p = C.POINTER(C.c_int8)
_i32p = C.POINTER(C.c_int32)


def effective_cpus() -> int:
    """CPUs this process may really use: min(sched affinity, cgroup cpu.max quota).  A container
    can see 256 logical CPUs in nproc while its cgroup grants a handful; OpenMP teams sized by
    nproc then spin against each other (observed: 13 s/token instead of ~0.1 s)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.999)))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, (q + per - 1) // per))
        except Exception:
            pass
    return max(1, n)


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        # idle OpenMP workers must sleep, not spin, when cores are shared or quota-limited
        os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
        os.environ.setdefault("OMP_PROC_BIND", "false")
        build()
        L = C.CDLL(_LIB_PATH)
        L.ko_set_threads.argtypes = [C.c_int]
        L.ko_get_threads.restype = C.c_int
        L.ko_matmul_f32.argtypes = [_fp, _fp, _fp, C.c_int, C.c_int, C.c_float, C.c_int]
        L.ko_matmul_q8.argtypes = [_fp, _i8p, _fp, C.c_int, _fp, C.c_int, C.c_int, C.c_int]
        L.ko_rmsnorm_f32.argtypes = [_fp, _fp, _fp, C.c_int, C.c_float]
        L.ko_sincos_cache.argtypes = [C.c_int, C.c_int, C.c_float, _fp, _fp]
        L.ko_rope_f32.argtypes = [C.c_int, C.c_int, C.c_int, _fp, _fp, C.c_int, _fp, _fp, C.c_int]
        L.ko_softmax_f32.argtypes = [_fp, C.c_int]
        L.ko_scale_sum_f32.argtypes = [_fp, _fp, _fp, C.c_int, C.c_int, C.c_int]
        L.ko_scale_f32.argtypes = [C.c_float, _fp, C.c_int]
        L.ko_mha_f32.argtypes = [C.c_int] * 7 + [_fp, _fp, _fp, _fp, _fp, C.c_int]
        L.ko_swiglu_f32.argtypes = [_fp, _fp, _fp, C.c_int]
        L.ko_add_f32.argtypes = [_fp, _fp, _fp, C.c_int]
        L.ko_embedding_f32.argtypes = [_i32p, C.c_int, _fp, _fp, C.c_int, C.c_int]
        L.ko_embedding_f32.restype = C.c_int
        L.ko_argmax_f32.argtypes = [_fp, C.c_size_t]
        L.ko_argmax_f32.restype = C.c_size_t
        L.ko_quantize_q80.argtypes = [_fp, C.c_size_t, C.c_int, _i8p, _fp]
        L.ko_model_create.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int,
                                      C.c_float, C.c_float, C.c_int]
        L.ko_model_create.restype = C.c_void_p
        L.ko_model_destroy.argtypes = [C.c_void_p]
        L.ko_model_config.argtypes = [C.c_void_p]
        L.ko_model_config.restype = C.POINTER(_Config)
        L.ko_model_expected_bytes.argtypes = [C.c_void_p]
        L.ko_model_expected_bytes.restype = C.c_size_t
        L.ko_model_forward.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int]
        L.ko_model_forward.restype = C.c_int
        L.ko_model_logits.argtypes = [C.c_void_p]
        L.ko_model_logits.restype = _fp
        L.ko_model_kcache.argtypes = [C.c_void_p]
        L.ko_model_kcache.restype = _fp
        L.ko_model_vcache.argtypes = [C.c_void_p]
        L.ko_model_vcache.restype = _fp
        L.ko_model_generate.argtypes = [C.c_void_p, _i32p, C.c_int, C.c_int, _i32p, C.c_int]
        L.ko_model_generate.restype = C.c_int
        L.ko_model_generate_until.argtypes = [C.c_void_p, _i32p, C.c_int, C.c_int, _i32p, C.c_int,
                                              _i32p, C.c_int]
        L.ko_model_generate_until.restype = C.c_int
        _lib = L
    return _lib


def _f(a: np.ndarray):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_fp)


def f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


_blas = None


def use_openblas(threads: int) -> bool:
    """Timing only (bench.py): route the oracle's fp32 matmuls through the OpenBLAS that ships
    inside numpy (ILP64 `scipy_cblas_sgemv64_`), mirroring the reference CPU backend's
    Armadillo -> BLAS sgemv (cpu/matmul_kernel.cpp:37-40).  threads <= 0 switches it off.
    Returns False when no such library is present."""
    global _blas
    L = lib()
    L.ko_set_sgemv.argtypes = [C.c_void_p]
    if threads <= 0:
        L.ko_set_sgemv(None)
        return True
    if _blas is None:
        import glob
        cands = glob.glob(os.path.join(os.path.dirname(np.__file__), "..", "numpy.libs",
                                       "libscipy_openblas64_*.so"))
        if not cands:
            return False
        _blas = C.CDLL(cands[0])
    try:
        fn = C.cast(_blas.scipy_cblas_sgemv64_, C.c_void_p)
        _blas.scipy_openblas_set_num_threads64_.argtypes = [C.c_int]
        _blas.scipy_openblas_set_num_threads64_(int(threads))
    except AttributeError:
        return False
    L.ko_set_sgemv(fn)
    return True


def set_threads(n: int) -> None:
    lib().ko_set_threads(int(n))


def get_threads() -> int:
    return int(lib().ko_get_threads())


# ---- op-level wrappers (numpy in / numpy out) ------------------------------------------------
def matmul(x, w, scale: float = 1.0, acc: int = ACC_F32) -> np.ndarray:
    x, w = f32(x), f32(w)
    K, M = w.shape
    y = np.empty(K, np.float32)
    lib().ko_matmul_f32(_f(x), _f(w), _f(y), M, K, scale, acc)
    return y


def matmul_q8(x, w8, scales, group: int, acc: int = ACC_F32) -> np.ndarray:
    x, scales = f32(x), f32(scales)
    w8 = np.ascontiguousarray(w8, dtype=np.int8)
    K, M = w8.shape
    y = np.empty(K, np.float32)
    lib().ko_matmul_q8(_f(x), w8.ctypes.data_as(_i8p), _f(scales), group, _f(y), M, K, acc)
    return y


def rmsnorm(x, w, eps: float) -> np.ndarray:
    x, w = f32(x), f32(w)
    out = np.empty_like(x)
    lib().ko_rmsnorm_f32(_f(x), _f(w), _f(out), x.size, eps)
    return out


def sincos_cache(head_size: int, seq_len: int, theta: float):
    s = np.empty((seq_len, head_size), np.float32)
    c = np.empty((seq_len, head_size), np.float32)
    lib().ko_sincos_cache(head_size, seq_len, theta, _f(s), _f(c))
    return s, c


def rope(q, k, pos: int, sin_cache, cos_cache, head_size: int, mode: int):
    q, k = f32(q).copy(), f32(k).copy()
    lib().ko_rope_f32(q.size, k.size, head_size, _f(q), _f(k), pos, _f(f32(sin_cache)),
                      _f(f32(cos_cache)), mode)
    return q, k


def softmax(x) -> np.ndarray:
    x = f32(x).copy()
    lib().ko_softmax_f32(_f(x), x.size)
    return x


def mha(pos, head_num, layer, seq_len, kv_dim, kv_mul, head_size, q, kcache, vcache,
        acc: int = ACC_F32):
    q, kcache, vcache = f32(q), f32(kcache), f32(vcache)
    out = np.zeros(head_num * head_size, np.float32)
    score = np.zeros((head_num, seq_len), np.float32)
    lib().ko_mha_f32(pos, head_num, layer, seq_len, kv_dim, kv_mul, head_size, _f(out), _f(q),
                     _f(score), _f(kcache), _f(vcache), acc)
    return out, score


def swiglu(a, b) -> np.ndarray:
    a, b = f32(a), f32(b)
    out = np.empty_like(a)
    lib().ko_swiglu_f32(_f(a), _f(b), _f(out), a.size)
    return out


def add(a, b) -> np.ndarray:
    a, b = f32(a), f32(b)
    out = np.empty_like(a)
    lib().ko_add_f32(_f(a), _f(b), _f(out), a.size)
    return out


def embedding(tokens, w) -> np.ndarray:
    tokens = np.ascontiguousarray(tokens, dtype=np.int32)
    w = f32(w)
    out = np.empty((tokens.size, w.shape[1]), np.float32)
    rc = lib().ko_embedding_f32(tokens.ctypes.data_as(_i32p), tokens.size, _f(w), _f(out),
                                w.shape[1], w.shape[0])
    if rc:
        raise IndexError("token out of range")
    return out


def argmax(logits) -> int:
    logits = f32(logits)
    return int(lib().ko_argmax_f32(_f(logits), logits.size))


def quantize_q80(w, group: int):
    w = f32(w).reshape(-1)
    q = np.empty(w.size, np.int8)
    s = np.empty(w.size // group, np.float32)
    lib().ko_quantize_q80(_f(w), w.size, group, q.ctypes.data_as(_i8p), _f(s))
    return q, s


class OracleModel:
    """CPU model over a ``.bin`` image held in a numpy uint8 array (kept alive here)."""

    def __init__(self, image: np.ndarray, *, family: int = FAMILY_LLAMA, quant: bool = False,
                 rope_mode: int = ROPE_INTERLEAVED, rope_theta: float = 10000.0,
                 rms_eps: float = 1e-5, cache_len: int = 0):
        assert image.dtype == np.uint8 and image.flags["C_CONTIGUOUS"]
        self._image = image
        self._h = lib().ko_model_create(image.ctypes.data, image.size, family, int(quant),
                                        rope_mode, rope_theta, rms_eps, cache_len)
        if not self._h:
            raise ValueError("oracle: malformed or unsupported .bin image")
        self.cfg = lib().ko_model_config(self._h).contents

    @classmethod
    def from_spec(cls, image: np.ndarray, spec, cache_len: int = 0) -> "OracleModel":
        return cls(image, family=spec.family, quant=spec.quant, rope_mode=spec.rope_mode,
                   rope_theta=spec.rope_theta, rms_eps=spec.rms_eps, cache_len=cache_len)

    def close(self):
        if getattr(self, "_h", None):
            lib().ko_model_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown: module globals may already be gone
            pass

    def expected_bytes(self) -> int:
        return int(lib().ko_model_expected_bytes(self._h))

    def forward(self, token: int, pos: int, acc: int = ACC_F32) -> np.ndarray:
        rc = lib().ko_model_forward(self._h, token, pos, acc)
        if rc:
            raise ValueError("oracle forward: token/pos out of range")
        return self.logits()

    def logits(self) -> np.ndarray:
        p = lib().ko_model_logits(self._h)
        return np.ctypeslib.as_array(p, shape=(self.cfg.vocab_size,)).copy()

    def kv_cache(self):
        c = self.cfg
        shp = (c.layer_num, c.cache_len, c.kv_dim)
        k = np.ctypeslib.as_array(lib().ko_model_kcache(self._h), shape=shp)
        v = np.ctypeslib.as_array(lib().ko_model_vcache(self._h), shape=shp)
        return k, v

    def generate(self, prompt: Sequence[int], total_steps: int, acc: int = ACC_F32,
                 stop: Sequence[int] = ()):
        pr = np.ascontiguousarray(prompt, dtype=np.int32)
        st = np.ascontiguousarray(list(stop) or [0], dtype=np.int32)
        out = np.empty(total_steps, np.int32)
        n = lib().ko_model_generate_until(self._h, pr.ctypes.data_as(_i32p), pr.size, total_steps,
                                          st.ctypes.data_as(_i32p), len(list(stop)),
                                          out.ctypes.data_as(_i32p), acc)
        if n < 0:
            raise ValueError("oracle generate failed")
        return out[:n].tolist()
