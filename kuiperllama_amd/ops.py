"""Operator-level Python mirror of the reference's kernel interface
(kuiper/source/op/kernels/kernels_interface.h:6-68) over libkuiper_hip.so.

Arguments are torch tensors living on the GPU (torch is only the owner of device memory and
streams here); every call goes through the C-ABI and launches a hand-written HIP kernel on
torch's current stream.  There is no eager/CPU fallback: a CPU tensor is an error.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _ffi

ROPE_INTERLEAVED, ROPE_HALF = 0, 1


def _p(t: Optional[torch.Tensor], dtype=None):
    if t is None:
        return None
    if not t.is_cuda:
        raise ValueError("kuiperllama_amd.ops: tensors must be on the GPU (no CPU fallback)")
    if not t.is_contiguous():
        raise ValueError("kuiperllama_amd.ops: tensors must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"expected {dtype}, got {t.dtype}")
    return t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def add(in1, in2, out):
    _ffi.check(_ffi.lib().kh_add_f32(_p(in1, torch.float32), _p(in2, torch.float32),
                                     _p(out, torch.float32), in1.numel(), _stream()), "kh_add_f32")
    return out


def matmul(x, w, out, scale: float = 1.0):
    K, M = w.shape
    _ffi.check(_ffi.lib().kh_matmul_f32(_p(x, torch.float32), _p(w, torch.float32),
                                        _p(out, torch.float32), M, K, scale, _stream()),
               "kh_matmul_f32")
    return out


def matmul_q8(x, w8, scales, group_size: int, out):
    K, M = w8.shape
    _ffi.check(_ffi.lib().kh_matmul_q8(_p(x, torch.float32), _p(w8, torch.int8),
                                       _p(scales, torch.float32), group_size,
                                       _p(out, torch.float32), M, K, _stream()), "kh_matmul_q8")
    return out


def embedding(tokens, w, out):
    V, dim = w.shape
    _ffi.check(_ffi.lib().kh_embedding_f32(_p(tokens, torch.int32), tokens.numel(),
                                           _p(w, torch.float32), _p(out, torch.float32), dim, V,
                                           _stream()), "kh_embedding_f32")
    return out


def embedding_host_tokens(tokens, w, out):
    """EmbeddingKernel as the reference calls it: `tokens` is a HOST int32 array (numpy); the ids
    travel in the kernel arguments, 64 per launch (kh_embedding_f32_host)."""
    import numpy as np
    t = np.ascontiguousarray(tokens, dtype=np.int32)
    _ffi.check(_ffi.lib().kh_embedding_f32_host(t.ctypes.data, t.size, _p(w), _p(out), w.shape[1],
                                                w.shape[0], _stream()), "kh_embedding_f32_host")


def swiglu(a, b, out):
    _ffi.check(_ffi.lib().kh_swiglu_f32(_p(a, torch.float32), _p(b, torch.float32),
                                        _p(out, torch.float32), a.numel(), _stream()),
               "kh_swiglu_f32")
    return out


def rmsnorm(x, w, out, eps: float):
    _ffi.check(_ffi.lib().kh_rmsnorm_f32(_p(x, torch.float32), _p(w, torch.float32),
                                         _p(out, torch.float32), x.numel(), eps, _stream()),
               "kh_rmsnorm_f32")
    return out


def rope(q, k, pos, sin_cache, cos_cache, head_size: int, mode: int):
    """pos: python int, or a 1-element int32 GPU tensor (graph-capturable form)."""
    d_pos, ipos = (None, int(pos)) if not torch.is_tensor(pos) else (_p(pos, torch.int32), 0)
    _ffi.check(_ffi.lib().kh_rope_f32(q.numel(), k.numel(), head_size, _p(q, torch.float32),
                                      _p(k, torch.float32), d_pos, ipos,
                                      _p(sin_cache, torch.float32), _p(cos_cache, torch.float32),
                                      mode, _stream()), "kh_rope_f32")
    return q, k


def sincos_cache(head_size: int, seq_len: int, theta: float, sin_cache, cos_cache):
    _ffi.check(_ffi.lib().kh_sincos_cache_f32(head_size, seq_len, theta,
                                              _p(sin_cache, torch.float32),
                                              _p(cos_cache, torch.float32), _stream()),
               "kh_sincos_cache_f32")
    return sin_cache, cos_cache


def mha(pos, head_num, layer_index, seq_len, kv_dim, kv_mul, head_size, mha_out, q, score,
        kcache, vcache):
    d_pos, ipos = (None, int(pos)) if not torch.is_tensor(pos) else (_p(pos, torch.int32), 0)
    _ffi.check(_ffi.lib().kh_mha_f32(d_pos, ipos, head_num, layer_index, seq_len, kv_dim, kv_mul,
                                     head_size, _p(mha_out, torch.float32), _p(q, torch.float32),
                                     _p(score, torch.float32) if score is not None else None,
                                     _p(kcache, torch.float32),
                                     _p(vcache, torch.float32), _stream()), "kh_mha_f32")
    return mha_out


def mha_prefill(pos0, n_tokens, head_num, layer_index, seq_len, kv_dim, kv_mul, head_size, mha_out, q,
                kcache, vcache):
    """Causal attention of n_tokens consecutive queries (rows of q) on the matrix cores."""
    _ffi.sync_env()  # KH_PG_ATTN_QT
    _ffi.check(_ffi.lib().kh_mha_prefill_f32(
        int(pos0), int(n_tokens), head_num, layer_index, seq_len, kv_dim, kv_mul, head_size,
        _p(mha_out, torch.float32), _p(q, torch.float32), _p(kcache, torch.float32),
        _p(vcache, torch.float32), _stream()), "kh_mha_prefill_f32")
    return mha_out


def mha_decode_workspace(head_num: int, head_size: int, seq_len: int, device):
    """Zeroed workspace tensor for mha_decode (None when no time split is needed)."""
    _ffi.sync_env()  # KH_ATTN_TLONG
    n = int(_ffi.lib().kh_mha_decode_workspace_bytes(head_num, head_size, seq_len))
    if n < 0:
        _ffi.check(n, "kh_mha_decode_workspace_bytes")
    return torch.zeros(n, dtype=torch.uint8, device=device) if n else None


def mha_decode(pos, head_num, layer_index, seq_len, kv_dim, kv_mul, head_size, mha_out, q, kcache,
               vcache, workspace):
    d_pos, ipos = (None, int(pos)) if not torch.is_tensor(pos) else (_p(pos, torch.int32), 0)
    _ffi.check(_ffi.lib().kh_mha_decode_f32(
        d_pos, ipos, head_num, layer_index, seq_len, kv_dim, kv_mul, head_size,
        _p(mha_out, torch.float32), _p(q, torch.float32), _p(kcache, torch.float32),
        _p(vcache, torch.float32), _p(workspace) if workspace is not None else None,
        workspace.numel() if workspace is not None else 0, _stream()), "kh_mha_decode_f32")
    return mha_out


def argmax(logits, out_index):
    _ffi.check(_ffi.lib().kh_argmax_f32(_p(logits, torch.float32), logits.numel(),
                                        _p(out_index, torch.int32), _stream()), "kh_argmax_f32")
    return out_index


def argmax_host(logits) -> int:
    import ctypes as C
    r = C.c_int64(-1)
    _ffi.check(_ffi.lib().kh_argmax_f32_host(_p(logits, torch.float32), logits.numel(),
                                             C.byref(r), _stream()), "kh_argmax_f32_host")
    return int(r.value)


def softmax_(x):
    _ffi.check(_ffi.lib().kh_softmax_f32(_p(x, torch.float32), x.numel(), _stream()),
               "kh_softmax_f32")
    return x


def scale_(scale: float, x):
    _ffi.check(_ffi.lib().kh_scale_f32(scale, _p(x, torch.float32), x.numel(), _stream()),
               "kh_scale_f32")
    return x


def scale_sum(value, scale, out, pos: int, size: int, stride: int):
    _ffi.check(_ffi.lib().kh_scale_sum_f32(_p(value, torch.float32), _p(scale, torch.float32),
                                           _p(out, torch.float32), pos, size, stride, _stream()),
               "kh_scale_sum_f32")
    return out
