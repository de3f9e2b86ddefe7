"""Model-level Python mirror of model::LLama2Model / Qwen2Model for the decode path
(kuiper/include/model/model.h:20-56, kuiper/source/model/llama3.cpp:107-167, 642-650,
733-745; demo/main.cpp:5-47) over libkuiper_hip.so.  All compute is in the HIP library.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _ffi, binfmt
from ._ffi import KH_EXEC_FUSED, KH_EXEC_GRAPH, KH_EXEC_UNFUSED  # noqa: F401

EXEC = {"graph": KH_EXEC_GRAPH, "fused": KH_EXEC_FUSED, "unfused": KH_EXEC_UNFUSED}



def _opts(spec: binfmt.ModelSpec, max_seq_len: int, device: int, flags: int = 0) -> _ffi.ModelOpts:
    return _ffi.ModelOpts(spec.family, int(spec.quant), spec.rope_mode, spec.rope_theta,
                          spec.rms_eps, max_seq_len, device, flags)


class KuiperModel:
    """Owns a kh_model handle.  Construct with one of the from_* classmethods."""

    def __init__(self, handle: int, spec: binfmt.ModelSpec, keepalive=None):
        self._h = C.c_void_p(handle)
        self.spec = spec
        self._keep = keepalive
        cfg = _ffi.Config()
        _ffi.check(_ffi.lib().kh_model_get_config(self._h, C.byref(cfg)), "kh_model_get_config")
        self.cfg = cfg

    # ---- construction ------------------------------------------------------------------
    @classmethod
    def from_file(cls, path: str, spec: binfmt.ModelSpec, max_seq_len: int = 0,
                  device: int = 0, flags: int = 0) -> "KuiperModel":
        _ffi.sync_env()
        h = C.c_void_p()
        o = _opts(spec, max_seq_len, device, flags)
        _ffi.check(_ffi.lib().kh_model_create_from_file(path.encode(), C.byref(o), C.byref(h)),
                   "kh_model_create_from_file")
        return cls(h.value, spec)

    @classmethod
    def from_host_image(cls, image: np.ndarray, spec: binfmt.ModelSpec, max_seq_len: int = 0,
                        device: int = 0, flags: int = 0) -> "KuiperModel":
        assert image.dtype == np.uint8 and image.flags["C_CONTIGUOUS"]
        _ffi.sync_env()
        h = C.c_void_p()
        o = _opts(spec, max_seq_len, device, flags)
        _ffi.check(_ffi.lib().kh_model_create_from_host_image(image.ctypes.data, image.size,
                                                              C.byref(o), C.byref(h)),
                   "kh_model_create_from_host_image")
        return cls(h.value, spec)

    @classmethod
    def from_device_image(cls, image: torch.Tensor, spec: binfmt.ModelSpec, max_seq_len: int = 0,
                          device: int = 0, flags: int = 0) -> "KuiperModel":
        """image: uint8 GPU tensor with the .bin bytes (header included), resident on `device`.
        The library needs the bytes after the 28 / 32-byte header at a 16-byte aligned address:
        they are used in place when they already are, otherwise copied once into an aligned
        tensor (which transiently doubles the weight memory).  Not owned by the library."""
        assert image.is_cuda and image.dtype == torch.uint8
        if image.device.index != device:
            raise ValueError(f"image lives on cuda:{image.device.index} but the model is created on "
                             f"device {device}: kernels would dereference another GPU's pointers")
        hb = spec.header_bytes()
        header = np.frombuffer(image[:32].cpu().numpy().tobytes(), dtype=np.int32).copy()
        if (image.data_ptr() + hb) % 16 == 0:
            weights = image[hb:]
            keep = image
        else:
            weights = torch.empty(image.numel() - hb, dtype=torch.uint8, device=image.device)
            weights.copy_(image[hb:])
            keep = weights
        torch.cuda.synchronize()
        _ffi.sync_env()
        h = C.c_void_p()
        o = _opts(spec, max_seq_len, device, flags)
        hdr = (C.c_int32 * 8)(*header.tolist()[:8])
        _ffi.check(_ffi.lib().kh_model_create_from_device_weights(hdr, weights.data_ptr(),
                                                                  weights.numel(), C.byref(o),
                                                                  C.byref(h)),
                   "kh_model_create_from_device_weights")
        return cls(h.value, spec, keepalive=keep)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            _ffi.lib().kh_model_destroy(self._h)
            self._h = C.c_void_p()
        self._keep = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- Model::predict / forward ----------------------------------------------------------
    def predict(self, token: int, pos: int, is_prompt: bool = False, exec: str = "fused") -> int:
        nxt = C.c_int32(-1)
        _ffi.check(_ffi.lib().kh_model_predict(self._h, token, pos, int(is_prompt), EXEC[exec],
                                               C.byref(nxt)), "kh_model_predict")
        return int(nxt.value)

    def logits(self) -> np.ndarray:
        out = np.empty(self.cfg.vocab_size, np.float32)
        _ffi.check(_ffi.lib().kh_model_get_logits(self._h, out.ctypes.data), "kh_model_get_logits")
        return out

    def kv_bytes(self) -> Tuple[int, int]:
        """(reserved, committed) bytes of the KV cache: the address range of the reference's up-front allocation and
        the HBM backing it right now (mapped on demand, kh_model_kv_bytes)."""
        r, c = C.c_int64(0), C.c_int64(0)
        _ffi.check(_ffi.lib().kh_model_kv_bytes(self._h, C.byref(r), C.byref(c)), "kh_model_kv_bytes")
        return int(r.value), int(c.value)

    def kv_cache_ptrs(self) -> Tuple[int, int]:
        k, v = C.c_void_p(), C.c_void_p()
        _ffi.check(_ffi.lib().kh_model_get_kv(self._h, C.byref(k), C.byref(v)), "kh_model_get_kv")
        return k.value, v.value

    def read_kv(self, layer: int, row0: int, nrows: int):
        k = np.empty((nrows, self.cfg.kv_dim), np.float32)
        v = np.empty((nrows, self.cfg.kv_dim), np.float32)
        _ffi.check(_ffi.lib().kh_model_read_kv(self._h, layer, row0, nrows, k.ctypes.data,
                                               v.ctypes.data), "kh_model_read_kv")
        return k, v

    def write_kv(self, layer: int, row0: int, k: np.ndarray, v: np.ndarray) -> None:
        """Overwrite cache rows [row0, row0 + len(k)) of `layer` (rotated keys / raw values)."""
        k = np.ascontiguousarray(k, np.float32)
        v = np.ascontiguousarray(v, np.float32)
        assert k.shape == v.shape and k.ndim == 2 and k.shape[1] == self.cfg.kv_dim
        _ffi.check(_ffi.lib().kh_model_write_kv(self._h, layer, row0, k.shape[0], k.ctypes.data,
                                                v.ctypes.data), "kh_model_write_kv")

    def write_kv_device(self, layer: int, row0: int, k: torch.Tensor, v: torch.Tensor) -> None:
        """write_kv from contiguous fp32 GPU tensors [nrows, kv_dim] on the model's device."""
        assert k.is_cuda and v.is_cuda and k.dtype == v.dtype == torch.float32
        assert k.is_contiguous() and v.is_contiguous() and k.shape == v.shape
        assert k.dim() == 2 and k.shape[1] == self.cfg.kv_dim
        torch.cuda.current_stream(k.device).synchronize()  # the copy runs on the model's own stream
        _ffi.check(_ffi.lib().kh_model_write_kv(self._h, layer, row0, k.shape[0], k.data_ptr(),
                                                v.data_ptr()), "kh_model_write_kv")

    # ---- demo/main.cpp generate() ---------------------------------------------------------
    def generate(self, prompt: Sequence[int], total_steps: int, exec: str = "graph",
                 stop: Optional[Sequence[int]] = None) -> Tuple[List[int], float]:
        """Returns (words, elapsed_ms of the step loop measured with HIP events).  `stop` = the
        reference's sentence-ending token ids (demo/main.cpp:30-32); the stop token itself is not
        part of `words`."""
        _ffi.sync_env()  # KH_PREFILL, KH_PG_*
        pr = (C.c_int32 * len(prompt))(*[int(t) for t in prompt])
        st = list(stop or [])
        sp = (C.c_int32 * max(len(st), 1))(*[int(t) for t in st])
        words = (C.c_int32 * total_steps)()
        n = C.c_int32(0)
        ms = C.c_float(0.0)
        _ffi.check(_ffi.lib().kh_model_generate_until(self._h, pr, len(prompt), total_steps,
                                                      EXEC[exec], sp, len(st), words,
                                                      C.byref(n), C.byref(ms)),
                   "kh_model_generate_until")
        return list(words[: n.value]), float(ms.value)

    def first_sample(self) -> Optional[dict]:
        """Near-tie report of the last generate() whose prompt ran as a prefill: the two largest logits of its
        first sampled step (kh_model_first_sample).  None when that generate had no prefill phase."""
        fs = _ffi.FirstSample()
        rc = _ffi.lib().kh_model_first_sample(self._h, C.byref(fs))
        if rc == -2:  # KH_ERR_UNSUPPORTED: no prefill phase
            return None
        _ffi.check(rc, "kh_model_first_sample")
        return {"pos": fs.pos, "prefill_mode": {1: "gemv", 2: "gemm"}.get(fs.prefill_mode, str(fs.prefill_mode)),
                "top1_id": fs.top1_id, "top2_id": fs.top2_id, "top1": float(fs.top1), "top2": float(fs.top2),
                "margin": float(fs.top1) - float(fs.top2)}

    def prefill(self, tokens: Sequence[int], pos0: int = 0) -> None:
        """Forward of `tokens` at positions pos0.. without logits, 8 (fp32) / 4 (int8) tokens per
        weight pass on the VALU; the K/V rows are bit-identical to token-by-token
        predict(is_prompt=True)."""
        t = (C.c_int32 * len(tokens))(*[int(x) for x in tokens])
        _ffi.check(_ffi.lib().kh_model_prefill(self._h, t, len(tokens), pos0), "kh_model_prefill")
        torch.cuda.synchronize()

    def prefill_gemm(self, tokens: Sequence[int], pos0: int = 0) -> None:
        """Forward of `tokens` at positions pos0.. as fp32-MFMA GEMMs (up to 128 tokens per weight
        pass); K/V rows equal the token-by-token ones to fp32 round-off."""
        _ffi.sync_env()
        t = (C.c_int32 * len(tokens))(*[int(x) for x in tokens])
        _ffi.check(_ffi.lib().kh_model_prefill_gemm(self._h, t, len(tokens), pos0),
                   "kh_model_prefill_gemm")
        torch.cuda.synchronize()

    PREFILL_MODES = {"token": 0, "gemv": 1, "gemm": 2}

    def time_prefill(self, tokens: Sequence[int], pos0: int = 0, mode: str = "gemm") -> float:
        """Milliseconds (HIP events on the model stream) of the prompt phase alone for `tokens`:
        "token" = one forward pass per token (the reference), "gemv" = B-token VALU kernels,
        "gemm" = fp32-MFMA GEMM prefill."""
        _ffi.sync_env()
        t = (C.c_int32 * len(tokens))(*[int(x) for x in tokens])
        ms = C.c_float(0.0)
        _ffi.check(_ffi.lib().kh_model_time_prefill(self._h, t, len(tokens), pos0,
                                                    self.PREFILL_MODES[mode], C.byref(ms)),
                   "kh_model_time_prefill")
        return float(ms.value)

    def time_step(self, pos: int, reps: int = 9) -> List[float]:
        """Microseconds of one graph-replayed decode step at `pos`, `reps` samples."""
        us = (C.c_float * reps)()
        _ffi.check(_ffi.lib().kh_model_time_step(self._h, pos, reps, us), "kh_model_time_step")
        return [float(v) for v in us]

    def profile_kernels(self, pos: int, reps: int = 8):
        """Back-to-back average launch duration (us) of every kernel class at position `pos`
        (kh_model_profile_kernel); clobbers activations and KV row `pos`."""
        out = {}
        for i in range(_ffi.KH_NUM_KCLASS):
            us = C.c_float(0.0)
            _ffi.check(_ffi.lib().kh_model_profile_kernel(self._h, i, pos, reps, C.byref(us)),
                       "kh_model_profile_kernel")
            out[_ffi.lib().kh_kclass_name(i).decode()] = float(us.value)
        return out

    def profile_kernel(self, name: str, pos: int, reps: int = 8) -> float:
        """Back-to-back average launch duration (us) of ONE kernel class ("attn", "ffn13", ...)."""
        names = [_ffi.lib().kh_kclass_name(i).decode() for i in range(_ffi.KH_NUM_KCLASS)]
        us = C.c_float(0.0)
        _ffi.check(_ffi.lib().kh_model_profile_kernel(self._h, names.index(name), pos, reps,
                                                      C.byref(us)), "kh_model_profile_kernel")
        return float(us.value)

    def profile_step(self, start_pos: int, n_steps: int):
        """Per-kernel-class average launch duration (us) of the fused step, HIP events."""
        avg = (C.c_float * _ffi.KH_NUM_KCLASS)()
        cnt = (C.c_int32 * _ffi.KH_NUM_KCLASS)()
        _ffi.check(_ffi.lib().kh_model_profile_step(self._h, start_pos, n_steps, avg, cnt),
                   "kh_model_profile_step")
        names = [_ffi.lib().kh_kclass_name(i).decode() for i in range(_ffi.KH_NUM_KCLASS)]
        return {n: {"avg_us": float(avg[i]), "launches_per_step": int(cnt[i])}
                for i, n in enumerate(names)}

    @property
    def load_ms(self) -> float:
        """Host image -> HBM upload time of the loader (pinned double-buffered chunks)."""
        return float(_ffi.lib().kh_model_get_load_ms(self._h))

    @property
    def stream(self) -> int:
        return int(_ffi.lib().kh_model_stream(self._h) or 0)
