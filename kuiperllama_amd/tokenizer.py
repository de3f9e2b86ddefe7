"""SentencePiece-BPE tokenizer on the C-ABI (kh_spm_*): the host-side mirror of the reference's
op::SpeEncodeLayer (kuiper/source/op/encode.cpp:10-57) — encode(sentence) with BOS on for the
Llama family (model.cpp:158-165), decode(ids), is_sentence_ending(id) == eos (encode.cpp:48-51)."""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence

from . import _ffi


class SpmTokenizer:
    def __init__(self, handle):
        self._h = handle

    @classmethod
    def from_file(cls, path: str) -> "SpmTokenizer":
        h = C.c_void_p()
        _ffi.check(_ffi.lib().kh_spm_create_from_file(path.encode(), C.byref(h)),
                   "kh_spm_create_from_file")
        return cls(h)

    @classmethod
    def from_bytes(cls, model_proto: bytes) -> "SpmTokenizer":
        h = C.c_void_p()
        buf = C.create_string_buffer(model_proto, len(model_proto))
        _ffi.check(_ffi.lib().kh_spm_create_from_memory(buf, len(model_proto), C.byref(h)),
                   "kh_spm_create_from_memory")
        return cls(h)

    def close(self) -> None:
        if self._h:
            _ffi.lib().kh_spm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    vocab_size = property(lambda s: int(_ffi.lib().kh_spm_vocab_size(s._h)))
    bos_id = property(lambda s: int(_ffi.lib().kh_spm_bos_id(s._h)))
    eos_id = property(lambda s: int(_ffi.lib().kh_spm_eos_id(s._h)))
    unk_id = property(lambda s: int(_ffi.lib().kh_spm_unk_id(s._h)))

    def is_sentence_ending(self, token_id: int) -> bool:
        return token_id == self.eos_id

    def encode(self, text: str, bos: bool = True, eos: bool = False) -> List[int]:
        raw = text.encode("utf-8")
        cap = 4 * len(raw) + 8
        n = C.c_int32(0)
        while True:
            out = (C.c_int32 * cap)()
            rc = _ffi.lib().kh_spm_encode(self._h, raw, len(raw), int(bos), int(eos), out, cap,
                                          C.byref(n))
            if rc == _ffi.KH_ERR_RANGE:
                cap = n.value
                continue
            _ffi.check(rc, "kh_spm_encode")
            return list(out[: n.value])

    def decode(self, ids: Sequence[int]) -> str:
        arr = (C.c_int32 * max(len(ids), 1))(*[int(i) for i in ids])
        cap = 16 * len(ids) + 16
        ln = C.c_int64(0)
        while True:
            out = C.create_string_buffer(cap)
            rc = _ffi.lib().kh_spm_decode(self._h, arr, len(ids), out, cap, C.byref(ln))
            if rc == _ffi.KH_ERR_RANGE:
                cap = ln.value
                continue
            _ffi.check(rc, "kh_spm_decode")
            return out.raw[: ln.value].decode("utf-8", errors="replace")
