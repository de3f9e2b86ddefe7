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


LLAMA3, QWEN2 = 0, 1
REF_SPACES = 1


class BpeTokenizer:
    """Byte-level BPE over a HuggingFace tokenizer.json on the C-ABI (kh_bpe_*): the host-side mirror
    of op::BpeEncodeLayer / op::QwenEncodeLayer (kuiper/source/op/encode.cpp:59-183).  `ref_spaces`
    (default on, like the reference) replaces every ' ' with 'Ġ' before encoding and back after
    decoding (encode.cpp:108-111, 124-126); off = encode the text as it is (HF-compatible).  BOS is on
    for Llama-3 and off for Qwen2 (model.cpp:158-165)."""

    def __init__(self, handle, flavor: int, ref_spaces: bool = True):
        self._h = handle
        self.flavor = flavor
        self.flags = REF_SPACES if ref_spaces else 0

    @classmethod
    def from_file(cls, path: str, flavor: int = LLAMA3, ref_spaces: bool = True) -> "BpeTokenizer":
        h = C.c_void_p()
        _ffi.check(_ffi.lib().kh_bpe_create_from_file(path.encode(), flavor, C.byref(h)),
                   "kh_bpe_create_from_file")
        return cls(h, flavor, ref_spaces)

    @classmethod
    def from_bytes(cls, tokenizer_json: bytes, flavor: int = LLAMA3, ref_spaces: bool = True) -> "BpeTokenizer":
        h = C.c_void_p()
        buf = C.create_string_buffer(tokenizer_json, len(tokenizer_json))
        _ffi.check(_ffi.lib().kh_bpe_create_from_memory(buf, len(tokenizer_json), flavor, C.byref(h)),
                   "kh_bpe_create_from_memory")
        return cls(h, flavor, ref_spaces)

    def close(self) -> None:
        if self._h:
            _ffi.lib().kh_bpe_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    vocab_size = property(lambda s: int(_ffi.lib().kh_bpe_vocab_size(s._h)))
    bos_id = property(lambda s: int(_ffi.lib().kh_bpe_bos_id(s._h)))
    eos_id = property(lambda s: int(_ffi.lib().kh_bpe_eos_id(s._h)))
    stop_ids = property(lambda s: [int(_ffi.lib().kh_bpe_stop_id(s._h, 0)),
                                   int(_ffi.lib().kh_bpe_stop_id(s._h, 1))])

    def is_sentence_ending(self, token_id: int) -> bool:
        return token_id in self.stop_ids

    def encode(self, text: str, bos: bool | None = None, eos: bool = False) -> List[int]:
        if bos is None:
            bos = self.flavor == LLAMA3
        raw = text.encode("utf-8")
        cap = 2 * len(raw) + 8
        n = C.c_int32(0)
        while True:
            out = (C.c_int32 * cap)()
            rc = _ffi.lib().kh_bpe_encode(self._h, raw, len(raw), int(bos), int(eos), self.flags, out,
                                          cap, C.byref(n))
            if rc == _ffi.KH_ERR_RANGE:
                cap = n.value
                continue
            _ffi.check(rc, "kh_bpe_encode")
            return list(out[: n.value])

    def decode_bytes(self, ids: Sequence[int]) -> bytes:
        arr = (C.c_int32 * max(len(ids), 1))(*[int(i) for i in ids])
        cap = 32 * len(ids) + 16
        ln = C.c_int64(0)
        while True:
            out = C.create_string_buffer(cap)
            rc = _ffi.lib().kh_bpe_decode(self._h, arr, len(ids), self.flags, out, cap, C.byref(ln))
            if rc == _ffi.KH_ERR_RANGE and ln.value > cap:
                cap = ln.value
                continue
            _ffi.check(rc, "kh_bpe_decode")
            return out.raw[: ln.value]

    def decode(self, ids: Sequence[int]) -> str:
        return self.decode_bytes(ids).decode("utf-8", errors="replace")
