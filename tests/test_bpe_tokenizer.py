"""Byte-level BPE tokenizer of the C-ABI (kh_bpe_*, kh_bpe.cpp) — the reference's BpeEncodeLayer /
QwenEncodeLayer (kuiper/source/op/encode.cpp:59-183: nlohmann::json + tiktoken.h + RE2) — against the
HuggingFace `tokenizers` package on two tokenizer.json files trained in-container with the
reference's split pattern (tests/golden/make_bpe_golden.py).  Both modes are pinned: the text as it
is ("plain" = HF behaviour) and the reference's " " -> "Ġ" pre-replacement ("ref",
encode.cpp:108-111).  Host only: runs without a GPU."""
import json
import os
import random

import pytest

from conftest import GOLDEN

FLAVOR = {"llama3_like": 0, "qwen2_like": 1}


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(GOLDEN, "bpe_golden.json"), encoding="utf-8") as f:
        return json.load(f)


def _tok(name, ref_spaces):
    from kuiperllama_amd.tokenizer import BpeTokenizer
    return BpeTokenizer.from_file(os.path.join(GOLDEN, f"bpe_{name}.json"), FLAVOR[name], ref_spaces)


@pytest.mark.parametrize("name", sorted(FLAVOR))
def test_encode_decode_match_hf_tokenizers_goldens(golden, name):
    g = golden[name]
    plain, ref = _tok(name, False), _tok(name, True)
    assert plain.vocab_size == g["vocab_size"]  # |model.vocab| + |added_tokens| (encode.cpp:105)
    for c in g["cases"]:
        got = plain.encode(c["text"], bos=False)
        assert got == c["plain"], (name, "plain", c["text"], got, c["plain"])
        got = ref.encode(c["text"], bos=False)
        assert got == c["ref"], (name, "ref", c["text"], got, c["ref"])
        assert plain.decode(c["plain"]) == c["plain_decoded"], (name, c["text"])
        assert ref.decode(c["ref"]) == c["ref_decoded"], (name, c["text"])
        # byte-level BPE is lossless: decode(encode(text)) == text; in the reference's mode a
        # literal "Ġ" of the input comes back as a space (encode.cpp:124-126)
        assert plain.decode(c["plain"]) == c["text"]
        assert ref.decode(c["ref"]) == c["text"].replace("Ġ", " ")


def test_special_ids_bos_policy_and_stop_tokens(golden):
    sp = golden["llama3_like"]["special"]
    t = _tok("llama3_like", True)
    # encode.cpp:97-103: bos <|begin_of_text|>, eos <|end_of_text|>, stop ids {eos, <|eot_id|>}
    assert (t.bos_id, t.eos_id) == (sp["<|begin_of_text|>"], sp["<|end_of_text|>"])
    assert t.stop_ids == [sp["<|end_of_text|>"], sp["<|eot_id|>"]]
    assert t.is_sentence_ending(sp["<|eot_id|>"]) and not t.is_sentence_ending(sp["<|begin_of_text|>"])
    c = golden["llama3_like"]["cases"][3]
    assert t.encode(c["text"]) == [t.bos_id] + c["ref"]  # Llama: BOS on (model.cpp:158-165)
    assert t.encode(c["text"], bos=True, eos=True) == [t.bos_id] + c["ref"] + [t.eos_id]
    sq = golden["qwen2_like"]["special"]
    q = _tok("qwen2_like", True)
    # encode.cpp:170-176: bos <|im_start|>, eos <|im_end|>, stop ids {eos, <|endoftext|>}
    assert (q.bos_id, q.eos_id) == (sq["<|im_start|>"], sq["<|im_end|>"])
    assert q.stop_ids == [sq["<|im_end|>"], sq["<|endoftext|>"]]
    cq = golden["qwen2_like"]["cases"][3]
    assert q.encode(cq["text"]) == cq["ref"]  # Qwen: no BOS
    # special tokens inside the text become their ids (tiktoken.h:215-246, all specials allowed)
    ids = t.encode("<|begin_of_text|>hi<|eot_id|>", bos=False)
    assert ids[0] == sp["<|begin_of_text|>"] and ids[-1] == sp["<|eot_id|>"]
    assert t.decode(ids) == "<|begin_of_text|>hi<|eot_id|>"


def test_error_paths(tmp_path):
    from kuiperllama_amd import _ffi
    from kuiperllama_amd.tokenizer import BpeTokenizer
    with pytest.raises(_ffi.KhError) as ei:
        BpeTokenizer.from_file(str(tmp_path / "missing.json"))
    assert ei.value.code == -3
    with pytest.raises(_ffi.KhError) as ei:
        BpeTokenizer.from_bytes(b'{"model": {"type": "BPE"}}')  # no vocab
    assert ei.value.code == -4
    with pytest.raises(_ffi.KhError) as ei:  # not a byte-level vocabulary (U+2581 is no GPT-2 byte symbol)
        BpeTokenizer.from_bytes('{"added_tokens": [], "model": {"vocab": {"▁a": 0}}}'.encode())
    assert ei.value.code == -2
    t = _tok("llama3_like", True)
    with pytest.raises(_ffi.KhError):
        t.decode([10 ** 7])  # tiktoken.h:262: unknown token


ALPHABET = ("abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789 .,;:!?'\"()[]{}+-*/=%&#@_"
            "\n\t\r éñüßøçÄ 東京中文 данные αβγ ☃🙂 Ġ")


@pytest.mark.parametrize("name", sorted(FLAVOR))
def test_fuzz_against_hf_tokenizers(name):
    tokenizers = pytest.importorskip("tokenizers")
    hf = tokenizers.Tokenizer.from_file(os.path.join(GOLDEN, f"bpe_{name}.json"))
    plain, ref = _tok(name, False), _tok(name, True)
    r = random.Random(1234 + len(name))
    words = ["the", "quick", "it's", "DON'T", "we'RE", "2024", "3.14", "café", "東京", "  ", "\n\n", "(ok)",
             "<|eot_id|>", "<|im_end|>", "<|endoftext|>", "<|begin_of_text|>", " ", "\t", "x'll", "'d"]
    for k in range(400):
        if k % 2:
            s = "".join(r.choice(ALPHABET) for _ in range(r.randint(0, 40)))
        else:
            s = "".join(r.choice(words) + r.choice(["", " ", "  ", "\n"]) for _ in range(r.randint(1, 9)))
        want = hf.encode(s, add_special_tokens=False).ids
        got = plain.encode(s, bos=False)
        assert got == want, (name, "plain", repr(s), got, want)
        want = hf.encode(s.replace(" ", "Ġ"), add_special_tokens=False).ids
        got = ref.encode(s, bos=False)
        assert got == want, (name, "ref", repr(s), got, want)
        assert plain.decode(plain.encode(s, bos=False)) == s
        assert ref.decode(got) == s.replace("Ġ", " ")


def test_truncated_tokenizer_json_is_rejected_not_overread():
    """kh_bpe_create_from_memory takes (ptr, nbytes) - not a NUL-terminated string.  A file cut
    anywhere (inside a number, right behind a ':', inside a string) must come back as an error;
    the integer parser is bounded by the buffer end (strtol could read past it)."""
    import numpy as np
    from kuiperllama_amd import _ffi
    from kuiperllama_amd.tokenizer import BpeTokenizer, LLAMA3
    data = open(os.path.join(GOLDEN, "bpe_llama3_like.json"), "rb").read()
    assert BpeTokenizer.from_bytes(data, LLAMA3).vocab_size > 0
    rng = np.random.default_rng(0)
    cuts = sorted(set(int(c) for c in rng.integers(1, len(data) - 1, 150)))
    # plus cuts that end exactly inside / right before a vocabulary id
    import re
    for m in list(re.finditer(rb'": \d+', data))[:40]:
        cuts += [m.start() + 3, m.start() + 4, m.end() - 1]
    for c in cuts:
        with pytest.raises(_ffi.KhError):
            BpeTokenizer.from_bytes(data[:c], LLAMA3)
