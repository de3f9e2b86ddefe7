"""SentencePiece-BPE tokenizer of the C-ABI (kh_spm_*, kh_tokenizer.cpp) against the sentencepiece
library itself — the reference's SpeEncodeLayer is a wrapper over that library (encode.cpp:10-57).
Golden ids/text come from tests/golden/make_spm_golden.py; when the Python package is importable
the same models are also fuzzed live.  Host only: runs without a GPU."""
import json
import os
import random

import pytest

from conftest import GOLDEN


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(GOLDEN, "spm_golden.json")) as f:
        return json.load(f)


def _tok(name):
    from kuiperllama_amd.tokenizer import SpmTokenizer
    return SpmTokenizer.from_file(os.path.join(GOLDEN, f"spm_{name}.model"))


@pytest.mark.parametrize("name", ["llama_like", "plain"])
def test_encode_decode_match_sentencepiece_goldens(golden, name):
    g = golden[name]
    t = _tok(name)
    assert (t.vocab_size, t.bos_id, t.eos_id, t.unk_id) == (g["vocab_size"], g["bos"], g["eos"], g["unk"])
    for c in g["cases"]:
        ids = t.encode(c["text"], bos=False)
        assert ids == c["ids"], (name, c["text"], ids, c["ids"])
        assert t.decode(c["ids"]) == c["decoded"], (name, c["text"])
    # SpeEncodeLayer::encode: BOS in front for the Llama family, EOS optional (encode.cpp:37-44)
    c = g["cases"][3]
    assert t.encode(c["text"]) == [g["bos"]] + c["ids"]
    assert t.encode(c["text"], bos=True, eos=True) == [g["bos"]] + c["ids"] + [g["eos"]]
    assert t.is_sentence_ending(g["eos"]) and not t.is_sentence_ending(g["bos"])
    # control pieces vanish on decode, out-of-range ids are ignored
    assert t.decode([g["bos"]] + c["ids"] + [g["eos"], 10 ** 6]) == c["decoded"]
    t.close()


@pytest.mark.parametrize("name", ["llama_like", "plain"])
def test_fuzz_against_live_sentencepiece(name):
    spm = pytest.importorskip("sentencepiece")
    sp = spm.SentencePieceProcessor(model_file=os.path.join(GOLDEN, f"spm_{name}.model"))
    t = _tok(name)
    r = random.Random(7)
    alphabet = "abcdefghijklmnopqrstuvwxyz  ABC.,!?019 \té中▁\U0001F600"
    for _ in range(400):
        s = "".join(r.choice(alphabet) for _ in range(r.randint(0, 40)))
        want = sp.encode(s)
        assert t.encode(s, bos=False) == want, repr(s)
        assert t.decode(want) == sp.decode(want), repr(s)
    t.close()


def test_rejects_what_it_does_not_implement(tmp_path):
    from kuiperllama_amd import _ffi
    from kuiperllama_amd.tokenizer import SpmTokenizer
    with pytest.raises(Exception):
        SpmTokenizer.from_file(str(tmp_path / "missing.model"))
    bad = tmp_path / "garbage.model"
    bad.write_bytes(b"\xff" * 64)
    with pytest.raises(Exception):
        SpmTokenizer.from_file(str(bad))
    spm = pytest.importorskip("sentencepiece")
    import io
    model = io.BytesIO()
    spm.SentencePieceTrainer.train(sentence_iterator=iter(["hello world foo bar baz"] * 50),
                                   model_writer=model, vocab_size=16, model_type="unigram",
                                   hard_vocab_limit=False, minloglevel=2)
    with pytest.raises(Exception) as e:   # unigram model + NFKC character map: unsupported
        SpmTokenizer.from_bytes(model.getvalue())
    assert "unsupported" in str(e.value).lower() or str(_ffi.KH_ERR_RANGE) not in str(e.value)
