#!/usr/bin/env python3
"""Golden fixtures for the SentencePiece-BPE tokenizer (kh_tokenizer.cpp).

The reference tokenises through the external sentencepiece library (encode.cpp:10-57); no
tokenizer.model ships with it and none can be downloaded here, so two small BPE models are trained
in-container with the sentencepiece Python package (the same library, so its encoder is the
oracle), with Llama-2's normaliser settings (identity map, dummy prefix, byte fallback) and with a
second setting (no dummy prefix, extra whitespace removed, no byte fallback -> <unk>).
Writes tests/golden/spm_*.model and tests/golden/spm_golden.json (ids and decoded text).
"""
import io
import json
import os
import random

import sentencepiece as spm

HERE = os.path.dirname(os.path.abspath(__file__))
WORDS = ("the quick brown fox jumps over lazy dog once upon a time there was little girl who "
         "lived in village near forest she liked to play with her friends and read stories about "
         "dragons castles kings queens 2024 3.14 hello world token model cache rope attention").split()


def corpus(seed, n=600):
    r = random.Random(seed)
    out = []
    for _ in range(n):
        k = r.randint(3, 14)
        s = " ".join(r.choice(WORDS) for _ in range(k))
        if r.random() < 0.3:
            s = s.capitalize() + r.choice(".!?")
        out.append(s)
    return out


def train(name, **kw):
    model = io.BytesIO()
    spm.SentencePieceTrainer.train(sentence_iterator=iter(corpus(len(name))), model_writer=model,
                                   model_type="bpe", normalization_rule_name="identity",
                                   character_coverage=1.0, unk_id=0, bos_id=1, eos_id=2, pad_id=-1,
                                   minloglevel=2, **kw)
    with open(os.path.join(HERE, f"spm_{name}.model"), "wb") as f:
        f.write(model.getvalue())
    return spm.SentencePieceProcessor(model_proto=model.getvalue())


TEXTS = ["a", "", " ", "Once upon a time", "the quick brown fox", "hello  world", "  leading and trailing  ",
         "dragons and castles!", "café naïve ☃ \U0001F600", "tab\there", "new\nline",
         "MiXeD CaSe 12345", "a" * 40, "▁already escaped", "the the the the", "x",
         "3.14 is pi, 2024 is a year.", "中文字符"]


def main():
    out = {}
    for name, kw in (("llama_like", dict(vocab_size=420, byte_fallback=True, add_dummy_prefix=True,
                                          remove_extra_whitespaces=False)),
                     ("plain", dict(vocab_size=160, byte_fallback=False, add_dummy_prefix=False,
                                    remove_extra_whitespaces=True))):
        sp = train(name, **kw)
        cases = []
        for t in TEXTS + corpus(99, 40):
            ids = sp.encode(t)
            cases.append({"text": t, "ids": ids, "decoded": sp.decode(ids)})
        out[name] = {"vocab_size": sp.get_piece_size(), "bos": sp.bos_id(), "eos": sp.eos_id(),
                     "unk": sp.unk_id(), "cases": cases}
    with open(os.path.join(HERE, "spm_golden.json"), "w") as f:
        json.dump(out, f, ensure_ascii=True, indent=0)
    print({k: len(v["cases"]) for k, v in out.items()})


if __name__ == "__main__":
    main()
