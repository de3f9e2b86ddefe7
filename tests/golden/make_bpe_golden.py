#!/usr/bin/env python3
"""Golden fixtures for the byte-level BPE tokenizer (kh_bpe.cpp).

The reference's BpeEncodeLayer / QwenEncodeLayer (kuiper/source/op/encode.cpp:59-183) read a
HuggingFace tokenizer.json (Llama-3.x / Qwen2.5); none ships with the reference and none can be
downloaded here, so two small byte-level vocabularies are trained in-container with the HF
`tokenizers` package (0.22, present in the image) on a synthetic corpus and saved as
tests/golden/bpe_llama3_like.json / bpe_qwen2_like.json with
  * the reference's split pattern PAT_STR (encode.cpp:59-60) as a Split(isolated) pre-tokenizer
    followed by ByteLevel(use_regex=False) - the shape of Llama-3's tokenizer.json,
  * the special tokens each flavour looks up (encode.cpp:97-103, 170-176).
The `tokenizers` encoder is the oracle: ids for the text as it is ("plain") and for the text after
the reference's " " -> "Ġ" replacement ("ref", encode.cpp:108-111), plus the decoded strings.
Writes tests/golden/bpe_golden.json.
"""
import json
import os
import random

from tokenizers import Regex, Tokenizer, decoders, models, pre_tokenizers, trainers

HERE = os.path.dirname(os.path.abspath(__file__))
# kuiper/source/op/encode.cpp:59-60 (RE2 syntax; identical semantics in the oniguruma dialect)
PAT_STR = (r"(?i:'s|'t|'re|'ve|'m|'ll|'d)|[^\r\n\p{L}\p{N}]?\p{L}+|\p{N}| ?[^\s\p{L}\p{N}]+[\r\n]*"
           r"|\s*[\r\n]+|\s+(?:$|[^\S])|\s+")
WORDS = ("the quick brown fox jumps over lazy dog once upon a time there was little girl who lived in "
         "village near forest she liked to play with her friends and read stories about dragons castles "
         "kings queens hello world token model cache rope attention it's don't we're they've I'm you'll "
         "he'd café naïve über straße 東京 中文 данные 2024 3.14 100% (ok) [x] {y} a+b=c").split()
SPECIAL = {
    "llama3_like": ["<|begin_of_text|>", "<|end_of_text|>", "<|reserved_special_token_0|>",
                    "<|start_header_id|>", "<|end_header_id|>", "<|eot_id|>"],
    "qwen2_like": ["<|endoftext|>", "<|im_start|>", "<|im_end|>"],
}


def corpus(seed, n=1500):
    r = random.Random(seed)
    out = []
    for _ in range(n):
        k = r.randint(3, 14)
        s = " ".join(r.choice(WORDS) for _ in range(k))
        if r.random() < 0.3:
            s = s.capitalize() + r.choice(".!?\n")
        if r.random() < 0.2:
            s = s.replace(" ", "Ġ")  # so the merges also cover the reference's replaced text
        out.append(s)
    return out


def train(name, vocab_size):
    tok = Tokenizer(models.BPE())
    tok.pre_tokenizer = pre_tokenizers.Sequence([
        pre_tokenizers.Split(Regex(PAT_STR), behavior="isolated"),
        pre_tokenizers.ByteLevel(add_prefix_space=False, use_regex=False)])
    tok.decoder = decoders.ByteLevel()
    # ids: the 256 byte symbols first, merges in creation order after them, special tokens LAST
    # (Llama-3 / Qwen2 layout: ids = tiktoken ranks, specials above the vocabulary)
    trainer = trainers.BpeTrainer(vocab_size=vocab_size, special_tokens=[],
                                  initial_alphabet=pre_tokenizers.ByteLevel.alphabet(), show_progress=False)
    tok.train_from_iterator(corpus(len(name)), trainer)
    tok.add_special_tokens(SPECIAL[name])
    path = os.path.join(HERE, f"bpe_{name}.json")
    tok.save(path, pretty=False)
    return Tokenizer.from_file(path)


TEXTS = ["a", "", " ", "Once upon a time", "the quick brown fox", "hello  world", "  leading and trailing  ",
         "dragons and castles!", "café naïve ☃ \U0001F600", "tab\there", "new\nline\n\n  next", "it's I'M we'RE don't",
         "MiXeD CaSe 12345", "a" * 40, "Ġalready escaped", "the the the the", "x", "   three spaces then word",
         "3.14 is pi, 2024 is a year.", "中文字符 and данные", "(ok) [x] {y} a+b=c 100%", "trailing newline\r\n",
         "<|begin_of_text|>hello<|eot_id|> world<|end_of_text|>", "<|im_start|>user\nhi<|im_end|>\n<|endoftext|>",
         "semi<|eot_id", "\t\tindented\n\tmore", "ends with space "]


def main():
    out = {"pattern": PAT_STR}
    for name, vs in (("llama3_like", 700), ("qwen2_like", 520)):
        tok = train(name, vs)
        cases = []
        for t in TEXTS:
            plain = tok.encode(t, add_special_tokens=False).ids
            ref = tok.encode(t.replace(" ", "Ġ"), add_special_tokens=False).ids
            cases.append({"text": t, "plain": plain, "ref": ref,
                          "plain_decoded": tok.decode(plain, skip_special_tokens=False),
                          "ref_decoded": tok.decode(ref, skip_special_tokens=False).replace("Ġ", " ")})
        out[name] = {"vocab_size": tok.get_vocab_size(with_added_tokens=True),
                     "special": {s: tok.token_to_id(s) for s in SPECIAL[name]}, "cases": cases}
    with open(os.path.join(HERE, "bpe_golden.json"), "w") as f:
        json.dump(out, f, ensure_ascii=False, indent=0)
    print("wrote", {k: len(v["cases"]) for k, v in out.items() if isinstance(v, dict)})


if __name__ == "__main__":
    main()
