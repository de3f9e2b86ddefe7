"""KuiperLLama ``.bin`` checkpoint layout: header parsing, tensor offsets and a seeded
synthetic-model writer.

Host-side mirror of the reference's loader for the decode path:

* header + derived sizes ...... ``kuiper/source/model/model.cpp:41-151``
* fp32 tensor order ........... ``kuiper/source/model/llama3.cpp:290-423`` (reader) /
  ``tools/export.py:79-131`` (writer, "legacy_export", version 0)
* Qwen2 q/k/v bias interleave . ``kuiper/source/model/qwen2.cpp:305-333`` /
  ``tools/export_qwen2.py:100-110``
* int8 group-quant order ...... ``kuiper/source/model/llama3.cpp:184-288`` /
  ``tools/export.py:134-210`` ("legacy_export_quant", version 3); quantiser
  ``tools/export.py:49-73``

The C++ loader in ``csrc/kh_model_load.hip`` implements the same table; tests check the two
against each other and (in the build container only) against files written by the
reference's own exporter.  No reference code is imported here.

There are no real checkpoints on disk and no network, so benchmarks and parity tests use
:func:`synth_image`: a seeded random model of the exact BASELINE shape written straight
into an in-memory image with the byte layout above.
"""
from __future__ import annotations

import dataclasses
import math
import struct
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

ROPE_INTERLEAVED = 0  # cpu/rope_kernel.cpp:98-121 (default build: llama2.c / Meta export)
ROPE_HALF = 1  # cpu/rope_kernel.cpp:18-42 / 58-82 (LLAMA3_SUPPORT / QWEN2_SUPPORT builds)
FAMILY_LLAMA = 0
FAMILY_QWEN2 = 1

HEADER_FP32_BYTES = 28  # 7 x int32 (model/config.h:5-13)
HEADER_INT8_BYTES = 32  # + int32 group_size (model.cpp:64-70)


@dataclass
class ModelSpec:
    dim: int
    hidden_dim: int
    n_layers: int
    n_heads: int
    n_kv_heads: int
    vocab_size: int
    seq_len: int
    shared_classifier: bool = True
    family: int = FAMILY_LLAMA
    quant: bool = False
    group_size: int = 64
    rope_mode: int = ROPE_INTERLEAVED
    rope_theta: float = 10000.0
    rms_eps: float = 1e-5
    name: str = "custom"

    # derived exactly as Model::generate_model_infos (model.cpp:125-151)
    @property
    def kv_dim(self) -> int:
        return (self.dim * self.n_kv_heads) // self.n_heads

    @property
    def kv_mul(self) -> int:
        return self.n_heads // self.n_kv_heads

    @property
    def head_size(self) -> int:
        return self.dim // self.n_heads

    @property
    def has_bias(self) -> bool:
        return self.family == FAMILY_QWEN2 and not self.quant

    def header_bytes(self) -> int:
        return HEADER_INT8_BYTES if self.quant else HEADER_FP32_BYTES

    def weight_elems(self) -> int:
        """W_elems of SURVEY.md §8(d): matmul weights streamed per decoded token."""
        d, kv, h, L, v = self.dim, self.kv_dim, self.hidden_dim, self.n_layers, self.vocab_size
        return L * (2 * d * d + 2 * kv * d + 3 * h * d) + v * d

    def algorithmic_bytes_per_token(self, pos: float) -> float:
        """Algorithmic HBM bytes for one decode step at position ``pos``:
        weights once + KV read/write + norms/embedding row/biases/logits."""
        d, kv, L, v = self.dim, self.kv_dim, self.n_layers, self.vocab_size
        we = self.weight_elems()
        if self.quant:
            wbytes = we + (we // self.group_size) * 4
        else:
            wbytes = we * 4
        small = (2 * L + 1) * d * 4 + d * 4 + v * 4
        if self.has_bias:
            small += L * (d + 2 * kv) * 4
        kvb = L * (2 * (pos + 1) * kv * 4 + 2 * kv * 4)
        return float(wbytes + small + kvb)

    def kernel_bytes(self) -> Dict[str, float]:
        """Algorithmic HBM bytes of ONE launch of each weight-streaming kernel of the decode step:
        the rows of its matrices once (+ group scales), its input vector (+ norm weight) once, its output once."""
        d, kv, h, v = self.dim, self.kv_dim, self.hidden_dim, self.vocab_size

        def w(n):
            return float(n + (n // self.group_size) * 4 if self.quant else n * 4)
        bias = (d + 2 * kv) * 4 if self.has_bias else 0
        return {"qkv": w((d + 2 * kv) * d) + 2 * d * 4 + (d + 2 * kv) * 4 + bias,
                "wo": w(d * d) + d * 4 + 2 * d * 4,
                "ffn13": w(2 * h * d) + 2 * d * 4 + h * 4,
                "w2": w(d * h) + h * 4 + 2 * d * 4,
                "cls": w(v * d) + 2 * d * 4 + v * 4}


# BASELINE.json configs -------------------------------------------------
PRESETS: Dict[str, ModelSpec] = {
    "stories15M": ModelSpec(288, 768, 6, 6, 6, 32000, 256, True, FAMILY_LLAMA, False, 64,
                            ROPE_INTERLEAVED, 10000.0, 1e-5, "stories15M"),
    "llama3.2-1b": ModelSpec(2048, 8192, 16, 32, 8, 128256, 131072, True, FAMILY_LLAMA, False,
                             64, ROPE_HALF, 500000.0, 1e-5, "llama3.2-1b"),
    "llama2-7b-int8": ModelSpec(4096, 11008, 32, 32, 32, 32000, 2048, False, FAMILY_LLAMA, True,
                                64, ROPE_INTERLEAVED, 10000.0, 1e-5, "llama2-7b-int8"),
    "llama2-7b": ModelSpec(4096, 11008, 32, 32, 32, 32000, 2048, False, FAMILY_LLAMA, False, 64,
                           ROPE_INTERLEAVED, 10000.0, 1e-5, "llama2-7b"),
    "qwen2.5-0.5b": ModelSpec(896, 4864, 24, 14, 2, 151936, 32768, True, FAMILY_QWEN2, False,
                              64, ROPE_HALF, 1000000.0, 1e-6, "qwen2.5-0.5b"),
    "tinyllama-1.1b": ModelSpec(2048, 5632, 22, 32, 4, 32000, 2048, False, FAMILY_LLAMA, False,
                                64, ROPE_INTERLEAVED, 10000.0, 1e-5, "tinyllama-1.1b"),
}


def parse_header(image: bytes | memoryview | np.ndarray, quant: bool) -> Tuple[int, ...]:
    """Return (dim, hidden, layers, heads, kv_heads, signed_vocab, seq_len[, group_size])."""
    raw = bytes(memoryview(image)[: (HEADER_INT8_BYTES if quant else HEADER_FP32_BYTES)])
    n = 8 if quant else 7
    if len(raw) < 4 * n:
        raise ValueError("image too small for a KuiperLLama .bin header")
    return struct.unpack("<" + "i" * n, raw)


def spec_from_image(image, *, family: int = FAMILY_LLAMA, quant: bool = False,
                    rope_mode: int = ROPE_INTERLEAVED, rope_theta: float = 10000.0,
                    rms_eps: float = 1e-5, name: str = "from-image") -> ModelSpec:
    h = parse_header(image, quant)
    return ModelSpec(h[0], h[1], h[2], h[3], h[4], abs(h[5]), h[6], h[5] > 0, family, quant,
                     h[7] if quant else 64, rope_mode, rope_theta, rms_eps, name)


@dataclass
class TensorEntry:
    name: str
    offset: int  # byte offset from the start of the image (header included)
    shape: Tuple[int, ...]
    dtype: str  # "f32" | "i8"

    @property
    def nbytes(self) -> int:
        n = 1
        for s in self.shape:
            n *= s
        return n * (4 if self.dtype == "f32" else 1)


def layout(spec: ModelSpec) -> Tuple[List[TensorEntry], int]:
    """Tensor table (file order) and total image size in bytes."""
    L, d, kv, hid, V = spec.n_layers, spec.dim, spec.kv_dim, spec.hidden_dim, spec.vocab_size
    hs = spec.head_size
    ents: List[TensorEntry] = []
    off = spec.header_bytes()

    def take(name, shape, dtype="f32"):
        nonlocal off
        e = TensorEntry(name, off, tuple(shape), dtype)
        ents.append(e)
        off += e.nbytes

    if not spec.quant:
        take("tok_emb", (V, d))
        for l in range(L):
            take(f"att_norm.{l}", (d,))
        for nm, K in (("wq", d), ("wk", kv), ("wv", kv)):
            for l in range(L):
                take(f"{nm}.{l}", (K, d))
                if spec.has_bias:
                    take(f"b{nm[1]}.{l}", (K,))
        for l in range(L):
            take(f"wo.{l}", (d, d))
        for l in range(L):
            take(f"ffn_norm.{l}", (d,))
        for l in range(L):
            take(f"w1.{l}", (hid, d))
        for l in range(L):
            take(f"w2.{l}", (d, hid))
        for l in range(L):
            take(f"w3.{l}", (hid, d))
        take("final_norm", (d,))
        take("freqs_cos", (spec.seq_len, hs // 2))
        take("freqs_sin", (spec.seq_len, hs // 2))
        if not spec.shared_classifier:
            take("wcls", (V, d))
    else:
        if spec.shared_classifier or spec.family == FAMILY_QWEN2:
            raise ValueError("int8 export with a tied classifier / Qwen2 is broken in the "
                             "reference (llama3.cpp:259-262); not supported")
        gs = spec.group_size
        for nm, K, M in (("wq", d, d), ("wk", kv, d), ("wv", kv, d), ("wo", d, d),
                         ("w1", hid, d), ("w2", d, hid), ("w3", hid, d)):
            if (K * M) % gs:
                raise ValueError("weight size not divisible by group size")
            for l in range(L):
                take(f"{nm}.{l}", (K, M), "i8")
                take(f"{nm}.{l}.scales", (K * M // gs,))
        take("wcls", (V, d), "i8")
        take("wcls.scales", (V * d // gs,))
        take("tok_emb", (V, d))
        for l in range(L):
            take(f"att_norm.{l}", (d,))
        for l in range(L):
            take(f"ffn_norm.{l}", (d,))
        take("final_norm", (d,))
    return ents, off


def image_nbytes(spec: ModelSpec) -> int:
    return layout(spec)[1]


def quantize_q80_torch(w: torch.Tensor, group_size: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Symmetric per-group int8 quantiser, same arithmetic as tools/export.py:49-73:
    scale = max|w| / 127 ; q = round(w / scale) (round-half-even)."""
    flat = w.reshape(-1, group_size).float()
    wmax = flat.abs().max(dim=1).values
    scale = wmax / 127.0
    q = torch.round(flat / scale[:, None]).to(torch.int8)
    return q.reshape(-1), scale


def synth_image(spec: ModelSpec, seed: int = 1234, device: str | torch.device = "cpu",
                norm_jitter: float = 0.05, final_norm_std: Optional[float] = None) -> torch.Tensor:
    """Seeded random model of shape ``spec`` as a uint8 tensor holding the exact ``.bin``
    bytes (header included).  Init follows the reference's own Python model
    (tools/model.py:232-247): N(0, 0.02) for embeddings/linears, N(0, 0.02/sqrt(2L)) for
    wo and w3; Qwen2 biases N(0, 0.02).  Norm weights are 1 + N(0, norm_jitter) so that
    the norm-weight multiply is exercised.  freqs_cos/sin are written as the exporter
    does (the C++ reader skips them).

    ``final_norm_std``: when given, the final norm weight is N(0, final_norm_std) instead (zero
    mean).  With the init above a TIED classifier maps the residual stream - dominated by the
    input token's embedding row - back onto that very token, so greedy decoding sits on a fixed
    point (one id repeated); a signed final norm removes the self-match and the greedy sequence
    wanders over the vocabulary (parity tests that must not be satisfied by a repeated token).
    """
    device = torch.device(device)
    ents, total = layout(spec)
    # Place the image so that the bytes BEHIND the header - what the library uses in place when the image is handed
    # over on the device - start on a 4-KiB boundary, as they do in the arena kh_model_create_from_file /
    # _from_host_image allocate.  Every tensor of an exported image is a multiple of 256 bytes long in the BASELINE
    # geometries, so all weight rows then start on a line boundary; with the int8 header (32 bytes: a 16-byte
    # aligned, usable address) every row sat 32 bytes off one and each 1-KiB wave request straddled nine 128-byte
    # lines instead of eight (measured on ffn13 int8: 18.0 vs 17.4 us, profiles/r5_int8_ring_ab.txt).
    pad = 4096  # hipMalloc's own alignment is at least this
    buf = torch.empty(total + pad, dtype=torch.uint8, device=device)
    lead = (-(buf.data_ptr() + spec.header_bytes())) % pad
    img = buf[lead: lead + total]
    hdr = [spec.dim, spec.hidden_dim, spec.n_layers, spec.n_heads, spec.n_kv_heads,
           spec.vocab_size if spec.shared_classifier else -spec.vocab_size, spec.seq_len]
    if spec.quant:
        hdr.append(spec.group_size)
    hb = np.frombuffer(struct.pack("<" + "i" * len(hdr), *hdr), dtype=np.uint8).copy()
    img[: hb.size] = torch.from_numpy(hb).to(device)

    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    std_res = 0.02 / math.sqrt(2 * spec.n_layers)
    pending_scale: Optional[torch.Tensor] = None

    def fview(e: TensorEntry) -> torch.Tensor:
        # fp32 tensors sit at 4-byte aligned offsets (28/32-byte header, all sizes x4)
        return img[e.offset: e.offset + e.nbytes].view(torch.float32)

    for e in ents:
        base = e.name.split(".")[0]
        if e.dtype == "i8":
            std = std_res if base in ("wo", "w3") else 0.02
            n = e.nbytes
            # one tensor at a time: the fp32 transient is at most 11008x4096x4 = 180 MB
            w = torch.empty(n, dtype=torch.float32, device=device).normal_(0.0, std, generator=gen)
            q, sc = quantize_q80_torch(w, spec.group_size)
            img[e.offset: e.offset + n] = q.view(torch.uint8)
            pending_scale = sc
            del w, q
            continue
        dst = fview(e)
        if e.name.endswith(".scales"):
            assert pending_scale is not None
            dst.copy_(pending_scale)
            pending_scale = None
        elif base == "final_norm" and final_norm_std is not None:
            dst.normal_(0.0, final_norm_std, generator=gen)
        elif base in ("att_norm", "ffn_norm", "final_norm"):
            dst.normal_(0.0, norm_jitter, generator=gen).add_(1.0)
        elif base == "freqs_cos" or base == "freqs_sin":
            hs = spec.head_size
            # writing 2 x seq_len x hs/2 floats only matters for byte-compat; keep it cheap
            # for 131072-long contexts by computing on `device`.
            fr = 1.0 / (spec.rope_theta ** (torch.arange(0, hs, 2, device=device)[: hs // 2].float() / hs))
            t = torch.arange(spec.seq_len, device=device)
            ang = torch.outer(t, fr).float()
            dst.copy_((torch.cos(ang) if base == "freqs_cos" else torch.sin(ang)).reshape(-1))
        elif base in ("wo", "w3"):
            dst.normal_(0.0, std_res, generator=gen)
        else:  # tok_emb, wq/wk/wv, w1/w2, wcls, biases
            dst.normal_(0.0, 0.02, generator=gen)
    return img


def tensor_from_image(img: torch.Tensor, e: TensorEntry) -> torch.Tensor:
    raw = img[e.offset: e.offset + e.nbytes]
    t = raw.view(torch.float32 if e.dtype == "f32" else torch.int8)
    return t.reshape(e.shape)


def spec_to_dict(spec: ModelSpec) -> dict:
    return dataclasses.asdict(spec)
