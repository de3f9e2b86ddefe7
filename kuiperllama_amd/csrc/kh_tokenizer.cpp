// kh_tokenizer.cpp — SentencePiece-BPE encode/decode behind the C-ABI (host only, no GPU).
//
// Reference boundary: op::SpeEncodeLayer (kuiper/source/op/encode.cpp:10-57) = thin wrapper over
// the external sentencepiece library (SentencePieceProcessor::Load/Encode/Decode, bos_id/eos_id,
// GetPieceSize).  That library is a dependency OUTSIDE /root/reference (CMake fetches
// google/sentencepiece, version unpinned); its published algorithm for BPE models is restated
// here so that a KuiperLLama build without it can still turn the demo's prompt string into ids
// and the `words` vector back into text:
//   * model file = serialized sentencepiece ModelProto (protobuf wire format, parsed by hand);
//   * normalisation = the NormalizerSpec flags add_dummy_prefix / remove_extra_whitespaces /
//     escape_whitespaces with the IDENTITY character map (Llama-2's tokenizer.model); a model
//     that carries a precompiled character map (NFKC etc.) is rejected as unsupported;
//   * segmentation = bpe_model.cc: start from UTF-8 characters, repeatedly merge the adjacent
//     pair whose concatenation is a vocabulary piece with the highest score (ties: leftmost),
//     then map pieces to ids with byte fallback (<0xXX> pieces) or unk;
//   * decoding = concatenate pieces, control pieces vanish, byte pieces re-form UTF-8,
//     U+2581 -> ' ', the dummy-prefix space is dropped.
// Pinned against the sentencepiece Python package on models trained in-container
// (tests/golden/make_spm_golden.py, tests/test_tokenizer.py).  The Llama-3 / Qwen byte-level BPE
// (encode.cpp:59-180, tiktoken.h) lives in kh_bpe.cpp.
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <new>
#include <queue>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/kuiper_hip.h"

namespace {

enum PieceType { kNormal = 1, kUnknown = 2, kControl = 3, kUserDefined = 4, kUnused = 5, kByte = 6 };

struct Piece {
  std::string text;
  float score = 0.f;
  int type = kNormal;
};

// ---- minimal protobuf wire reader -----------------------------------------------------------
struct Reader {
  const uint8_t* p;
  const uint8_t* end;
  bool ok = true;
  uint64_t varint() {
    uint64_t v = 0;
    int shift = 0;
    while (p < end && shift < 64) {
      const uint8_t b = *p++;
      v |= (uint64_t)(b & 0x7f) << shift;
      if (!(b & 0x80)) return v;
      shift += 7;
    }
    ok = false;
    return 0;
  }
  // reads a key; returns false at end of buffer
  bool key(uint32_t& field, uint32_t& wire) {
    if (p >= end || !ok) return false;
    const uint64_t k = varint();
    field = (uint32_t)(k >> 3);
    wire = (uint32_t)(k & 7);
    return ok;
  }
  Reader sub() {  // length-delimited payload
    const uint64_t n = varint();
    if (!ok || n > (uint64_t)(end - p)) {
      ok = false;
      return Reader{p, p};
    }
    Reader r{p, p + n};
    p += n;
    return r;
  }
  void skip(uint32_t wire) {
    switch (wire) {
      case 0: (void)varint(); break;
      case 1: p += 8; break;
      case 2: (void)sub(); break;
      case 5: p += 4; break;
      default: ok = false;
    }
    if (p > end) ok = false;
  }
  float f32() {
    if (end - p < 4) {
      ok = false;
      return 0.f;
    }
    float v;
    memcpy(&v, p, 4);
    p += 4;
    return v;
  }
};

const char kSpace[] = "\xE2\x96\x81";  // U+2581 LOWER ONE EIGHTH BLOCK, the whitespace marker

inline int utf8_len(unsigned char c) {
  if (c < 0x80) return 1;
  if ((c >> 5) == 0x6) return 2;
  if ((c >> 4) == 0xE) return 3;
  if ((c >> 3) == 0x1E) return 4;
  return 1;  // invalid lead byte: one byte at a time (sentencepiece does the same)
}

}  // namespace

struct kh_spm {
  std::vector<Piece> pieces;
  std::unordered_map<std::string, int> index;  // NORMAL / USER_DEFINED pieces usable by merges
  int byte_id[256];
  int unk_id = 0, bos_id = 1, eos_id = 2, pad_id = -1;
  bool byte_fallback = false;
  bool add_dummy_prefix = true, remove_extra_ws = true, escape_ws = true;
};

namespace {

int parse_model(kh_spm* t, const uint8_t* data, size_t n) {
  Reader r{data, data + n};
  uint32_t f, w;
  int model_type = 1;  // UNIGRAM is the proto default
  bool has_charsmap = false, ws_suffix = false;
  while (r.key(f, w)) {
    if (f == 1 && w == 2) {  // repeated SentencePiece pieces
      Reader s = r.sub();
      Piece pc;
      uint32_t sf, sw;
      while (s.key(sf, sw)) {
        if (sf == 1 && sw == 2) {
          Reader b = s.sub();
          pc.text.assign((const char*)b.p, (size_t)(b.end - b.p));
        } else if (sf == 2 && sw == 5) {
          pc.score = s.f32();
        } else if (sf == 3 && sw == 0) {
          pc.type = (int)s.varint();
        } else {
          s.skip(sw);
        }
      }
      if (!s.ok) return KH_ERR_FORMAT;
      t->pieces.push_back(std::move(pc));
    } else if (f == 2 && w == 2) {  // TrainerSpec
      Reader s = r.sub();
      uint32_t sf, sw;
      while (s.key(sf, sw)) {
        if (sw != 0) {
          s.skip(sw);
          continue;
        }
        const int64_t v = (int64_t)s.varint();
        switch (sf) {
          case 3: model_type = (int)v; break;
          case 24: ws_suffix = v != 0; break;
          case 35: t->byte_fallback = v != 0; break;
          case 40: t->unk_id = (int)v; break;
          case 41: t->bos_id = (int)v; break;
          case 42: t->eos_id = (int)v; break;
          case 43: t->pad_id = (int)v; break;
          default: break;
        }
      }
      if (!s.ok) return KH_ERR_FORMAT;
    } else if (f == 3 && w == 2) {  // NormalizerSpec
      Reader s = r.sub();
      uint32_t sf, sw;
      while (s.key(sf, sw)) {
        if (sf == 2 && sw == 2) {
          Reader b = s.sub();
          has_charsmap = b.end > b.p;
        } else if (sw == 0) {
          const uint64_t v = s.varint();
          if (sf == 3) t->add_dummy_prefix = v != 0;
          if (sf == 4) t->remove_extra_ws = v != 0;
          if (sf == 5) t->escape_ws = v != 0;
        } else {
          s.skip(sw);
        }
      }
      if (!s.ok) return KH_ERR_FORMAT;
    } else {
      r.skip(w);
    }
  }
  if (!r.ok || t->pieces.empty()) return KH_ERR_FORMAT;
  if (model_type != 2 /*BPE*/ || has_charsmap || ws_suffix) return KH_ERR_UNSUPPORTED;
  for (int i = 0; i < 256; ++i) t->byte_id[i] = -1;
  for (size_t i = 0; i < t->pieces.size(); ++i) {
    const Piece& p = t->pieces[i];
    if (p.type == kNormal || p.type == kUserDefined) t->index.emplace(p.text, (int)i);
    if (p.type == kByte && p.text.size() == 6) {  // "<0xXX>"
      unsigned v = 0;
      if (sscanf(p.text.c_str(), "<0x%02X>", &v) == 1 && v < 256) t->byte_id[v] = (int)i;
    }
  }
  return KH_OK;
}

// NormalizerSpec with the identity character map (normalizer.cc)
std::string normalize(const kh_spm* t, const char* s, size_t n) {
  std::string in(s, n);
  if (t->remove_extra_ws) {
    std::string o;
    bool prev_space = true;  // drops leading spaces
    for (char c : in) {
      if (c == ' ') {
        if (!prev_space) o.push_back(' ');
        prev_space = true;
      } else {
        o.push_back(c);
        prev_space = false;
      }
    }
    while (!o.empty() && o.back() == ' ') o.pop_back();
    in.swap(o);
  }
  if (in.empty()) return in;
  std::string out;
  out.reserve(in.size() + 8);
  if (t->add_dummy_prefix) out += t->escape_ws ? kSpace : " ";
  for (char c : in) {
    if (c == ' ' && t->escape_ws)
      out += kSpace;
    else
      out.push_back(c);
  }
  if (t->remove_extra_ws) {
    // normalizer.cc strips trailing whitespace on the ESCAPED text, so a literal U+2581 at the
    // end of the input goes too
    const std::string sp = t->escape_ws ? std::string(kSpace) : std::string(" ");
    while (out.size() >= sp.size() && out.compare(out.size() - sp.size(), sp.size(), sp) == 0)
      out.erase(out.size() - sp.size());
  }
  return out;
}

struct Sym {
  int prev, next;
  size_t begin, len;  // byte range in the normalised string; len == 0 => merged away
};
struct Cand {
  int left, right;
  float score;
  size_t size;
};
struct CandLess {  // max-heap: highest score first, ties -> leftmost (bpe_model.cc SymbolPairComparator)
  bool operator()(const Cand& a, const Cand& b) const {
    return a.score < b.score || (a.score == b.score && a.left > b.left);
  }
};

void encode_ids(const kh_spm* t, const std::string& norm, std::vector<int32_t>& out) {
  std::vector<Sym> sym;
  for (size_t i = 0; i < norm.size();) {
    size_t l = (size_t)utf8_len((unsigned char)norm[i]);
    if (i + l > norm.size()) l = norm.size() - i;
    Sym s;
    s.begin = i;
    s.len = l;
    s.prev = (int)sym.size() - 1;
    s.next = (int)sym.size() + 1;
    sym.push_back(s);
    i += l;
  }
  if (sym.empty()) return;
  sym.back().next = -1;
  std::priority_queue<Cand, std::vector<Cand>, CandLess> agenda;
  auto maybe_add = [&](int l, int r) {
    if (l < 0 || r < 0) return;
    const std::string piece = norm.substr(sym[l].begin, sym[l].len + sym[r].len);
    auto it = t->index.find(piece);
    if (it == t->index.end()) return;
    agenda.push(Cand{l, r, t->pieces[it->second].score, piece.size()});
  };
  for (int i = 1; i < (int)sym.size(); ++i) maybe_add(i - 1, i);
  while (!agenda.empty()) {
    const Cand c = agenda.top();
    agenda.pop();
    Sym& L = sym[c.left];
    Sym& R = sym[c.right];
    if (L.len == 0 || R.len == 0 || L.len + R.len != c.size) continue;  // stale
    L.len += R.len;  // adjacent by construction
    R.len = 0;
    L.next = R.next;
    if (R.next >= 0) sym[R.next].prev = c.left;
    maybe_add(L.prev, c.left);
    maybe_add(c.left, L.next);
  }
  bool prev_unk = false;
  for (int i = 0; i >= 0; i = sym[i].next) {
    const Sym& s = sym[i];
    const std::string piece = norm.substr(s.begin, s.len);
    auto it = t->index.find(piece);
    if (it != t->index.end()) {
      out.push_back(it->second);
      prev_unk = false;
    } else if (t->byte_fallback) {
      for (unsigned char ch : piece) out.push_back(t->byte_id[ch] >= 0 ? t->byte_id[ch] : t->unk_id);
      prev_unk = false;
    } else {
      // sentencepiece_processor.cc PopulateSentencePieceText: a run of unknown pieces is one <unk>
      if (!prev_unk) out.push_back(t->unk_id);
      prev_unk = true;
    }
  }
}

inline bool valid_utf8(const std::string& b, size_t i, size_t& l) {
  const unsigned char c = (unsigned char)b[i];
  l = (size_t)utf8_len(c);
  if (c >= 0x80 && l == 1) return false;
  if (i + l > b.size()) return false;
  for (size_t k = 1; k < l; ++k)
    if (((unsigned char)b[i + k] & 0xC0) != 0x80) return false;
  return true;
}

std::string decode_ids(const kh_spm* t, const int32_t* ids, int n) {
  std::string out, bytes;
  auto flush_bytes = [&]() {
    for (size_t i = 0; i < bytes.size();) {
      size_t l;
      if (valid_utf8(bytes, i, l)) {
        out.append(bytes, i, l);
        i += l;
      } else {
        out += "\xEF\xBF\xBD";  // U+FFFD
        i += 1;
      }
    }
    bytes.clear();
  };
  bool first = true;
  for (int k = 0; k < n; ++k) {
    const int id = ids[k];
    if (id < 0 || id >= (int)t->pieces.size()) continue;
    const Piece& p = t->pieces[id];
    if (p.type == kByte) {
      unsigned v = 0;
      if (sscanf(p.text.c_str(), "<0x%02X>", &v) == 1) bytes.push_back((char)v);
      continue;
    }
    flush_bytes();
    if (p.type == kControl) continue;
    if (p.type == kUnknown) {
      out += " \xE2\x81\x87 ";  // default unk_surface " ?? " (U+2047)
      first = false;
      continue;
    }
    std::string s = p.text;
    for (size_t pos = 0; (pos = s.find(kSpace, pos)) != std::string::npos;) s.replace(pos, 3, " ");
    // sentencepiece_processor.cc DecodeIds: a whitespace marker at the very start of the text is
    // dropped when the normaliser would have produced it (dummy prefix) or removed it anyway
    if (first && (t->add_dummy_prefix || t->remove_extra_ws) && !s.empty() && s[0] == ' ')
      s.erase(0, 1);
    first = false;
    out += s;
  }
  flush_bytes();
  return out;
}

}  // namespace

extern "C" int kh_spm_create_from_memory(const void* data, int64_t nbytes, kh_spm** out) {
  if (!data || nbytes <= 0 || !out) return KH_ERR_INVALID_ARG;
  kh_spm* t = new (std::nothrow) kh_spm();
  if (!t) return KH_ERR_IO;
  const int rc = parse_model(t, (const uint8_t*)data, (size_t)nbytes);
  if (rc != KH_OK) {
    delete t;
    return rc;
  }
  *out = t;
  return KH_OK;
}

extern "C" int kh_spm_create_from_file(const char* path, kh_spm** out) {
  if (!path || !out) return KH_ERR_INVALID_ARG;
  FILE* f = fopen(path, "rb");
  if (!f) return KH_ERR_IO;
  std::vector<char> buf;
  char tmp[65536];
  size_t got;
  while ((got = fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + got);
  fclose(f);
  if (buf.empty()) return KH_ERR_FORMAT;
  return kh_spm_create_from_memory(buf.data(), (int64_t)buf.size(), out);
}

extern "C" void kh_spm_destroy(kh_spm* t) { delete t; }
extern "C" int32_t kh_spm_vocab_size(const kh_spm* t) { return t ? (int32_t)t->pieces.size() : 0; }
extern "C" int32_t kh_spm_bos_id(const kh_spm* t) { return t ? t->bos_id : -1; }
extern "C" int32_t kh_spm_eos_id(const kh_spm* t) { return t ? t->eos_id : -1; }
extern "C" int32_t kh_spm_unk_id(const kh_spm* t) { return t ? t->unk_id : -1; }

extern "C" int kh_spm_encode(const kh_spm* t, const char* utf8, int64_t len, int32_t add_bos,
                             int32_t add_eos, int32_t* out_ids, int32_t cap, int32_t* n_ids) {
  if (!t || (!utf8 && len > 0) || len < 0 || !n_ids || cap < 0 || (cap > 0 && !out_ids))
    return KH_ERR_INVALID_ARG;
  std::vector<int32_t> ids;
  if (add_bos) ids.push_back(t->bos_id);  // encode.cpp:37-41
  encode_ids(t, normalize(t, utf8, (size_t)len), ids);
  if (add_eos) ids.push_back(t->eos_id);  // encode.cpp:42-44
  *n_ids = (int32_t)ids.size();
  if ((int64_t)ids.size() > cap) return KH_ERR_RANGE;  // *n_ids = the capacity needed
  if (!ids.empty()) memcpy(out_ids, ids.data(), ids.size() * sizeof(int32_t));
  return KH_OK;
}

extern "C" int kh_spm_decode(const kh_spm* t, const int32_t* ids, int32_t n, char* out,
                             int64_t cap, int64_t* out_len) {
  if (!t || (!ids && n > 0) || n < 0 || !out_len || cap < 0 || (cap > 0 && !out))
    return KH_ERR_INVALID_ARG;
  const std::string s = decode_ids(t, ids, n);
  *out_len = (int64_t)s.size();
  if ((int64_t)s.size() > cap) return KH_ERR_RANGE;  // *out_len = the capacity needed
  if (!s.empty()) memcpy(out, s.data(), s.size());
  return KH_OK;
}
