// kh_bpe.cpp — byte-level BPE tokenizer over a HuggingFace tokenizer.json (host only).
//
// Replaces op::BpeEncodeLayer / op::QwenEncodeLayer (kuiper/source/op/encode.cpp:59-183) and the
// libraries they sit on: nlohmann::json (file parsing), the vendored tiktoken.h
// (kuiper/include/base/tiktoken.h:17-268: special-token split, regex pre-split, rank-ordered byte
// pair merging), RE2 (the pattern PAT_STR, encode.cpp:59-60), abseil (the " " <-> "Ġ" replacement,
// encode.cpp:108-111,124-126) and the vendored unicode tables.  Nothing of those is linked here:
//   * a small JSON reader that walks tokenizer.json and keeps `added_tokens` and `model.vocab`;
//   * the GPT-2 byte <-> code-point table (unicode.cpp:155-201) to turn vocab keys into raw bytes
//     exactly as the constructor at encode.cpp:84-95 does;
//   * a hand-written scanner for PAT_STR (no regex engine): leftmost-first alternation, RE2's
//     ASCII-only \s, \p{L} / \p{N} from generated range tables (kh_unicode_tables.h);
//   * tiktoken's merge loop: repeatedly merge the adjacent pair whose CONCATENATION has the lowest
//     rank (= token id), leftmost on ties (tiktoken.h:17-98).
// Reference behaviours kept on purpose:
//   * KH_BPE_REF_SPACES: every ' ' of the input becomes "Ġ" (U+0120) BEFORE splitting and the
//     reverse after decoding (encode.cpp:108-111,124-126) - which makes spaces letters for the
//     pattern; without the flag the text is encoded as it is (what HF `tokenizers` does with the
//     same pattern and ByteLevel(use_regex=False));
//   * all added_tokens are recognised inside the text (tiktoken.h:189-190 passes every special token
//     as allowed); BOS on for Llama-3 / off for Qwen2 is the caller's flag (model.cpp:158-165);
//   * two stop ids (encode.cpp:100-103, 173-176); vocab size = |vocab| + |added_tokens|.
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/kuiper_hip.h"
#include "kh_unicode_tables.h"

namespace {

// ---- UTF-8 ---------------------------------------------------------------------------------------
// decodes one code point at s[i]; malformed bytes come back as themselves (length 1), which keeps
// the scanner total: such a byte is "other" for the pattern and goes through the byte-level BPE.
inline uint32_t utf8_next(const std::string& s, size_t i, size_t* len) {
  const unsigned char c = (unsigned char)s[i];
  auto cont = [&](size_t k) { return i + k < s.size() && ((unsigned char)s[i + k] & 0xC0) == 0x80; };
  if (c < 0x80) {
    *len = 1;
    return c;
  }
  if ((c & 0xE0) == 0xC0 && cont(1)) {
    *len = 2;
    return ((c & 0x1Fu) << 6) | ((unsigned char)s[i + 1] & 0x3Fu);
  }
  if ((c & 0xF0) == 0xE0 && cont(1) && cont(2)) {
    *len = 3;
    return ((c & 0x0Fu) << 12) | (((unsigned char)s[i + 1] & 0x3Fu) << 6) | ((unsigned char)s[i + 2] & 0x3Fu);
  }
  if ((c & 0xF8) == 0xF0 && cont(1) && cont(2) && cont(3)) {
    *len = 4;
    return ((c & 0x07u) << 18) | (((unsigned char)s[i + 1] & 0x3Fu) << 12) |
           (((unsigned char)s[i + 2] & 0x3Fu) << 6) | ((unsigned char)s[i + 3] & 0x3Fu);
  }
  *len = 1;
  return 0xFFFD0000u | c;  // not a code point: classified as "other"
}
inline void utf8_put(std::string& out, uint32_t cp) {
  if (cp < 0x80) {
    out += (char)cp;
  } else if (cp < 0x800) {
    out += (char)(0xC0 | (cp >> 6));
    out += (char)(0x80 | (cp & 0x3F));
  } else if (cp < 0x10000) {
    out += (char)(0xE0 | (cp >> 12));
    out += (char)(0x80 | ((cp >> 6) & 0x3F));
    out += (char)(0x80 | (cp & 0x3F));
  } else {
    out += (char)(0xF0 | (cp >> 18));
    out += (char)(0x80 | ((cp >> 12) & 0x3F));
    out += (char)(0x80 | ((cp >> 6) & 0x3F));
    out += (char)(0x80 | (cp & 0x3F));
  }
}

template <size_t N>
inline bool in_ranges(const KhCpRange (&t)[N], uint32_t cp) {
  size_t lo = 0, hi = N;
  while (lo < hi) {
    const size_t mid = (lo + hi) / 2;
    if (cp < t[mid].lo) hi = mid;
    else if (cp > t[mid].hi) lo = mid + 1;
    else return true;
  }
  return false;
}
enum : uint8_t { C_OTHER = 0, C_L = 1, C_N = 2, C_SPACE = 3, C_NL = 4 };  // C_NL: \r \n (also \s)
inline uint8_t classify(uint32_t cp) {
  if (cp == '\r' || cp == '\n') return C_NL;
  if (cp == ' ' || cp == '\t' || cp == '\f') return C_SPACE;  // RE2's \s is ASCII: [\t\n\f\r ]
  if (cp < 0x80) {
    if ((cp >= 'a' && cp <= 'z') || (cp >= 'A' && cp <= 'Z')) return C_L;
    if (cp >= '0' && cp <= '9') return C_N;
    return C_OTHER;
  }
  if (cp > 0x10FFFF) return C_OTHER;
  if (in_ranges(kh_unicode_L, cp)) return C_L;
  if (in_ranges(kh_unicode_N, cp)) return C_N;
  return C_OTHER;
}

// ---- JSON (only what tokenizer.json needs) ------------------------------------------------------
struct Json {
  const char* p;
  const char* end;
  bool ok = true;
  void ws() {
    while (p < end && (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t')) ++p;
  }
  bool eat(char c) {
    ws();
    if (p < end && *p == c) {
      ++p;
      return true;
    }
    return false;
  }
  bool peek(char c) {
    ws();
    return p < end && *p == c;
  }
  static int hex4(const char* q) {
    int v = 0;
    for (int i = 0; i < 4; ++i) {
      const char c = q[i];
      v <<= 4;
      if (c >= '0' && c <= '9') v |= c - '0';
      else if (c >= 'a' && c <= 'f') v |= c - 'a' + 10;
      else if (c >= 'A' && c <= 'F') v |= c - 'A' + 10;
      else return -1;
    }
    return v;
  }
  bool str(std::string* out) {  // out may be null (skip)
    ws();
    if (p >= end || *p != '"') return ok = false;
    ++p;
    while (p < end && *p != '"') {
      if (*p == '\\') {
        if (p + 1 >= end) return ok = false;
        const char e = p[1];
        p += 2;
        uint32_t cp;
        switch (e) {
          case 'n': cp = '\n'; break;
          case 't': cp = '\t'; break;
          case 'r': cp = '\r'; break;
          case 'b': cp = '\b'; break;
          case 'f': cp = '\f'; break;
          case 'u': {
            if (p + 4 > end) return ok = false;
            int v = hex4(p);
            if (v < 0) return ok = false;
            p += 4;
            cp = (uint32_t)v;
            if (cp >= 0xD800 && cp < 0xDC00 && p + 6 <= end && p[0] == '\\' && p[1] == 'u') {
              const int lo = hex4(p + 2);
              if (lo >= 0xDC00 && lo < 0xE000) {
                cp = 0x10000 + ((cp - 0xD800) << 10) + ((uint32_t)lo - 0xDC00);
                p += 6;
              }
            }
            break;
          }
          default: cp = (unsigned char)e; break;  // \" \\ \/
        }
        if (out) utf8_put(*out, cp);
      } else {
        if (out) *out += *p;
        ++p;
      }
    }
    if (p >= end) return ok = false;
    ++p;
    return true;
  }
  bool integer(long* v) {
    // digits parsed by hand, bounded by `end`: the buffer (kh_bpe_create_from_memory's ptr/nbytes)
    // is not NUL-terminated, so strtol could read past a file that is truncated inside a number
    ws();
    const char* q = p;
    bool neg = false;
    if (q < end && (*q == '-' || *q == '+')) neg = *q++ == '-';
    if (q >= end || *q < '0' || *q > '9') return ok = false;
    long x = 0;
    while (q < end && *q >= '0' && *q <= '9') {
      if (x > (0x7fffffffL - 9) / 10) return ok = false;  // ids are int32
      x = x * 10 + (*q++ - '0');
    }
    // a fraction / exponent would not be an id
    p = q;
    if (v) *v = neg ? -x : x;
    return true;
  }
  bool skip() {  // any value
    ws();
    if (p >= end) return ok = false;
    if (*p == '"') return str(nullptr);
    if (*p == '{') {
      ++p;
      if (eat('}')) return true;
      do {
        if (!str(nullptr) || !eat(':') || !skip()) return ok = false;
      } while (eat(','));
      return eat('}') ? true : (ok = false);
    }
    if (*p == '[') {
      ++p;
      if (eat(']')) return true;
      do {
        if (!skip()) return ok = false;
      } while (eat(','));
      return eat(']') ? true : (ok = false);
    }
    while (p < end && *p != ',' && *p != '}' && *p != ']' && *p != ' ' && *p != '\n' && *p != '\r' && *p != '\t') ++p;
    return true;  // number / true / false / null
  }
  // iterate the members of an object: fn(key) must consume the value
  template <class F>
  bool object(F&& fn) {
    if (!eat('{')) return ok = false;
    if (eat('}')) return true;
    do {
      std::string key;
      if (!str(&key) || !eat(':')) return ok = false;
      if (!fn(key)) return ok = false;
    } while (eat(','));
    return eat('}') ? true : (ok = false);
  }
};

}  // namespace

struct kh_bpe {
  std::unordered_map<std::string, int32_t> encoder;  // raw bytes -> id (= merge rank)
  std::vector<std::string> decoder;                  // id -> raw bytes ("" = hole)
  std::vector<std::pair<std::string, int32_t>> special;  // added_tokens, file order
  std::unordered_map<int32_t, std::string> special_by_id;
  int32_t bos = 0, eos = 0, stop1 = 0, stop2 = 0;
  int32_t num_token = 0;
  int cp_to_byte[0x144];  // GPT-2 code point -> byte (-1: not a byte-level character)
};

namespace {

void build_cp_table(int* t) {
  for (int i = 0; i < 0x144; ++i) t[i] = -1;
  bool direct[256] = {false};
  for (int ch = 0x21; ch <= 0x7E; ++ch) direct[ch] = true;
  for (int ch = 0xA1; ch <= 0xAC; ++ch) direct[ch] = true;
  for (int ch = 0xAE; ch <= 0xFF; ++ch) direct[ch] = true;
  int n = 0;
  for (int ch = 0; ch < 256; ++ch) {
    if (direct[ch]) t[ch] = ch;
    else t[256 + n++] = ch;  // unicode.cpp:169-175
  }
}

int find_special(const kh_bpe* t, const char* name) {
  for (const auto& s : t->special)
    if (s.first == name) return s.second;
  return -1;
}

// PAT_STR (encode.cpp:59-60), leftmost-first like RE2, on the code points cps[b, e):
//   (?i:'s|'t|'re|'ve|'m|'ll|'d) | [^\r\n\p{L}\p{N}]?\p{L}+ | \p{N} | ?[^\s\p{L}\p{N}]+[\r\n]* |
//   \s*[\r\n]+ | \s+(?:$|[^\S]) | \s+
// returns the end (exclusive) of the piece that starts at p; p < e.
size_t scan_piece(const std::vector<uint32_t>& cps, const std::vector<uint8_t>& cls, size_t p, size_t e) {
  auto lower = [&](size_t i) -> uint32_t {
    const uint32_t c = cps[i];
    if (c >= 'A' && c <= 'Z') return c + 32;
    if (c == 0x17F) return 's';  // RE2 case folding: LATIN SMALL LETTER LONG S folds to s
    return c;
  };
  // 1. contractions
  if (cps[p] == '\'' && p + 1 < e) {
    const uint32_t a = lower(p + 1);
    if (a == 's' || a == 't' || a == 'm' || a == 'd') return p + 2;
    if (p + 2 < e) {
      const uint32_t b = lower(p + 2);
      if ((a == 'r' && b == 'e') || (a == 'v' && b == 'e') || (a == 'l' && b == 'l')) return p + 3;
    }
  }
  // 2. optional one non-letter/number/CR/LF, then letters
  {
    size_t q = p;
    if (cls[q] != C_L && cls[q] != C_N && cls[q] != C_NL && q + 1 < e && cls[q + 1] == C_L) ++q;
    if (cls[q] == C_L) {
      while (q < e && cls[q] == C_L) ++q;
      return q;
    }
  }
  // 3. one number character
  if (cls[p] == C_N) return p + 1;
  // 4. optional space, run of "other", then CR/LF*
  {
    size_t q = p;
    if (cps[q] == ' ' && q + 1 < e && cls[q + 1] == C_OTHER) ++q;
    if (cls[q] == C_OTHER) {
      while (q < e && cls[q] == C_OTHER) ++q;
      while (q < e && cls[q] == C_NL) ++q;
      return q;
    }
  }
  // 5.-7. whitespace run W = [p, w)
  size_t w = p;
  while (w < e && (cls[w] == C_SPACE || cls[w] == C_NL)) ++w;
  // 5. \s*[\r\n]+ : up to and including the LAST CR/LF of the run
  for (size_t q = w; q > p; --q)
    if (cls[q - 1] == C_NL) return q;
  // 6. \s+(?:$|[^\S]) consumes the whole run when it ends the text or has >= 2 characters;
  // 7. \s+ takes the whole run otherwise - the same piece either way
  return w > p ? w : p + 1;
}

// tiktoken's byte_pair_encode (tiktoken.h:17-114): ranks = ids
bool bpe_piece(const kh_bpe* t, const std::string& piece, std::vector<int32_t>* out) {
  if (piece.size() == 1) {
    auto it = t->encoder.find(piece);
    if (it == t->encoder.end()) return false;
    out->push_back(it->second);
    return true;
  }
  const int32_t NONE = INT32_MAX;
  struct Part { int32_t start, rank; };
  std::vector<Part> parts(piece.size() + 1);
  for (size_t i = 0; i <= piece.size(); ++i) parts[i] = {(int32_t)i, NONE};
  auto rank_of = [&](size_t start_idx, size_t skip) -> int32_t {
    if (start_idx + skip + 2 < parts.size()) {
      const int32_t s = parts[start_idx].start, e = parts[start_idx + skip + 2].start;
      auto it = t->encoder.find(piece.substr((size_t)s, (size_t)(e - s)));
      if (it != t->encoder.end()) return it->second;
    }
    return NONE;
  };
  for (size_t i = 0; i + 2 < parts.size(); ++i) parts[i].rank = rank_of(i, 0);
  while (parts.size() > 1) {
    int32_t best = NONE;
    size_t bi = 0;
    for (size_t i = 0; i + 1 < parts.size(); ++i)
      if (parts[i].rank < best) {  // strict: leftmost of equal ranks
        best = parts[i].rank;
        bi = i;
      }
    if (best == NONE) break;
    parts[bi].rank = rank_of(bi, 1);
    if (bi > 0) parts[bi - 1].rank = rank_of(bi - 1, 1);
    parts.erase(parts.begin() + (long)bi + 1);
  }
  for (size_t i = 0; i + 1 < parts.size(); ++i) {
    auto it = t->encoder.find(piece.substr((size_t)parts[i].start, (size_t)(parts[i + 1].start - parts[i].start)));
    if (it == t->encoder.end()) return false;  // tiktoken: ranks.at() throws
    out->push_back(it->second);
  }
  return true;
}

bool encode_ordinary(const kh_bpe* t, const std::string& seg, std::vector<int32_t>* out) {
  if (seg.empty()) return true;
  std::vector<uint32_t> cps;
  std::vector<uint8_t> cls;
  std::vector<size_t> off;
  cps.reserve(seg.size());
  for (size_t i = 0; i < seg.size();) {
    size_t len = 1;
    const uint32_t cp = utf8_next(seg, i, &len);
    cps.push_back(cp);
    cls.push_back(classify(cp));
    off.push_back(i);
    i += len;
  }
  off.push_back(seg.size());
  for (size_t p = 0; p < cps.size();) {
    const size_t q = scan_piece(cps, cls, p, cps.size());
    const std::string piece = seg.substr(off[p], off[q] - off[p]);
    auto it = t->encoder.find(piece);
    if (it != t->encoder.end()) out->push_back(it->second);
    else if (!bpe_piece(t, piece, out)) return false;
    p = q;
  }
  return true;
}

int parse_tokenizer_json(kh_bpe* t, const char* data, size_t n) {
  Json j{data, data + n};
  std::vector<std::pair<std::string, int32_t>> vocab;
  bool have_vocab = false;
  const bool ok = j.object([&](const std::string& key) {
    if (key == "added_tokens") {
      if (j.peek('n')) return j.skip();
      if (!j.eat('[')) return false;
      if (j.eat(']')) return true;
      do {
        long id = -1;
        std::string content;
        bool has_id = false, has_c = false;
        if (!j.object([&](const std::string& k) {
              if (k == "id") return has_id = j.integer(&id);
              if (k == "content") return has_c = j.str(&content);
              return j.skip();
            }))
          return false;
        if (has_id && has_c) t->special.emplace_back(content, (int32_t)id);
      } while (j.eat(','));
      return j.eat(']');
    }
    if (key == "model") {
      return j.object([&](const std::string& k) {
        if (k != "vocab") return j.skip();
        have_vocab = true;
        return j.object([&](const std::string& tok) {
          long id = -1;
          if (!j.integer(&id)) return false;
          vocab.emplace_back(tok, (int32_t)id);
          return true;
        });
      });
    }
    return j.skip();
  });
  if (!ok || !j.ok || !have_vocab) return KH_ERR_FORMAT;
  build_cp_table(t->cp_to_byte);
  int32_t max_id = -1;
  for (const auto& kv : vocab) {
    // vocab keys are GPT-2 "byte-level" text: every code point stands for one byte (encode.cpp:84-95)
    std::string raw;
    for (size_t i = 0; i < kv.first.size();) {
      size_t len = 1;
      const uint32_t cp = utf8_next(kv.first, i, &len);
      i += len;
      if (cp >= 0x144 || t->cp_to_byte[cp] < 0) return KH_ERR_UNSUPPORTED;  // not a byte-level vocabulary
      raw += (char)t->cp_to_byte[cp];
    }
    t->encoder[raw] = kv.second;
    if (kv.second > max_id) max_id = kv.second;
  }
  for (const auto& s : t->special) {
    t->special_by_id[s.second] = s.first;
    if (s.second > max_id) max_id = s.second;
  }
  if (max_id < 0 || max_id > (1 << 24)) return KH_ERR_FORMAT;
  t->decoder.assign((size_t)max_id + 1, std::string());
  for (const auto& kv : t->encoder)
    if (kv.second >= 0) t->decoder[(size_t)kv.second] = kv.first;
  t->num_token = (int32_t)(t->encoder.size() + t->special.size());
  return KH_OK;
}

}  // namespace

extern "C" int kh_bpe_create_from_memory(const void* json, int64_t nbytes, int32_t flavor, kh_bpe** out) {
  if (!json || nbytes <= 0 || !out) return KH_ERR_INVALID_ARG;
  if (flavor != KH_BPE_LLAMA3 && flavor != KH_BPE_QWEN2) return KH_ERR_INVALID_ARG;
  *out = nullptr;
  kh_bpe* t = new (std::nothrow) kh_bpe();
  if (!t) return KH_ERR_IO;
  const int rc = parse_tokenizer_json(t, (const char*)json, (size_t)nbytes);
  if (rc != KH_OK) {
    delete t;
    return rc;
  }
  // encode.cpp:97-103 (Llama-3) / :170-176 (Qwen2).  The reference indexes its map with operator[],
  // which yields id 0 for a name the file does not define; reported here as -1 instead.
  const bool q = flavor == KH_BPE_QWEN2;
  t->bos = find_special(t, q ? "<|im_start|>" : "<|begin_of_text|>");
  t->eos = find_special(t, q ? "<|im_end|>" : "<|end_of_text|>");
  t->stop1 = t->eos;
  t->stop2 = find_special(t, q ? "<|endoftext|>" : "<|eot_id|>");
  *out = t;
  return KH_OK;
}

extern "C" int kh_bpe_create_from_file(const char* path, int32_t flavor, kh_bpe** out) {
  if (!path || !out) return KH_ERR_INVALID_ARG;
  *out = nullptr;
  FILE* f = fopen(path, "rb");
  if (!f) return KH_ERR_IO;  // reference: CHECK(f.is_open()) aborts
  std::string buf;
  char tmp[1 << 16];
  size_t n;
  while ((n = fread(tmp, 1, sizeof(tmp), f)) > 0) buf.append(tmp, n);
  fclose(f);
  return kh_bpe_create_from_memory(buf.data(), (int64_t)buf.size(), flavor, out);
}

extern "C" void kh_bpe_destroy(kh_bpe* t) { delete t; }
extern "C" int32_t kh_bpe_vocab_size(const kh_bpe* t) { return t ? t->num_token : -1; }
extern "C" int32_t kh_bpe_bos_id(const kh_bpe* t) { return t ? t->bos : -1; }
extern "C" int32_t kh_bpe_eos_id(const kh_bpe* t) { return t ? t->eos : -1; }
extern "C" int32_t kh_bpe_stop_id(const kh_bpe* t, int32_t which) {
  if (!t) return -1;
  return which == 0 ? t->stop1 : (which == 1 ? t->stop2 : -1);
}

extern "C" int kh_bpe_encode(const kh_bpe* t, const char* utf8, int64_t len, int32_t add_bos,
                             int32_t add_eos, int32_t flags, int32_t* out_ids, int32_t cap,
                             int32_t* n_ids) {
  if (!t || (!utf8 && len > 0) || len < 0 || !n_ids || cap < 0 || (cap > 0 && !out_ids))
    return KH_ERR_INVALID_ARG;
  std::string text(utf8 ? utf8 : "", (size_t)len);
  if (flags & KH_BPE_REF_SPACES) {  // encode.cpp:108-111
    std::string r;
    r.reserve(text.size() + text.size() / 4);
    for (char c : text) {
      if (c == ' ') r += "\xC4\xA0";
      else r += c;
    }
    text.swap(r);
  }
  std::vector<int32_t> ids;
  if (add_bos) ids.push_back(t->bos);
  // tiktoken.h:215-246: cut at every special token (all are allowed), encode the text in between
  size_t pos = 0;
  while (true) {
    size_t at = std::string::npos;
    int32_t sid = -1;
    size_t slen = 0;
    if (!t->special.empty()) {
      for (size_t i = pos; i < text.size() && at == std::string::npos; ++i)
        for (const auto& s : t->special)  // file order = the alternation order of the reference's regex
          if (!s.first.empty() && text.compare(i, s.first.size(), s.first) == 0) {
            at = i;
            sid = s.second;
            slen = s.first.size();
            break;
          }
    }
    const std::string seg = text.substr(pos, at == std::string::npos ? std::string::npos : at - pos);
    if (!encode_ordinary(t, seg, &ids)) return KH_ERR_FORMAT;  // a byte the vocabulary does not hold
    if (at == std::string::npos) break;
    ids.push_back(sid);
    pos = at + slen;
  }
  if (add_eos) ids.push_back(t->eos);
  *n_ids = (int32_t)ids.size();
  if ((int64_t)ids.size() > cap) return KH_ERR_RANGE;
  if (!ids.empty()) memcpy(out_ids, ids.data(), ids.size() * sizeof(int32_t));
  return KH_OK;
}

extern "C" int kh_bpe_decode(const kh_bpe* t, const int32_t* ids, int32_t n, int32_t flags,
                             char* out_utf8, int64_t cap, int64_t* out_len) {
  if (!t || (n > 0 && !ids) || n < 0 || !out_len || cap < 0 || (cap > 0 && !out_utf8))
    return KH_ERR_INVALID_ARG;
  std::string s;
  for (int32_t i = 0; i < n; ++i) {
    const int32_t id = ids[i];
    auto sp = t->special_by_id.find(id);
    if (id >= 0 && (size_t)id < t->decoder.size() && t->encoder.count(t->decoder[(size_t)id]) &&
        t->encoder.at(t->decoder[(size_t)id]) == id) {
      s += t->decoder[(size_t)id];
    } else if (sp != t->special_by_id.end()) {
      s += sp->second;
    } else {
      return KH_ERR_RANGE;  // tiktoken.h:262: unknown token
    }
  }
  if (flags & KH_BPE_REF_SPACES) {  // encode.cpp:124-126
    std::string r;
    r.reserve(s.size());
    for (size_t i = 0; i < s.size(); ++i) {
      if ((unsigned char)s[i] == 0xC4 && i + 1 < s.size() && (unsigned char)s[i + 1] == 0xA0) {
        r += ' ';
        ++i;
      } else {
        r += s[i];
      }
    }
    s.swap(r);
  }
  *out_len = (int64_t)s.size();
  if ((int64_t)s.size() > cap) return KH_ERR_RANGE;
  if (!s.empty()) memcpy(out_utf8, s.data(), s.size());
  return KH_OK;
}
