// TEST-ONLY stand-in for <sentencepiece_processor.h> (the external library the reference links, CMakeLists.txt):
// the five calls op::SpeEncodeLayer makes (kuiper/source/op/encode.cpp:13-57: Load, EncodeAsIds, DecodeIds, bos_id /
// eos_id, GetPieceSize) answered by THIS repo's SentencePiece-BPE tokenizer through the C-ABI (kh_spm_*,
// include/kuiper_hip.h), so that the reference's own encode layer - compiled where it lies - runs on it.
#pragma once
#include <string>
#include <vector>

#include "kuiper_hip.h"

namespace sentencepiece {
namespace util {
enum class StatusCode { kOk = 0, kNotFound = 5, kInternal = 13 };
class Status {
 public:
  Status() = default;
  explicit Status(StatusCode c) : c_(c) {}
  StatusCode code() const { return c_; }
  bool ok() const { return c_ == StatusCode::kOk; }

 private:
  StatusCode c_ = StatusCode::kOk;
};
}  // namespace util

class SentencePieceProcessor {
 public:
  SentencePieceProcessor() = default;
  SentencePieceProcessor(const SentencePieceProcessor&) = delete;
  SentencePieceProcessor& operator=(const SentencePieceProcessor&) = delete;
  ~SentencePieceProcessor() {
    if (t_) kh_spm_destroy(t_);
  }
  util::Status Load(const std::string& path) {
    if (t_) kh_spm_destroy(t_);
    t_ = nullptr;
    return util::Status(kh_spm_create_from_file(path.c_str(), &t_) == KH_OK ? util::StatusCode::kOk
                                                                              : util::StatusCode::kNotFound);
  }
  std::vector<int> EncodeAsIds(const std::string& s) const {
    std::vector<int> ids(s.size() + 8);
    int32_t n = 0;
    int rc = kh_spm_encode(t_, s.data(), (int64_t)s.size(), 0, 0, ids.data(), (int32_t)ids.size(), &n);
    if (rc == KH_ERR_RANGE) {
      ids.resize((size_t)n);
      rc = kh_spm_encode(t_, s.data(), (int64_t)s.size(), 0, 0, ids.data(), (int32_t)ids.size(), &n);
    }
    ids.resize(rc == KH_OK ? (size_t)n : 0);
    return ids;
  }
  std::string DecodeIds(const std::vector<int>& ids) const {
    std::string out(16 * ids.size() + 16, '\0');
    int64_t len = 0;
    int rc = kh_spm_decode(t_, ids.data(), (int32_t)ids.size(), &out[0], (int64_t)out.size(), &len);
    if (rc == KH_ERR_RANGE) {
      out.assign((size_t)len + 1, '\0');
      rc = kh_spm_decode(t_, ids.data(), (int32_t)ids.size(), &out[0], (int64_t)out.size(), &len);
    }
    out.resize(rc == KH_OK ? (size_t)len : 0);
    return out;
  }
  int bos_id() const { return kh_spm_bos_id(t_); }
  int eos_id() const { return kh_spm_eos_id(t_); }
  int GetPieceSize() const { return kh_spm_vocab_size(t_); }

 private:
  kh_spm* t_ = nullptr;
};
}  // namespace sentencepiece
