// kh_debug.cpp — the one place the library looks at tuning / test hooks.
//
// Every hook ("KH_SHAPE_QKV", "KH_PREFILL", "KH_PG_CHUNK", ...) lives in a process-wide key -> value
// table.  The table is seeded ONCE, at first use, from the KH_* variables of the process environment
// (so `KH_PREFILL=gemv ./kuiper_demo ...` keeps working), and is changed afterwards only through
// kh_debug_set() - no launch path calls into the C library's environment, and a test that wants another
// mode says so through the API instead of mutating the environment of a running process.
// (The Python binding mirrors os.environ into the table before the calls that read hooks,
// kuiperllama_amd/_ffi.py::sync_env, so `monkeypatch.setenv` in the test-suite still works.)
#include <string.h>

#include <map>
#include <mutex>
#include <set>
#include <string>

#include "../../include/kuiper_hip.h"

extern char** environ;

namespace {
std::mutex g_mu;
// Values are interned and never freed: a pointer handed out by dbg() / kh_debug_get() stays valid for the life of
// the process, whatever another thread sets or clears meanwhile (std::set nodes do not move; the pool grows by
// one string per DISTINCT value ever set - a handful).
const std::string* intern(const std::string& v) {
  static std::set<std::string> pool;
  return &*pool.insert(v).first;
}
std::map<std::string, const std::string*>& table() {
  static std::map<std::string, const std::string*> t = [] {
    std::map<std::string, const std::string*> m;
    for (char** e = environ; e && *e; ++e) {
      if (strncmp(*e, "KH_", 3) != 0) continue;
      const char* eq = strchr(*e, '=');
      if (!eq) continue;
      m.emplace(std::string(*e, (size_t)(eq - *e)), intern(std::string(eq + 1)));
    }
    return m;
  }();
  return t;
}
}  // namespace

namespace khm {
// Value of a hook, or nullptr when it is not set.  The pointer stays valid for the life of the process (interned).
const char* dbg(const char* key) {
  if (!key) return nullptr;
  std::lock_guard<std::mutex> lk(g_mu);
  auto& t = table();
  auto it = t.find(key);
  return it == t.end() ? nullptr : it->second->c_str();
}
}  // namespace khm

extern "C" int kh_debug_set(const char* key, const char* value) {
  if (!key || strncmp(key, "KH_", 3) != 0) return KH_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lk(g_mu);
  auto& t = table();
  if (value)
    t[key] = intern(value);
  else
    t.erase(key);
  return KH_OK;
}

extern "C" const char* kh_debug_get(const char* key) { return khm::dbg(key); }

// Names currently set, '\n'-separated, into buf (always NUL-terminated); returns the number of bytes needed.
extern "C" int64_t kh_debug_list(char* buf, int64_t cap) {
  std::lock_guard<std::mutex> lk(g_mu);
  std::string all;
  for (const auto& kv : table()) {
    all += kv.first;
    all += '\n';
  }
  if (buf && cap > 0) {
    const size_t n = all.size() < (size_t)cap - 1 ? all.size() : (size_t)cap - 1;
    memcpy(buf, all.data(), n);
    buf[n] = 0;
  }
  return (int64_t)all.size() + 1;
}
