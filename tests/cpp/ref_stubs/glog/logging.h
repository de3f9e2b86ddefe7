// TEST-ONLY stand-in for <glog/logging.h>: just enough for the reference's headers and the host
// sources behind tensor::Tensor (LOG(severity) << ..., CHECK*(...) << ...).
#pragma once
#include <unistd.h>  // the real glog header brings it in; model.cpp:84 calls close() relying on that
#include <cstdlib>
#include <iostream>
#include <sstream>
// the real header pulls these in transitively and the reference relies on it (base/alloc.h uses
// std::vector without including <vector>)
#include <map>
#include <string>
#include <vector>
namespace refstub {
struct LogMessage {
  bool fatal;
  std::ostringstream os;
  LogMessage(bool f, const char* file, int line) : fatal(f) { os << file << ":" << line << ": "; }
  ~LogMessage() {
    if (fatal) {
      std::cerr << os.str() << std::endl;
      std::abort();
    }
  }
  std::ostream& stream() { return os; }
};
struct Voidify {
  void operator&(std::ostream&) {}
};
constexpr bool kINFO = false, kWARNING = false, kERROR = false, kFATAL = true;
}  // namespace refstub
#define LOG(sev) refstub::LogMessage(refstub::k##sev, __FILE__, __LINE__).stream()
#define CHECK(c) \
  (c) ? (void)0 : refstub::Voidify() & refstub::LogMessage(true, __FILE__, __LINE__).stream() << "Check failed: " #c " "
#define CHECK_EQ(a, b) CHECK((a) == (b))
#define CHECK_NE(a, b) CHECK((a) != (b))
#define CHECK_LT(a, b) CHECK((a) < (b))
#define CHECK_LE(a, b) CHECK((a) <= (b))
#define CHECK_GT(a, b) CHECK((a) > (b))
#define CHECK_GE(a, b) CHECK((a) >= (b))
// LOG_IF(severity, condition) << ...   (llama3.cpp:591, demo/main.cpp:9)
#define LOG_IF(sev, cond) \
  !(cond) ? (void)0 : refstub::Voidify() & refstub::LogMessage(refstub::k##sev, __FILE__, __LINE__).stream()
