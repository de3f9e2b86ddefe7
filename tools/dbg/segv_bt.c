// LD_PRELOAD helper for GPU-box debugging: prints a native backtrace on SIGSEGV / SIGABRT / SIGBUS, then re-raises.
// Build: gcc -shared -fPIC -O1 -o tools/dbg/libsegv_bt.so tools/dbg/segv_bt.c
// Use:   LD_PRELOAD=tools/dbg/libsegv_bt.so python -m pytest -p no:faulthandler ...
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>

static void on_fault(int sig, siginfo_t* si, void* ctx) {
  (void)ctx;
  char head[128];
  int n = snprintf(head, sizeof head, "\n[segv_bt] signal %d at address %p - native backtrace:\n", sig, si ? si->si_addr : 0);
  if (n > 0) (void)!write(2, head, (size_t)n);
  void* frames[64];
  int k = backtrace(frames, 64);
  backtrace_symbols_fd(frames, k, 2);
  signal(sig, SIG_DFL);
  raise(sig);
}

__attribute__((constructor)) static void install(void) {
  void* warm[2];
  backtrace(warm, 2);  // loads libgcc now, not inside the handler
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_sigaction = on_fault;
  sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
  sigaction(SIGSEGV, &sa, 0);
  sigaction(SIGBUS, &sa, 0);
  sigaction(SIGABRT, &sa, 0);
}
