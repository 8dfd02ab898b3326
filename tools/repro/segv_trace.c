/* Debug aid: prints the native call stack on SIGSEGV (python's faulthandler shows only the Python frames).
 * Loaded through ctypes when CTR_SEGV_TRACE is set (tests/conftest.py). */
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static void on_segv(int sig, siginfo_t* info, void* ctx) {
  void* frames[64];
  const char msg[] = "\n==== native stack at SIGSEGV ====\n";
  (void)ctx; (void)info;
  if (write(2, msg, sizeof(msg) - 1) < 0) {}
  const int n = backtrace(frames, 64);
  backtrace_symbols_fd(frames, n, 2);
  signal(sig, SIG_DFL);
  raise(sig);
}

int segv_trace_install(void) {
  struct sigaction sa;
  memset(&sa, 0, sizeof(sa));
  sa.sa_sigaction = on_segv;
  sa.sa_flags = SA_SIGINFO | SA_NODEFER;
  sigemptyset(&sa.sa_mask);
  return sigaction(SIGSEGV, &sa, 0);
}
