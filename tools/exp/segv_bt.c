// Debug aid (not product): LD_PRELOAD-able SIGSEGV handler that prints the native backtrace (glibc backtrace()) before the
// process dies -- which frame of the HIP runtime a segfault inside hipGraphLaunch comes from.  gcc -shared -fPIC -o segv_bt.so segv_bt.c
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>

static void handler(int sig, siginfo_t* si, void* ctx) {
  void* frames[96];
  const char msg[] = "\n==== segv_bt: native backtrace ====\n";
  write(2, msg, sizeof(msg) - 1);
  int n = backtrace(frames, 96);
  backtrace_symbols_fd(frames, n, 2);
  char buf[128];
  int k = snprintf(buf, sizeof(buf), "==== fault address %p ====\n", si->si_addr);
  write(2, buf, k);
  signal(sig, SIG_DFL);
  raise(sig);
}

__attribute__((constructor)) static void install(void) {
  static char stack[1 << 16];
  stack_t ss = {.ss_sp = stack, .ss_size = sizeof(stack), .ss_flags = 0};
  sigaltstack(&ss, 0);
  struct sigaction sa;
  memset(&sa, 0, sizeof(sa));
  sa.sa_sigaction = handler;
  sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
  sigaction(SIGSEGV, &sa, 0);
}
