/* LD_PRELOAD helper: print a C backtrace when the process aborts (a GPU box has no gdb).  gcc -shared -fPIC -o abort_trace.so abort_trace.c */
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <unistd.h>
static void on_abort(int sig) {
    void *bt[64];
    int n = backtrace(bt, 64);
    dprintf(2, "\n==== signal %d: C backtrace ====\n", sig);
    backtrace_symbols_fd(bt, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}
/* also callable later (ctypes): a runtime that installs handlers of its own after start-up would otherwise replace these */
void sct_install_abort_trace(void) {
    signal(SIGABRT, on_abort);
    signal(SIGSEGV, on_abort);
    signal(SIGBUS, on_abort);
}
__attribute__((constructor)) static void init(void) { sct_install_abort_trace(); }
