/* Debugging aid (not part of the product): print a native backtrace on SIGSEGV / SIGABRT / SIGBUS.
 * Used once in round 2 to locate the hipGraphLaunch crash of round 1 (bench.py, PRCNN_SEGV_BACKTRACE=1).
 * Build: gcc -O1 -g -shared -fPIC -o tools/libsegv_bt.so tools/segv_bt.c */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static void handler(int sig, siginfo_t *info, void *ctx)
{
    void *frames[64];
    const char msg[] = "\n==== segv_bt: fatal signal, native backtrace ====\n";
    (void)info; (void)ctx;
    if (write(2, msg, sizeof(msg) - 1) < 0) { }
    int n = backtrace(frames, 64);
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}

void segv_bt_install(void)
{
    struct sigaction sa;
    void *warm[4];
    backtrace(warm, 4);             /* loads libgcc now, not inside the handler */
    memset(&sa, 0, sizeof(sa));
    sa.sa_sigaction = handler;
    sa.sa_flags = SA_SIGINFO | SA_RESETHAND;
    sigaction(SIGSEGV, &sa, 0);
    sigaction(SIGABRT, &sa, 0);
    sigaction(SIGBUS, &sa, 0);
}
