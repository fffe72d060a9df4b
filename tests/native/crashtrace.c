/* Test infrastructure (not product code): a SIGABRT / SIGSEGV / SIGBUS handler that writes the NATIVE backtrace of the faulting
 * thread to a file descriptor of our own and then the name of the running test as the last line, and
 * re-raises with the default action.  conftest installs it first and faulthandler on top, so the Python stacks are printed before.  The round-4 driver
 * run of the GPU suite ended in "Aborted (core dumped)" with no native frame on record; with this loaded, an abort names the
 * library and the call chain that raised it (HSA memory-fault handler, std::terminate of a watchdog thread, glibc heap check ...).
 * Built by tests/conftest.py with gcc; loaded with ctypes. */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>
#include <sys/syscall.h>

static int g_fd = 2;
static char g_note[256];

static void put(const char* s) { if (write(g_fd, s, strlen(s)) < 0) {} }

static void handler(int sig, siginfo_t* info, void* uc) {
    void* frames[64];
    char num[32];
    long tid = (long)syscall(SYS_gettid);
    int i = 30; num[31] = 0; num[30] = '\n';
    long v = tid; do { num[--i] = (char)('0' + v % 10); v /= 10; } while (v && i > 0);
    put("\n[bnerv-crashtrace] signal ");
    put(sig == SIGABRT ? "SIGABRT" : sig == SIGSEGV ? "SIGSEGV" : sig == SIGBUS ? "SIGBUS" : "other");
    put(" while in: "); put(g_note); put("  thread tid "); put(num + i);
    int n = backtrace(frames, 64);
    backtrace_symbols_fd(frames, n, g_fd);
    /* the test's name ONCE MORE as the very last line: a log keeper that stores only the tail of the output still learns where the
     * run died (tests/conftest.py installs this handler UNDER faulthandler, whose dump of the Python stacks therefore comes first) */
    put("[bnerv-crashtrace] died in: "); put(g_note); put("\n");
    fsync(g_fd);
    (void)info; (void)uc;
    signal(sig, SIG_DFL);
    raise(sig);
}

void bnerv_crashtrace_note(const char* s) { strncpy(g_note, s, sizeof(g_note) - 1); }

int bnerv_crashtrace_install(int fd) {
    static char stack[1 << 16];
    stack_t ss; ss.ss_sp = stack; ss.ss_size = sizeof(stack); ss.ss_flags = 0;
    sigaltstack(&ss, 0);
    g_fd = fd;
    void* warm[4]; backtrace(warm, 4);              /* loads libgcc now, not inside the handler */
    int sigs[3] = {SIGABRT, SIGSEGV, SIGBUS};
    for (int k = 0; k < 3; ++k) {
        struct sigaction sa; memset(&sa, 0, sizeof(sa));
        sa.sa_sigaction = handler; sa.sa_flags = SA_SIGINFO | SA_ONSTACK; sigemptyset(&sa.sa_mask);
        if (sigaction(sigs[k], &sa, 0) != 0) return -1;
    }
    return 0;
}
