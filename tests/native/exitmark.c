/* Test helper (tests/test_host_cpu.py): a C-level exit handler, the way a profiler's tool library flushes its output.
 * Loading the library registers the handler; it writes one line to stdout when the process leaves through exit(). */
#include <stdlib.h>
#include <unistd.h>

static void mark(void) {
    static const char msg[] = "C-LEVEL-EXIT-HANDLER\n";
    ssize_t r = write(1, msg, sizeof(msg) - 1);
    (void)r;
}

__attribute__((constructor)) static void init(void) { atexit(mark); }
