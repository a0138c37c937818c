/* LD_PRELOAD / dlopen aid (tests/conftest.py loads it when it is built): on SIGABRT print the C call stack of the thread that aborted (who called abort(): the HIP / HSA
 * runtime, glibc's heap checks, libstdc++'s terminate ...), then die as before.  gcc -shared -fPIC -o abort_trace.so abort_trace.c */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

char abort_trace_note[256];             /* whoever loaded this may leave a line here (what it was doing): printed with the stack */
static struct sigaction previous;
static int out_fd = 2;                 /* a copy of stderr as it was when this was loaded: pytest points fd 2 at a capture file
                                          while a test runs, and what is written there dies with the process */

static void on_abort(int sig)
{
    void *buf[64];
    static const char msg[] = "\n=== abort_trace: SIGABRT raised from ===\n";
    (void)!write(out_fd, msg, sizeof(msg) - 1);
    if (abort_trace_note[0]) {
        (void)!write(out_fd, abort_trace_note, strnlen(abort_trace_note, sizeof(abort_trace_note)));
        (void)!write(out_fd, "\n", 1);
    }
    const int n = backtrace(buf, 64);
    backtrace_symbols_fd(buf, n, out_fd);
    /* what the aborting library said before it called abort(): if fd 2 is a capture file now, its last 2 KB */
    const off_t end = lseek(2, 0, SEEK_CUR);
    if (end > 0) {
        static char tail[2048];
        const off_t from = end > (off_t)sizeof(tail) ? end - (off_t)sizeof(tail) : 0;
        const ssize_t got = pread(2, tail, (size_t)(end - from), from);
        if (got > 0) {
            static const char hdr[] = "=== abort_trace: the end of the captured stderr ===\n";
            (void)!write(out_fd, hdr, sizeof(hdr) - 1);
            (void)!write(out_fd, tail, (size_t)got);
            (void)!write(out_fd, "\n", 1);
        }
    }
    sigaction(sig, &previous, NULL);     /* whoever was there before (Python's faulthandler, or the default) goes on from here */
    raise(sig);
}

__attribute__((constructor)) static void install(void)
{
    void *warm[2];
    const int fd = dup(2);
    if (fd >= 0) out_fd = fd;
    backtrace(warm, 2);                 /* loads libgcc now, not inside the handler */
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_handler = on_abort;
    sigaction(SIGABRT, &sa, &previous);
}
