/* Diagnostic SIGSEGV/SIGABRT handler (tools/diag_pmc2.py): prints lib(+offset) frames of the faulting thread with glibc's
 * backtrace_symbols_fd, then the executable mappings, so that the frames can be resolved with addr2line/nm off the box. */
#define _GNU_SOURCE
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>
#include <sys/syscall.h>
static void handler(int sig, siginfo_t *si, void *uc)
{
    void *fr[64];
    char buf[256];
    int n = snprintf(buf, sizeof buf, "\n[segv_trace] signal %d addr %p tid %ld\n", sig, si ? si->si_addr : 0, (long)syscall(SYS_gettid));
    write(2, buf, n);
    int k = backtrace(fr, 64);
    backtrace_symbols_fd(fr, k, 2);
    int fd = open("/proc/self/maps", O_RDONLY);
    if (fd >= 0) {
        static char m[1 << 20];
        long got = 0, r;
        while ((r = read(fd, m + got, sizeof m - 1 - got)) > 0) got += r;
        m[got] = 0;
        write(2, "[segv_trace] r-x mappings:\n", 27);
        for (char *l = strtok(m, "\n"); l; l = strtok(0, "\n"))
            if (strstr(l, " r-xp ")) { write(2, l, strlen(l)); write(2, "\n", 1); }
        close(fd);
    }
    _exit(128 + sig);
}
void segv_trace_install(void)
{
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = handler;
    sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
    sigaction(SIGSEGV, &sa, 0);
    sigaction(SIGBUS, &sa, 0);
    sigaction(SIGABRT, &sa, 0);
}
