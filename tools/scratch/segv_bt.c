/* SIGSEGV handler on its own stack that prints the native backtrace (glibc backtrace_symbols_fd) and exits: loaded with
 * ctypes by the capture reproducers to see which library the crash is in (a stack overflow needs the alternate stack).
 * build: gcc -shared -fPIC -O1 -o segv_bt.so segv_bt.c */
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>
static char alt_stack[1 << 16];
static void on_segv(int sig) {
    void* frames[48];
    int n = backtrace(frames, 48);
    backtrace_symbols_fd(frames, n, 2);
    _exit(128 + sig);
}
void install_segv_bt(void) {
    void* warm[2];
    backtrace(warm, 2);                       /* loads libgcc now, not inside the handler */
    stack_t ss;
    ss.ss_sp = alt_stack; ss.ss_size = sizeof(alt_stack); ss.ss_flags = 0;
    sigaltstack(&ss, 0);
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_handler = on_segv;
    sa.sa_flags = SA_ONSTACK;
    sigaction(SIGSEGV, &sa, 0);
}
