/* LD_PRELOAD helper for the crash hunt (profiles/tools/segv_hunt.py): on SIGSEGV / SIGBUS / SIGABRT write the faulting
 * address, the native backtrace of the faulting thread and /proc/self/maps to $SEGV_TRACE_FILE, then hand the signal
 * on to whoever was installed before (python's faulthandler re-raises into us; we re-raise into the default action,
 * so the core is still written).  Test tooling only: nothing in the product links or loads this.
 *   gcc -O1 -g -shared -fPIC -o segv_trace.so segv_trace.c */
#define _GNU_SOURCE
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <ucontext.h>

static struct sigaction old_segv, old_bus, old_abrt;
static char path[512];

static void
wr(int fd, const char* s)
{
	ssize_t r = write(fd, s, strlen(s));
	(void)r;
}

static void
handler(int sig, siginfo_t* si, void* uc_)
{
	int fd = path[0] ? open(path, O_WRONLY | O_CREAT | O_APPEND, 0644) : 2;
	if (fd < 0)
		fd = 2;
	char line[256];
	ucontext_t* uc = (ucontext_t*)uc_;
	snprintf(line, sizeof line, "\n=== signal %d  si_code %d  fault address %p  rip %p  pid %d tid %ld\n", sig, si ? si->si_code : 0,
	    si ? si->si_addr : NULL, uc ? (void*)uc->uc_mcontext.gregs[REG_RIP] : NULL, (int)getpid(), (long)gettid());
	wr(fd, line);
	void* frames[96];
	int n = backtrace(frames, 96);
	backtrace_symbols_fd(frames, n, fd);
	wr(fd, "=== maps\n");
	int m = open("/proc/self/maps", O_RDONLY);
	if (m >= 0) {
		char buf[8192];
		ssize_t r;
		while ((r = read(m, buf, sizeof buf)) > 0) {
			ssize_t w = write(fd, buf, (size_t)r);
			(void)w;
		}
		close(m);
	}
	if (fd != 2)
		close(fd);
	struct sigaction* old = sig == SIGSEGV ? &old_segv : sig == SIGBUS ? &old_bus : &old_abrt;
	sigaction(sig, old, NULL); /* whoever was there before (usually SIG_DFL: the core) */
	raise(sig);
}

__attribute__((constructor)) static void
install(void)
{
	const char* p = getenv("SEGV_TRACE_FILE");
	if (p)
		snprintf(path, sizeof path, "%s", p);
	void* warm[4];
	backtrace(warm, 4); /* loads libgcc now, not inside the handler */
	static char stack[1 << 16];
	stack_t ss = { .ss_sp = stack, .ss_size = sizeof stack, .ss_flags = 0 };
	sigaltstack(&ss, NULL);
	struct sigaction sa;
	memset(&sa, 0, sizeof sa);
	sa.sa_sigaction = handler;
	sa.sa_flags = SA_SIGINFO | SA_ONSTACK | SA_NODEFER;
	sigaction(SIGSEGV, &sa, &old_segv);
	sigaction(SIGBUS, &sa, &old_bus);
	sigaction(SIGABRT, &sa, &old_abrt);
}
