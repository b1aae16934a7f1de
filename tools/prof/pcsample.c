/* pcsample.c -- a tiny statistical profiler for the host side (no perf in the image): LD_PRELOAD it, it samples the
 * program counter of every thread once per millisecond of CPU that thread burns (one CLOCK_THREAD_CPUTIME_ID timer per
 * thread, armed from an interposed pthread_create; a process-wide ITIMER_PROF hands most of its signals to whichever thread
 * sleeps interruptibly -- on the GPU box that made `ioctl` and `futex` the top entries) and prints the hottest symbols
 * (dladdr) at exit.
 *   gcc -O2 -fPIC -shared tools/prof/pcsample.c -o /tmp/pcsample.so -ldl
 *   LD_PRELOAD=/tmp/pcsample.so PCSAMPLE_OUT=/tmp/prof.txt python tools/fake8.py ...                 (measuring tool only)
 * PCSAMPLE_FOCUS=<part of a symbol name> adds a histogram of the samples inside that symbol by offset (16-byte buckets),
 * to be read next to `objdump -d`. */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include <ucontext.h>
#include <unistd.h>
#include <sys/auxv.h>
#include <pthread.h>
#include <sys/syscall.h>
#include <time.h>

#define CAP (1 << 22)
static void** g_pc;
static int* g_tid;
static volatile long g_n;
static __thread int t_tid;

static void on_prof(int sig, siginfo_t* si, void* uc) {
    (void)sig; (void)si;
    long i = __sync_fetch_and_add(&g_n, 1);
    if (i < CAP) {
        g_pc[i] = (void*)((ucontext_t*)uc)->uc_mcontext.gregs[REG_RIP];
        g_tid[i] = t_tid;
    }
}

static void arm_this_thread(void) {
    struct sigevent sev;
    memset(&sev, 0, sizeof sev);
    sev.sigev_notify = SIGEV_THREAD_ID;
    sev.sigev_signo = SIGPROF;
    sev._sigev_un._tid = t_tid = (int)syscall(SYS_gettid);
    timer_t t;
    if (timer_create(CLOCK_THREAD_CPUTIME_ID, &sev, &t) != 0) return;
    struct itimerspec its = {{0, 1000000}, {0, 1000000}};
    timer_settime(t, 0, &its, NULL);
}
static struct { int tid; char comm[32]; void* fn; } g_names[4096];
static volatile int g_nnames;
static __thread void* t_fn;
static void note_name(int tid) {
    char cp[128], comm[32] = "?";
    snprintf(cp, sizeof cp, "/proc/self/task/%d/comm", tid);
    FILE* c = fopen(cp, "r");
    if (!c) return;
    if (fgets(comm, sizeof comm, c)) comm[strcspn(comm, "\n")] = 0;
    fclose(c);
    int k = __sync_fetch_and_add(&g_nnames, 1);
    if (k < 4096) { g_names[k].tid = tid; memcpy(g_names[k].comm, comm, sizeof comm); g_names[k].fn = tid == t_tid ? t_fn : NULL; }
}
struct tramp { void* (*fn)(void*); void* arg; };
static void* thread_entry(void* p) {
    struct tramp t = *(struct tramp*)p;
    free(p);
    t_fn = (void*)t.fn;
    if (g_pc) { arm_this_thread(); note_name(t_tid); }
    void* r = t.fn(t.arg);
    if (g_pc) note_name(t_tid);
    return r;
}
int pthread_create(pthread_t* th, const pthread_attr_t* at, void* (*fn)(void*), void* arg) {
    static int (*real)(pthread_t*, const pthread_attr_t*, void* (*)(void*), void*);
    if (!real) real = (int (*)(pthread_t*, const pthread_attr_t*, void* (*)(void*), void*))dlsym(RTLD_NEXT, "pthread_create");
    struct tramp* t = malloc(sizeof *t);
    t->fn = fn;
    t->arg = arg;
    return real(th, at, thread_entry, t);
}

struct ent { const char* name; const char* file; long n; };
static int cmp(const void* a, const void* b) { long d = ((const struct ent*)b)->n - ((const struct ent*)a)->n; return d > 0 ? 1 : d < 0 ? -1 : 0; }

__attribute__((constructor)) static void start(void) {
    if (!getenv("PCSAMPLE_OUT")) return;
    g_pc = calloc(CAP, sizeof(void*));
    g_tid = calloc(CAP, sizeof(int));
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_prof;
    sa.sa_flags = SA_SIGINFO | SA_RESTART;
    sigaction(SIGPROF, &sa, NULL);
    arm_this_thread();
}
__attribute__((destructor)) static void stop(void) {
    const char* out = getenv("PCSAMPLE_OUT");
    if (!out || !g_pc) return;
    signal(SIGPROF, SIG_IGN);
    long n = g_n < CAP ? g_n : CAP;
    if (n < 100) return;
    struct ent* e = calloc(65536, sizeof *e);
    int ne = 0;
    for (long i = 0; i < n; ++i) {
        Dl_info di;
        const char* nm = "?";
        const char* fl = "?";
        static char unk[4096][48];
        static int nunk;
        unsigned long vdso = getauxval(AT_SYSINFO_EHDR);
        if (vdso && (unsigned long)g_pc[i] >= vdso && (unsigned long)g_pc[i] < vdso + 0x4000) { nm = "[vdso] (clock_gettime ...)"; fl = "vdso"; }
        else if (dladdr(g_pc[i], &di)) {
            if (di.dli_fname) fl = di.dli_fname;
            if (di.dli_sname) nm = di.dli_sname;
            else if (nunk < 4096) {  /* no symbol: bucket by 4 KiB page offset inside the module */
                snprintf(unk[nunk], sizeof unk[0], "+0x%lx", ((unsigned long)g_pc[i] - (unsigned long)di.dli_fbase) & ~0xfful);
                int u;
                for (u = 0; u < nunk; ++u) if (!strcmp(unk[u], unk[nunk])) break;
                nm = unk[u];
                if (u == nunk) ++nunk;
            }
        }
        int k;
        for (k = 0; k < ne; ++k) if (e[k].name == nm || !strcmp(e[k].name, nm)) break;
        if (k == ne && ne < 65536) { e[ne].name = nm; e[ne].file = fl; ++ne; }
        if (k < ne) ++e[k].n;
    }
    qsort(e, ne, sizeof *e, cmp);
    char path[512];
    snprintf(path, sizeof path, "%s.%d", out, (int)getpid());
    FILE* f = fopen(path, "w");
    if (!f) return;
    fprintf(f, "# %ld samples (1 ms of CPU each)\n", n);
    for (int k = 0; k < ne && k < 60; ++k) {
        const char* b = strrchr(e[k].file, '/');
        fprintf(f, "%6.2f%% %8ld  %s  [%s]\n", 100.0 * e[k].n / n, e[k].n, e[k].name, b ? b + 1 : e[k].file);
    }
    {   /* per thread: CPU milliseconds, name, and how much of it sat in ioctl / syscall (futex) */
        static int tids[4096];
        static long tn[4096], tio[4096], tsys[4096];
        int nt = 0;
        for (long i = 0; i < n; ++i) {
            int k;
            for (k = 0; k < nt; ++k) if (tids[k] == g_tid[i]) break;
            if (k == nt) { if (nt == 4096) continue; tids[nt++] = g_tid[i]; }
            ++tn[k];
            Dl_info di;
            if (dladdr(g_pc[i], &di) && di.dli_sname) {
                if (!strcmp(di.dli_sname, "ioctl")) ++tio[k];
                if (!strcmp(di.dli_sname, "syscall")) ++tsys[k];
            }
        }
        fprintf(f, "# threads (tid, comm, samples, in ioctl, in syscall):\n");
        for (int k = 0; k < nt; ++k) {
            if (tn[k] * 200 < n) continue;
            const char* comm = "?";
            note_name(tids[k]);  /* still alive: read it now */
            void* fn = NULL;
            for (int q = 0; q < g_nnames && q < 4096; ++q) if (g_names[q].tid == tids[k]) { comm = g_names[q].comm; if (g_names[q].fn) fn = g_names[q].fn; }
            Dl_info di;
            char where[256] = "";
            if (fn && dladdr(fn, &di)) {
                const char* b = di.dli_fname ? strrchr(di.dli_fname, '/') : NULL;
                snprintf(where, sizeof where, "  started in %s %s+0x%lx", b ? b + 1 : "?", di.dli_sname ? di.dli_sname : "", (unsigned long)fn - (unsigned long)(di.dli_sname ? di.dli_saddr : di.dli_fbase));
            }
            fprintf(f, "  %7d %-18s %7ld %7ld %7ld%s\n", tids[k], comm, tn[k], tio[k], tsys[k], where);
        }
    }
    const char* focus = getenv("PCSAMPLE_FOCUS");
    if (focus && *focus) {
        static long hist[1 << 16];
        long tot = 0;
        const char* full = NULL;
        for (long i = 0; i < n; ++i) {
            Dl_info di;
            if (!dladdr(g_pc[i], &di) || !di.dli_sname || !strstr(di.dli_sname, focus)) continue;
            unsigned long off = ((unsigned long)g_pc[i] - (unsigned long)di.dli_saddr) >> 4;
            if (off < (1 << 16)) { ++hist[off]; ++tot; full = di.dli_sname; }
        }
        fprintf(f, "# focus %s (%s): %ld samples\n", focus, full ? full : "?", tot);
        for (long b = 0; b < (1 << 16); ++b)
            if (hist[b] * 200 >= tot && hist[b]) fprintf(f, "  +0x%04lx %6ld %5.1f%%\n", b << 4, hist[b], 100.0 * hist[b] / tot);
    }
    fclose(f);
}
