/* pcsample.c -- a tiny statistical profiler for the host side (no perf in the image): LD_PRELOAD it, it samples the
 * program counter of whichever thread burns CPU (ITIMER_PROF, 1 kHz) and prints the hottest symbols (dladdr) at exit.
 *   gcc -O2 -fPIC -shared tools/prof/pcsample.c -o /tmp/pcsample.so -ldl
 *   LD_PRELOAD=/tmp/pcsample.so PCSAMPLE_OUT=/tmp/prof.txt python tools/fake8.py ...                 (measuring tool only) */
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

#define CAP (1 << 22)
static void** g_pc;
static volatile long g_n;

static void on_prof(int sig, siginfo_t* si, void* uc) {
    (void)sig; (void)si;
    long i = __sync_fetch_and_add(&g_n, 1);
    if (i < CAP) g_pc[i] = (void*)((ucontext_t*)uc)->uc_mcontext.gregs[REG_RIP];
}

struct ent { const char* name; const char* file; long n; };
static int cmp(const void* a, const void* b) { long d = ((const struct ent*)b)->n - ((const struct ent*)a)->n; return d > 0 ? 1 : d < 0 ? -1 : 0; }

__attribute__((constructor)) static void start(void) {
    if (!getenv("PCSAMPLE_OUT")) return;
    g_pc = calloc(CAP, sizeof(void*));
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_prof;
    sa.sa_flags = SA_SIGINFO | SA_RESTART;
    sigaction(SIGPROF, &sa, NULL);
    struct itimerval it = {{0, 1000}, {0, 1000}};
    setitimer(ITIMER_PROF, &it, NULL);
}
__attribute__((destructor)) static void stop(void) {
    const char* out = getenv("PCSAMPLE_OUT");
    if (!out || !g_pc) return;
    struct itimerval it = {{0, 0}, {0, 0}};
    setitimer(ITIMER_PROF, &it, NULL);
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
    fclose(f);
}
