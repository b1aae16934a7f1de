#include "fiber.h"

#include <linux/futex.h>
#include <pthread.h>
#include <sys/mman.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <climits>
#include <cstdint>
#include <cstdlib>
#include <stdexcept>
#include <thread>

#if !defined(__x86_64__)
#error "fiber.cc: the context switch is written for x86-64 System V"
#endif

// void sayuri_fiber_switch(void** save_sp, void* load_sp): save the callee-saved registers and the stack pointer of the
// running context into *save_sp, continue the context whose stack pointer is load_sp.
extern "C" __attribute__((visibility("hidden"))) void sayuri_fiber_switch(void** save_sp, void* load_sp);
asm(R"(
    .text
    .globl sayuri_fiber_switch
    .hidden sayuri_fiber_switch
    .type sayuri_fiber_switch, @function
sayuri_fiber_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    subq $8, %rsp
    stmxcsr (%rsp)
    fnstcw 4(%rsp)
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    ldmxcsr (%rsp)
    fldcw 4(%rsp)
    addq $8, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size sayuri_fiber_switch, .-sayuri_fiber_switch
)");

namespace sayuri_fiber {

struct FiberPool::Fiber {
    std::function<void()> entry;
    void* sp = nullptr;               // saved stack pointer while suspended
    void* stack = nullptr;
    std::size_t stack_bytes = 0;
    const std::atomic<int>* wait_addr = nullptr;  // suspended until *wait_addr != wait_value
    int wait_value = 0;
    bool finished = false;
    void** scheduler_sp = nullptr;    // where the owning thread's context is saved while this fiber runs
    long long stamp = 0;              // a word for the fiber's user (FiberStamp)
};

namespace {
thread_local FiberPool::Fiber* t_current = nullptr;
std::atomic<int> g_wake_epoch{0};

void FutexWait(std::atomic<int>* addr, int expected, long timeout_us) {
    timespec ts{timeout_us / 1000000, (timeout_us % 1000000) * 1000};
    syscall(SYS_futex, reinterpret_cast<int*>(addr), FUTEX_WAIT_PRIVATE, expected, &ts, nullptr, 0);
}
void FutexWakeAll(std::atomic<int>* addr) { syscall(SYS_futex, reinterpret_cast<int*>(addr), FUTEX_WAKE_PRIVATE, INT_MAX, nullptr, nullptr, 0); }

extern "C" void sayuri_fiber_trampoline();
}  // namespace

// First activation of a fiber "returns" here from sayuri_fiber_switch.
extern "C" __attribute__((visibility("hidden"))) void sayuri_fiber_main() {
    FiberPool::Fiber* f = t_current;
    try {
        f->entry();
    } catch (...) {
        // a game loop reports its failures itself (SelfplayPipe::Run catches inside the entry); nothing may unwind
        // through the hand-made frame below
    }
    f->finished = true;
    void* dummy;
    sayuri_fiber_switch(&dummy, *f->scheduler_sp);  // never comes back
    std::abort();
}
asm(R"(
    .text
    .type sayuri_fiber_trampoline, @function
sayuri_fiber_trampoline:
    call sayuri_fiber_main
    ud2
    .size sayuri_fiber_trampoline, .-sayuri_fiber_trampoline
)");

bool InFiber() { return t_current != nullptr; }
long long* FiberStamp() { return t_current ? &t_current->stamp : nullptr; }

void WaitWhileEqual(const std::atomic<int>* addr, int value) {
    FiberPool::Fiber* f = t_current;
    if (!f) throw std::logic_error("WaitWhileEqual outside a fiber");
    if (addr->load(std::memory_order_acquire) != value) return;
    f->wait_addr = addr;
    f->wait_value = value;
    sayuri_fiber_switch(&f->sp, *f->scheduler_sp);
    // resumed by the scheduler thread (always the same one: t_current is still valid)
}

void NotifyAll() {
    g_wake_epoch.fetch_add(1, std::memory_order_release);
    FutexWakeAll(&g_wake_epoch);
}

FiberPool::~FiberPool() {
    for (Fiber* f : fibers_) {
        if (f->stack) munmap(f->stack, f->stack_bytes);
        delete f;
    }
}

void FiberPool::Add(std::function<void()> entry) {
    Fiber* f = new Fiber;
    f->entry = std::move(entry);
    fibers_.push_back(f);
}

void FiberPool::Run(int threads, const std::function<void(int)>& on_thread_start) {
    if (fibers_.empty()) return;
    threads = std::max(1, std::min<int>(threads, static_cast<int>(fibers_.size())));
    const long page = sysconf(_SC_PAGESIZE);
    for (Fiber* f : fibers_) {
        // one guard page below the stack: an overflow faults instead of scribbling over a neighbour
        f->stack_bytes = (stack_bytes_ + page - 1) / page * page + page;
        void* m = mmap(nullptr, f->stack_bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (m == MAP_FAILED) throw std::runtime_error("fiber stack mmap failed");
        mprotect(m, page, PROT_NONE);
        f->stack = m;
        // initial frame: [mxcsr|fpucw][r15 r14 r13 r12 rbx rbp][return address = trampoline]; after the `ret` the stack
        // pointer is 16-byte aligned, the trampoline's `call` then gives sayuri_fiber_main the alignment of any callee
        std::uintptr_t top = (reinterpret_cast<std::uintptr_t>(m) + f->stack_bytes) & ~std::uintptr_t(15);
        std::uint64_t* spw = reinterpret_cast<std::uint64_t*>(top);
        *--spw = reinterpret_cast<std::uint64_t>(&sayuri_fiber_trampoline);  // ret target; rsp after the ret = top (16-aligned)
        for (int i = 0; i < 6; ++i) *--spw = 0;                            // rbp rbx r12 r13 r14 r15
        *--spw = (std::uint64_t(0x037f) << 32) | 0x1f80;                   // x87 control word | mxcsr (defaults)
        f->sp = spw;
    }
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t)
        th.emplace_back([this, t, threads, &on_thread_start] {
            pthread_setname_np(pthread_self(), "sayuri-games");  // (what a profile or `top -H` shows)
            if (on_thread_start) on_thread_start(t);
            std::vector<Fiber*> mine;
            for (std::size_t i = static_cast<std::size_t>(t); i < fibers_.size(); i += static_cast<std::size_t>(threads)) mine.push_back(fibers_[i]);
            void* my_sp = nullptr;
            for (Fiber* f : mine) f->scheduler_sp = &my_sp;
            std::size_t alive = mine.size();
            while (alive > 0) {
                const int epoch = g_wake_epoch.load(std::memory_order_acquire);
                bool ran = false;
                for (Fiber* f : mine) {
                    if (f->finished) continue;
                    if (f->wait_addr) {
                        if (f->wait_addr->load(std::memory_order_acquire) == f->wait_value) continue;
                        f->wait_addr = nullptr;
                    }
                    t_current = f;
                    sayuri_fiber_switch(&my_sp, f->sp);
                    t_current = nullptr;
                    ran = true;
                    if (f->finished) --alive;
                }
                // nothing was runnable: sleep until the pump reports a finished batch or a re-opened staging set (the
                // timeout only bounds the damage of a word that changes without a NotifyAll)
                if (!ran && alive > 0) FutexWait(&g_wake_epoch, epoch, 1000);
            }
        });
    for (auto& x : th) x.join();
}

}  // namespace sayuri_fiber
