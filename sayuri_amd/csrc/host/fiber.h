// fiber.h -- M:N scheduling of self-play games: stackful coroutines on a small pool of OS threads.
//
// The reference parks one OS thread per game on a condition variable while its leaf is in a batch
// (src/selfplay/pipe.cc:235-296 + src/neural/batch_forward_pipe.cc:7-70).  That works up to about a thousand games;
// beyond it the threads themselves are the cost (measured here on one MI355X, 20b x 256: 59.5 k evals/s at 1024
// games, 18.5 k at 2048, 7.4 k at 4096, profiles/r02_selfplay_g*.json).  A game is a sequential program that blocks in
// exactly one place -- NetworkForwardPipe::Forward -- so it can just as well be a coroutine: the forward pipe, when it
// is called from a fiber, hands the request to the batching queue and SWITCHES to the next runnable game of the same
// OS thread instead of sleeping; the pump flips the request's flag when the batch is back and bumps one wake word.
// The plugin interface stays what it is (a blocking Forward), the games do not know they are fibers.
//
// x86-64 System V only (the hosts of this GPU pool); the context switch saves the callee-saved registers and the stack
// pointer, nothing else (no signal mask: swapcontext's sigprocmask syscall would cost more than the switch).
#pragma once

#include <atomic>
#include <cstddef>
#include <functional>
#include <vector>

namespace sayuri_fiber {

// True while the calling code runs on a fiber of a FiberPool.
bool InFiber();
long long* FiberStamp();  // a word of per-fiber storage (nullptr outside a fiber); the pipe's trace keeps a time stamp there
// Suspend the calling fiber until *addr != value (checked by its scheduler thread whenever it looks for work).
// Must only be called when InFiber().
void WaitWhileEqual(const std::atomic<int>* addr, int value);
// To be called by whoever changes a word fibers may be waiting on (the pump, once per finished batch): wakes the
// scheduler threads that went to sleep because none of their fibers was runnable.
void NotifyAll();

class FiberPool {
public:
    explicit FiberPool(std::size_t stack_bytes = std::size_t(1) << 20) : stack_bytes_(stack_bytes) {}
    ~FiberPool();
    FiberPool(const FiberPool&) = delete;
    FiberPool& operator=(const FiberPool&) = delete;
    void Add(std::function<void()> entry);  // before Run()
    // Runs every fiber to completion on `threads` OS threads (fiber i lives on thread i % threads); `on_thread_start(t)`
    // runs first on each of them (pinning, arena warm-up).  Returns when all fibers have finished.
    void Run(int threads, const std::function<void(int)>& on_thread_start = {});
    std::size_t size() const { return fibers_.size(); }

    struct Fiber;  // opaque

private:
    std::vector<Fiber*> fibers_;
    std::size_t stack_bytes_;
};

}  // namespace sayuri_fiber
