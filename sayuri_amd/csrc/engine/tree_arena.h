// tree_arena.h -- memory of one game's search tree.
//
// A move's tree is ~400 nodes of ~7 KB (a node carries a 361-float ownership average; its edge list has up to 362
// entries) that are built in a few milliseconds and dropped when the move is played.  Through glibc malloc that is a
// per-move grow / shrink of the thread's arena; with more than a thousand games in flight the arenas spill into many
// 64 MiB heaps that are mapped and unmapped all the time (measured on the MI355X host: 11 of 16 busy cores in system
// time, 9 k evals/s instead of 63 k at 2048 games).  So a game keeps its tree memory for itself: power-of-two size
// classes with free lists, carved from 1 MiB slabs that are only returned when the game's Search dies.  One game is
// walked by one thread (or fiber) at a time, so there are no locks.
#pragma once

#include <sys/mman.h>

#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <new>
#include <vector>

namespace sayuri_engine {

class TreeArena {
public:
    TreeArena() = default;
    TreeArena(const TreeArena&) = delete;
    TreeArena& operator=(const TreeArena&) = delete;
    ~TreeArena() {
        for (void* s : slabs_) munmap(s, kSlabBytes);
        for (auto& b : big_) munmap(b.first, b.second);
    }

    // `arena` may be null: plain malloc (a Node built outside a Search, e.g. in a unit test)
    static void* Alloc(TreeArena* arena, std::size_t bytes) {
        if (!arena) {
            Header* h = static_cast<Header*>(std::malloc(sizeof(Header) + bytes));
            if (!h) throw std::bad_alloc();
            h->owner = nullptr;
            h->cls = 0;
            return h + 1;
        }
        return arena->AllocImpl(bytes);
    }
    static void Release(void* p) noexcept {
        if (!p) return;
        Header* h = static_cast<Header*>(p) - 1;
        if (!h->owner) {
            std::free(h);
            return;
        }
        h->owner->ReleaseImpl(h);
    }
    std::size_t slab_bytes() const { return slabs_.size() * kSlabBytes; }
    std::size_t live_blocks() const { return live_; }

    // With no live block left: unmap the big blocks and every slab above `keep_bytes`, and start carving the kept slabs
    // from the beginning again.  A no-op while anything is still allocated.
    void Reset(std::size_t keep_bytes) {
        if (live_ != 0) return;
        for (auto& b : big_) munmap(b.first, b.second);
        big_.clear();
        const std::size_t keep = keep_bytes / kSlabBytes;
        while (slabs_.size() > keep) {
            munmap(slabs_.back(), kSlabBytes);
            slabs_.pop_back();
        }
        used_slabs_ = 0;
        for (auto& f : free_) f = nullptr;
        bump_ = nullptr;
        bump_left_ = 0;
    }

private:
    struct Header {
        TreeArena* owner;
        std::uint64_t cls;  // size class; 16-byte header keeps the payload 16-byte aligned
    };
    struct FreeBlock { FreeBlock* next; };
    static constexpr std::size_t kSlabBytes = std::size_t(1) << 20;
    static constexpr int kMinShift = 6, kMaxShift = 14, kClasses = kMaxShift - kMinShift + 1;  // 64 B .. 16 KiB blocks

    void* AllocImpl(std::size_t bytes) {
        const std::size_t need = bytes + sizeof(Header);
        int shift = kMinShift;
        while ((std::size_t(1) << shift) < need && shift <= kMaxShift) ++shift;
        if (shift > kMaxShift) {  // larger than any class: its own mapping, released with the arena
            const std::size_t len = (need + 4095) & ~std::size_t(4095);
            void* m = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            if (m == MAP_FAILED) throw std::bad_alloc();
            big_.emplace_back(m, len);
            ++live_;
            Header* h = static_cast<Header*>(m);
            h->owner = this;
            h->cls = kClasses;  // not recycled
            return h + 1;
        }
        const int c = shift - kMinShift;
        if (!free_[c]) Refill(c, std::size_t(1) << shift);
        FreeBlock* b = free_[c];
        free_[c] = b->next;
        ++live_;
        Header* h = reinterpret_cast<Header*>(b);
        h->owner = this;
        h->cls = static_cast<std::uint64_t>(c);
        return h + 1;
    }
    void ReleaseImpl(Header* h) noexcept {
        --live_;
        if (h->cls >= static_cast<std::uint64_t>(kClasses)) return;  // big blocks are unmapped by Reset / with the arena
        FreeBlock* b = reinterpret_cast<FreeBlock*>(h);
        const int c = static_cast<int>(h->cls);
        b->next = free_[c];
        free_[c] = b;
    }
    void Refill(int c, std::size_t block) {
        if (!bump_ || bump_left_ < block) {
            void* m;
            if (used_slabs_ < slabs_.size()) {
                m = slabs_[used_slabs_];  // a slab kept by Reset
            } else {
                // MAP_POPULATE: the slab's 256 pages in one go under one acquisition of the address-space lock.  Faulted in one
                // by one from hundreds of game threads, the first trees of a run cost ~15 us per page (5 % of a rank's host time
                // over its first minute); a slab is carved to the end anyway.
                m = mmap(nullptr, kSlabBytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_POPULATE, -1, 0);
                if (m == MAP_FAILED) throw std::bad_alloc();
                slabs_.push_back(m);
            }
            ++used_slabs_;
            // (what was left of the previous slab -- less than one block of this class -- is given up)
            bump_ = static_cast<unsigned char*>(m);
            bump_left_ = kSlabBytes;
        }
        // carve a handful of blocks at a time so that one class does not eat a whole slab
        const std::size_t n = std::max<std::size_t>(1, std::min<std::size_t>(bump_left_ / block, 16));
        for (std::size_t i = 0; i < n; ++i) {
            FreeBlock* b = reinterpret_cast<FreeBlock*>(bump_);
            b->next = free_[c];
            free_[c] = b;
            bump_ += block;
            bump_left_ -= block;
        }
    }

    FreeBlock* free_[kClasses] = {};
    unsigned char* bump_ = nullptr;
    std::size_t bump_left_ = 0;
    std::vector<void*> slabs_;
    std::size_t used_slabs_ = 0;  // slabs_[0 .. used_slabs_) have been handed to the bump pointer since the last Reset
    std::size_t live_ = 0;        // blocks handed out and not released yet
    std::vector<std::pair<void*, std::size_t>> big_;
};

// std allocator over a TreeArena (for the edge lists of the nodes)
template <typename T> struct TreeArenaAllocator {
    using value_type = T;
    TreeArena* arena;
    explicit TreeArenaAllocator(TreeArena* a = nullptr) noexcept : arena(a) {}
    template <typename U> TreeArenaAllocator(const TreeArenaAllocator<U>& o) noexcept : arena(o.arena) {}
    T* allocate(std::size_t n) { return static_cast<T*>(TreeArena::Alloc(arena, n * sizeof(T))); }
    void deallocate(T* p, std::size_t) noexcept { TreeArena::Release(p); }
    template <typename U> bool operator==(const TreeArenaAllocator<U>& o) const noexcept { return arena == o.arena; }
    template <typename U> bool operator!=(const TreeArenaAllocator<U>& o) const noexcept { return arena != o.arena; }
};

}  // namespace sayuri_engine
