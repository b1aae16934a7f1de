#include "go_base.h"

#include <algorithm>
#include <utility>
#include <vector>

namespace sayuri_go {

namespace {
inline std::uint64_t Mix64(std::uint64_t z) {
    z += 0x9e3779b97f4a7c15ULL;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
inline std::uint64_t RotL(std::uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
} // namespace

void Rng::Seed(std::uint64_t seed) {
    // two chained SplitMix64 outputs become the 128-bit state (reference random.cc:42-52)
    seed = Mix64(seed);
    s_[0] = seed;
    seed = Mix64(seed);
    s_[1] = seed;
}

std::uint64_t Rng::Next() {
    const std::uint64_t a = s_[0];
    std::uint64_t b = s_[1];
    const std::uint64_t out = a + b;
    b ^= a;
    s_[0] = RotL(a, 55) ^ b ^ (b << 14);
    s_[1] = RotL(b, 36);
    return out;
}

bool Rng::Chance(double prob) {
    prob = std::min(std::max(prob, 0.0), 1.0);
    if (prob <= 0.0) return false;
    if (prob >= 1.0) return true;
    const long double two64 = 18446744073709551616.0L;
    const std::uint64_t threshold = static_cast<std::uint64_t>(prob * two64);
    return Next() <= threshold;
}

// ---------------------------------------------------------------------------------------------
std::atomic<const ZobristKeys*> ZobristKeys::ready_{nullptr};

const ZobristKeys& ZobristKeys::Build() {
    static const ZobristKeys keys = [] {
        ZobristKeys k;
        Rng rng(0xabcdabcd12345678ULL);
        for (;;) {
            std::vector<std::uint64_t> all = {kEmptyBoard, kBlackToMove, kHalfKomi, kNegativeKomi};
            auto draw = [&](std::uint64_t* dst, int n) {
                for (int i = 0; i < n; ++i) all.push_back(dst[i] = rng.Next());
            };
            for (int c = 0; c < 4; ++c) draw(k.state[c], kMaxVertices);
            for (int c = 0; c < 2; ++c) draw(k.prisoner[c], kMaxVertices * 2);
            draw(k.ko, kMaxVertices);
            draw(k.pass, 5);
            draw(k.rule, 2);
            draw(k.komi, kMaxVertices);
            std::sort(all.begin(), all.end());
            if (std::adjacent_find(all.begin(), all.end()) == all.end()) break; // redraw on a collision
        }
        return k;
    }();
    ready_.store(&keys, std::memory_order_release);
    return keys;
}

// ---------------------------------------------------------------------------------------------
SymmetryTables::SymmetryTables() {
    for (auto& b : idx_)
        for (auto& s : b)
            for (auto& v : s) v = 0;
    for (auto& b : vtx_)
        for (auto& s : b)
            for (auto& v : s) v = 0;
    for (int n = kMinBoard; n <= kMaxBoard; ++n) {
        for (int s = 0; s < kCount; ++s) {
            for (int y = 0; y < n; ++y) {
                for (int x = 0; x < n; ++x) {
                    int tx = x, ty = y;
                    if (s & 4) std::swap(tx, ty);
                    if (s & 2) tx = n - 1 - tx;
                    if (s & 1) ty = n - 1 - ty;
                    idx_[n][s][y * n + x] = static_cast<std::uint16_t>(ty * n + tx);
                    vtx_[n][s][(y + 1) * (n + 2) + x + 1] = static_cast<std::uint16_t>((ty + 1) * (n + 2) + tx + 1);
                }
            }
        }
    }
}

const SymmetryTables& SymmetryTables::Get() {
    static const SymmetryTables t;
    return t;
}

IndexTables::IndexTables() {
    for (auto& b : i2v)
        for (auto& v : b) v = 0;
    for (auto& b : v2i)
        for (auto& v : b) v = -1;
    for (int n = kMinBoard; n <= kMaxBoard; ++n)
        for (int y = 0; y < n; ++y)
            for (int x = 0; x < n; ++x) {
                const int i = y * n + x, v = (y + 1) * (n + 2) + x + 1;
                i2v[n][i] = static_cast<std::int16_t>(v);
                v2i[n][v] = static_cast<std::int16_t>(i);
            }
}

const IndexTables& IndexTables::Get() {
    static const IndexTables t;
    return t;
}

} // namespace sayuri_go
