// Go engine, shared basics: vertex geometry, colours, RNG, Zobrist keys, board symmetries.
//
// Behavioural contract (what must agree with the reference bit-for-bit, because hashes feed the NN
// cache / superko and the RNG stream decides fixed-seed move parity):
//   * vertex numbering: letter-box of (board+2)^2, vertex = (y+1)*(board+2)+(x+1); pass/resign/null
//     codes                                        -- reference src/game/types.h:9-21, board.h:541-560
//   * SplitMix64 seeding + xoroshiro128+ stream, RandFix / Roulette reductions
//                                                  -- reference src/utils/random.cc:8-15,42-79, random.h:68-95
//   * Zobrist table draw order from seed 0xabcdabcd12345678
//                                                  -- reference src/game/zobrist.cc:30-72, zobrist.h:14-27
//   * the 8 dihedral symmetries (bit2 = transpose, bit1 = mirror x, bit0 = mirror y)
//                                                  -- reference src/game/symmetry.cc:95-123
#pragma once

#include <atomic>
#include <array>
#include <cstdint>
#include <cstddef>

namespace sayuri_go {

constexpr int kMaxBoard = 19;
constexpr int kMaxLetter = kMaxBoard + 2;
constexpr int kMaxPoints = kMaxBoard * kMaxBoard;     // 361 intersections
constexpr int kMaxVertices = kMaxLetter * kMaxLetter; // 441 letter-box cells
constexpr int kMinBoard = 2;

constexpr int kNoVertex = 0;
constexpr int kPassMove = kMaxVertices + 1;
constexpr int kResignMove = kMaxVertices + 2;

enum Color : std::uint8_t { kBlack = 0, kWhite = 1, kEmpty = 2, kWall = 3 };
inline int Opp(int c) { return c ^ 1; }

enum Scoring : std::uint8_t { kAreaScoring = 0, kTerritoryScoring = 1 };
enum Winner : std::uint8_t { kBlackWon = 0, kWhiteWon = 1, kDrawGame = 2, kUndecided = 3 };

// Ladder codes of one intersection (encoder planes 34-37).
enum LadderMark : std::uint8_t { kLadderNone = 0, kLadderDeath, kLadderEscapable, kLadderAtari, kLadderTake };

// ---------------------------------------------------------------------------------------------
// xoroshiro128+ seeded through SplitMix64.  One explicit stream object; the engine owns one per
// search/game thread (the reference keeps a thread_local pair shared by every Random<> instance of
// the thread, random.h:57 -- an explicit object gives the same stream without the hidden sharing).
class Rng {
public:
    explicit Rng(std::uint64_t seed = 0) { Seed(seed); }
    void Seed(std::uint64_t seed);
    std::uint64_t Next();
    std::uint32_t Below(std::uint32_t range) { // [0, range)
        return static_cast<std::uint32_t>(((Next() >> 32) * static_cast<std::uint64_t>(range)) >> 32);
    }
    bool Chance(double prob);
    // std::uniform_random_bit_generator interface (libstdc++ distributions consume it the same way
    // they consume the reference's generator: full 64-bit range).
    using result_type = std::uint64_t;
    static constexpr result_type min() { return 0; }
    static constexpr result_type max() { return ~static_cast<result_type>(0); }
    result_type operator()() { return Next(); }

private:
    std::uint64_t s_[2];
};

// ---------------------------------------------------------------------------------------------
struct ZobristKeys {
    static constexpr std::uint64_t kEmptyBoard = 0x1234567887654321ULL;
    static constexpr std::uint64_t kBlackToMove = 0xabcdabcdabcdabcdULL;
    static constexpr std::uint64_t kHalfKomi = 0x5678876556788765ULL;
    static constexpr std::uint64_t kNegativeKomi = 0x4321432143214321ULL;

    std::uint64_t state[4][kMaxVertices];
    std::uint64_t prisoner[2][kMaxVertices * 2];
    std::uint64_t ko[kMaxVertices];
    std::uint64_t pass[5];
    std::uint64_t rule[2];
    std::uint64_t komi[kMaxVertices];

    // built once, thread-safe; after that an inlined pointer read (stone placement asks for the table at every call)
    static const ZobristKeys& Get() {
        const ZobristKeys* k = ready_.load(std::memory_order_acquire);
        return k ? *k : Build();
    }

private:
    static const ZobristKeys& Build();
    static std::atomic<const ZobristKeys*> ready_;
};

// ---------------------------------------------------------------------------------------------
class SymmetryTables {
public:
    static constexpr int kCount = 8;
    static constexpr int kIdentity = 0;
    static const SymmetryTables& Get();
    int Index(int board, int symm, int idx) const { return idx_[board][symm][idx]; }
    int Vertex(int board, int symm, int vtx) const { return vtx_[board][symm][vtx]; }

private:
    SymmetryTables();
    std::uint16_t idx_[kMaxBoard + 1][kCount][kMaxPoints];
    std::uint16_t vtx_[kMaxBoard + 1][kCount][kMaxVertices];
};

// ---------------------------------------------------------------------------------------------
// index (y * size + x) <-> vertex ((y + 1) * (size + 2) + x + 1) per board size: the board analyses walk the board by
// index and look the vertex up -- a table read instead of a division and a modulo by a run-time size in every loop
struct IndexTables {
    static const IndexTables& Get();
    std::int16_t i2v[kMaxBoard + 1][kMaxPoints];
    std::int16_t v2i[kMaxBoard + 1][kMaxVertices];  // -1 off the board

private:
    IndexTables();
};

} // namespace sayuri_go
