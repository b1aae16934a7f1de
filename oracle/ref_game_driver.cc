// oracle/ref_game_driver.cc -- TEST INFRASTRUCTURE ONLY (never part of the product path).
//
// extern "C" taps on the *unmodified* reference game / encoder classes, compiled in THIS container
// only by oracle/Makefile into oracle/_ref/libsayuri_ref.so.  They let tests and the fixture generator
// drive the reference's own GameState (src/game/game_state.h), Board analyses (src/game/board.h) and
// Encoder (src/neural/encoder.h) move by move and read back what the product's engine must reproduce.
// Nothing is copied from the reference: this file only calls its public interfaces.
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "config.h"
#include "game/game_state.h"
#include "game/symmetry.h"
#include "game/zobrist.h"
#include "neural/encoder.h"
#include "utils/random.h"

namespace {
bool g_game_ready = false;
void EnsureTables() {
    if (g_game_ready) return;
    Zobrist::Initialize();
    Symmetry::Get().Initialize();
    g_game_ready = true;
}
inline GameState* G(void* h) { return static_cast<GameState*>(h); }
} // namespace

extern "C" {

void* ref_game_new(int board, float komi, int scoring) {
    EnsureTables();
    auto* g = new GameState();
    g->Reset(board, komi, scoring);
    return g;
}
void* ref_game_clone(void* h) { return new GameState(*G(h)); }
void ref_game_free(void* h) { delete G(h); }

// move: intersection index, board*board = pass, -1 = resign.  color < 0: side to move.
static int ToVertex(GameState* g, int move) {
    if (move < 0) return kResign;
    if (move == g->GetNumIntersections()) return kPass;
    return g->IndexToVertex(move);
}
int ref_game_play(void* h, int move, int color) {
    auto* g = G(h);
    return g->PlayMove(ToVertex(g, move), color < 0 ? g->GetToMove() : color) ? 1 : 0;
}
int ref_game_append(void* h, int move, int color) { return G(h)->AppendMove(ToVertex(G(h), move), color) ? 1 : 0; }
int ref_game_undo(void* h) { return G(h)->UndoMove() ? 1 : 0; }
int ref_game_fixed_handicap(void* h, int n) { return G(h)->SetFixdHandicap(n) ? 1 : 0; }
void ref_game_set_komi(void* h, float komi) { G(h)->SetKomi(komi); }
void ref_game_set_rule(void* h, int scoring) { G(h)->SetRule(scoring); }
void ref_game_set_to_move(void* h, int color) { G(h)->SetToMove(color); }

// info[0..15]: hash, ko_hash, to_move, last_move(idx|N pass|-1 none), ko(idx|-1), passes, prisoners b, w,
// move_number, superko, game_over, handicap, winner, board, scoring, symmetry hash xor over the 8 symmetries
void ref_game_info(void* h, std::uint64_t* info) {
    auto* g = G(h);
    auto idx = [&](int v) -> std::uint64_t {
        if (v == kNullVertex) return static_cast<std::uint64_t>(-1);
        if (v == kPass) return static_cast<std::uint64_t>(g->GetNumIntersections());
        if (v == kResign) return static_cast<std::uint64_t>(-2);
        return static_cast<std::uint64_t>(g->VertexToIndex(v));
    };
    info[0] = g->GetHash();
    info[1] = g->GetKoHash();
    info[2] = g->GetToMove();
    info[3] = idx(g->GetLastMove());
    info[4] = idx(g->GetKoMove());
    info[5] = g->GetPasses();
    info[6] = g->GetPrisoner(kBlack);
    info[7] = g->GetPrisoner(kWhite);
    info[8] = g->GetMoveNumber();
    info[9] = g->IsSuperko();
    info[10] = g->IsGameOver();
    info[11] = g->GetHandicap();
    info[12] = g->GetWinner();
    info[13] = g->GetBoardSize();
    info[14] = g->GetScoringRule();
    std::uint64_t x = 0;
    for (int s = 0; s < 8; ++s) x ^= g->ComputeSymmetryHash(s) * (2 * s + 1);
    info[15] = x;
}
// scalars[0..5]: komi, komi with penalty, wave, final score (black), penalty, penalty offset(area<-territory)
void ref_game_scalars(void* h, float* out) {
    auto* g = G(h);
    out[0] = g->GetKomi();
    out[1] = g->GetKomiWithPenalty();
    out[2] = g->GetWave();
    out[3] = g->GetFinalScore(kBlack);
    out[4] = g->GetPenalty();
    out[5] = g->GetPenaltyOffset(kArea, kTerritory);
}
int ref_game_planes(void* h, int symmetry, int weights_version, float* out) {
    auto p = Encoder::Get().GetPlanes(*G(h), symmetry, weights_version);
    std::memcpy(out, p.data(), p.size() * sizeof(float));
    return static_cast<int>(p.size());
}
// maps[0]: cell colour, [1] legal for side to move (N+1 entries, last = pass), [2] liberties, [3] ladder code
// (0 none, 1 death, 2 escapable, 3 atari, 4 take), [4] strict safe area, [5] ownership, [6] raw ownership,
// [7] seki, [8] bit0 capture / bit1 atari / bit2 escape / bit3 self-atari / bit4 real eye / bit5 simple eye of the
// side to move; each map is (N+1) bytes
void ref_game_maps(void* h, std::uint8_t* out) {
    auto* g = G(h);
    const int n = g->GetNumIntersections();
    const int c = g->GetToMove();
    std::memset(out, 0, static_cast<size_t>(9 * (n + 1)));
    auto ladders = g->board_.GetLadderMap();
    auto safe = g->GetStrictSafeArea();
    auto own = g->GetOwnership();
    auto raw = g->GetRawOwnership();
    for (int i = 0; i < n; ++i) {
        const int v = g->IndexToVertex(i);
        out[0 * (n + 1) + i] = static_cast<std::uint8_t>(g->GetState(v));
        out[1 * (n + 1) + i] = g->IsLegalMove(v, c);
        out[2 * (n + 1) + i] = (g->GetState(v) == kBlack || g->GetState(v) == kWhite) ? static_cast<std::uint8_t>(std::min(g->GetLiberties(v), 255)) : 0;
        int code = 0;
        switch (ladders[i]) {
            case kLadderDeath: code = 1; break;
            case kLadderEscapable: code = 2; break;
            case kLadderAtari: code = 3; break;
            case kLadderTake: code = 4; break;
            default: break;
        }
        out[3 * (n + 1) + i] = static_cast<std::uint8_t>(code);
        out[4 * (n + 1) + i] = safe[i];
        out[5 * (n + 1) + i] = static_cast<std::uint8_t>(own[i]);
        out[6 * (n + 1) + i] = static_cast<std::uint8_t>(raw[i]);
        out[7 * (n + 1) + i] = g->IsSeki(v);
        std::uint8_t t = 0;
        if (g->GetState(v) == kEmpty) {
            const Board& b = g->board_;
            t |= b.IsCaptureMove(v, c) ? 1 : 0;
            t |= b.IsAtariMove(v, c) ? 2 : 0;
            t |= b.IsEscapeMove(v, c) ? 4 : 0;
            t |= b.IsSelfAtariMove(v, c) ? 8 : 0;
            t |= b.IsRealEye(v, c) ? 16 : 0;
            t |= b.IsSimpleEye(v, c) ? 32 : 0;
        }
        out[8 * (n + 1) + i] = t;
    }
    out[1 * (n + 1) + n] = 1;
}
// n raw draws, then n RandFix(range) draws, then n Roulette(prob) draws of one xoroshiro128+ stream.
void ref_rng_stream(std::uint64_t seed, int n, std::uint32_t range, double prob, std::uint64_t* out) {
    Random<kXoroShiro128Plus> rng(seed);
    for (int i = 0; i < n; ++i) out[i] = rng.Generate();
    for (int i = 0; i < n; ++i) out[n + i] = rng.RandFix(range);
    for (int i = 0; i < n; ++i) out[2 * n + i] = rng.Roulette(prob);
}
void ref_game_set_territory_helper_from_ownership(void* h) { G(h)->SetTerritoryHelper(G(h)->GetOwnership()); }

} // extern "C"
