// GameState -- a game in progress: the current Position, komi / rule / handicap, and the move log the
// encoder (last 8 boards), superko test and undo need.
//
// Mirrors the reference's `class GameState` (src/game/game_state.h:10-238, game_state.cc) member for
// member where the search, the encoder and the self-play loop call it.  What differs is the storage: the
// reference keeps a `shared_ptr<Board>` per move and deep-copies that vector for every playout
// (search.cc:56-58); here the log is a persistent list -- an immutable shared prefix plus a private tail
// of 0.4 KB frames -- so forking a state for a playout costs one Position memcpy and one pointer copy.
#pragma once

#include <array>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "position.h"

namespace sayuri_go {

// What later code needs to know about a past board.
struct Frame {
    std::uint8_t stones[kMaxPoints]; // Color per intersection
    std::int16_t last_move;          // the board's LastMove() (kNoVertex for the initial position)
    std::int16_t move_vertex;        // the move that produced it ...
    std::int8_t move_color;          // ... and its colour (frame 0: unused)
    std::uint64_t ko_hash;
};

class FrameLog {
public:
    int size() const { return base_len_ + static_cast<int>(tail_.size()); }
    const Frame& operator[](int i) const { return i < base_len_ ? (*base_)[i] : tail_[i - base_len_]; }
    void push(const Frame& f) { tail_.push_back(f); }
    void truncate(int n);
    void clear();
    void Freeze(); // fold the tail into a new shared prefix (call before forking many copies)

private:
    std::shared_ptr<const std::vector<Frame>> base_;
    int base_len_ = 0;
    std::vector<Frame> tail_;
};

class GameState {
public:
    Position board_;

    GameState() { Reset(kMaxBoard, 7.5f, kAreaScoring); }
    void Reset(int boardsize, float komi, int scoring); // game_state.cc:15-33
    void SetBoardSize(int boardsize);
    void ClearBoard();

    bool AppendMove(int vtx, int color); // set-up stone, game_state.cc:57-87
    bool PlayMove(int vtx) { return PlayMove(vtx, GetToMove()); }
    bool PlayMove(int vtx, int color);   // game_state.cc:93-120
    bool UndoMove();                     // game_state.cc:122-136
    void SetKomi(float komi);            // integer or half komi only, game_state.cc:312-340
    void SetToMove(int color) { board_.SetToMove(color); }
    void SetWinner(int w) { winner_ = w; }
    void SetRule(int scoring);
    void SetHandicap(int h) { handicap_ = h; }
    void SetTerritoryHelper(const std::vector<int>& ownership) {
        for (size_t i = 0; i < ownership.size() && i < territory_helper_.size(); ++i) territory_helper_[i] = ownership[i];
    }
    bool SetFixedHandicap(int handicap); // game_state.cc:394-452
    bool PlayHandicapStones(const std::vector<int>& vertices, bool kata_like_style); // game_state.cc:476-505

    float GetFinalScore(int color) const { return FinalScoreWith(color, territory_helper_.data()); }
    float GetFinalScore(int color, const std::vector<int>& territory_helper) const { return FinalScoreWith(color, territory_helper.data()); }
    std::vector<bool> GetStrictSafeArea() const;
    // The pass-alive analysis of the current stones, computed once per position: the encoder and the node expansion of the
    // same leaf both need it (the reference runs it twice: encoder.cc:206-262 and node.cc:150).  Keyed by the stones-only
    // Zobrist hash; the results are Position::SafeArea(false) / Position::ScoreAndSafeArea bit for bit.
    void SafeAreaCached(bool* safe) const;
    void ScoreAndSafeAreaCached(int* owner, bool* safe) const;
    std::vector<int> GetOwnership() const;    // pass-alive aware Tromp-Taylor owner per intersection
    std::vector<int> GetRawOwnership() const; // plain Tromp-Taylor reach
    void RemoveDeadStrings(const std::vector<int>& dead) { board_.RemoveMarked(dead.data(), static_cast<int>(dead.size())); }

    bool IsGameOver() const { return winner_ != kUndecided || GetPasses() >= 2; }
    bool IsSuperko() const; // game_state.cc:366-373
    bool IsLegalMove(int vtx) const { return board_.IsLegal(vtx, GetToMove()); }
    bool IsLegalMove(int vtx, int color) const { return board_.IsLegal(vtx, color); }
    bool IsNeighborColor(int vtx, int color) const { return board_.IsNeighbourColor(vtx, color); }
    bool IsSeki(int vtx) const { return board_.IsSeki(vtx); }

    float GetPenalty() const { return GetPenalty(scoring_); }
    float GetPenalty(int scoring) const;                               // game_state.cc:694-706
    float GetPenaltyOffset(int new_scoring, int old_scoring) const;    // game_state.cc:708-722
    float GetKomiWithPenalty() const { return GetKomi() + GetPenalty(); }
    float GetKomi() const;
    float GetWave() const; // game_state.cc:870-903

    int GetWinner() const { return winner_; }
    int GetHandicap() const { return handicap_; }
    int GetMoveNumber() const { return move_number_; }
    int GetBoardSize() const { return board_.BoardSize(); }
    int GetNumIntersections() const { return board_.NumPoints(); }
    int GetNumVertices() const { return board_.NumVertices(); }
    int GetToMove() const { return board_.ToMove(); }
    int GetLastMove() const { return board_.LastMove(); }
    int GetKoMove() const { return board_.KoMove(); }
    int GetPasses() const { return board_.Passes(); }
    int GetScoringRule() const { return scoring_; }
    int GetState(int vtx) const { return board_.At(vtx); }
    int GetLiberties(int vtx) const { return board_.Liberties(vtx); }
    int GetPrisoner(int c) const { return board_.Prisoners(c); }
    std::uint64_t GetKoHash() const { return board_.KoHash(); }
    std::uint64_t GetHash() const { return board_.Hash() ^ komi_hash_ ^ scoring_hash_; }
    std::uint64_t ComputeSymmetryHash(int symm) const { return board_.SymmetryHash(symm) ^ komi_hash_ ^ scoring_hash_; }
    std::uint64_t GetMoveHash(int vtx, int color) const { return board_.MoveHash(vtx, color); }

    int GetVertex(int x, int y) const { return board_.Vertex(x, y); }
    int GetIndex(int x, int y) const { return board_.Index(x, y); }
    int GetX(int vtx) const { return board_.X(vtx); }
    int GetY(int vtx) const { return board_.Y(vtx); }
    int IndexToVertex(int idx) const { return board_.IndexToVertex(idx); }
    int VertexToIndex(int vtx) const { return board_.VertexToIndex(vtx); }

    int TextToVertex(const std::string& text) const; // "q16", "pass", "resign"
    std::string VertexToText(int vtx) const;
    std::string VertexToSgf(int vtx) const;
    std::vector<int> GetAppendMoves(int color) const;

    // ---- move log
    // Past(p): the board p moves ago, p = 0 is the current one (GetPastBoard, game_state.cc:817-820).
    const Frame& Past(int p) const { return log_[move_number_ - p]; }
    // (vertex, colour) of move i, 1-based.
    std::pair<int, int> MoveAt(int i) const { return {log_[i].move_vertex, log_[i].move_color}; }
    void Freeze() { log_.Freeze(); }

private:
    void PushFrame(int vtx, int color);
    float FinalScoreWith(int color, const int* territory_helper) const;

    FrameLog log_; // log_[i] = board after move i (0 = start position incl. set-up stones)
    std::vector<std::pair<int, int>> setup_; // AppendMove stones
    std::array<int, kMaxPoints> territory_helper_; // inline: forking a state allocates nothing
    std::uint8_t scoring_ = kAreaScoring;
    int handicap_ = 0;
    int komi_integer_ = 0;
    bool komi_half_ = false, komi_negative_ = false;
    int move_number_ = 0;
    std::uint64_t komi_hash_ = 0, scoring_hash_ = 0;
    int winner_ = kUndecided;
    // cache of the pass-alive analysis (bit sets: 361 cells; owner as two planes black / white).  Written from const methods
    // (SafeAreaCached / ScoreAndSafeAreaCached, hence Encoder::Planes / Packed on a const GameState&): a GameState belongs to ONE
    // thread at a time -- every game / search owns its states; two threads encoding the same object would race on these fields.
    // The key mixes the board size with the ko hash (AreaKey), so a Reset() to another size cannot hit a stale entry.
    mutable std::uint64_t area_key_ = 0;
    mutable std::uint8_t area_have_ = 0;  // bit 0: safe, bit 1: owner
    mutable std::uint64_t area_safe_[6] = {}, area_black_[6] = {}, area_white_[6] = {};
};

} // namespace sayuri_go
