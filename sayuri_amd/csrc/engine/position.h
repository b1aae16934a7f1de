// Position -- one Go board: stones, chains with incremental liberty counts, ko, Zobrist hashes,
// plus the board analyses the network encoder and the search consume (ladder reading, Benson
// pass-alive / pass-dead areas, Tromp-Taylor reach, seki test).
//
// Restates the observable behaviour of the reference's `class Board` (src/game/board.h:17-330,
// src/game/board.cc); every result that reaches the network input planes or the search is
// bit-identical, including the iteration-order dependent ones (ladder candidate order follows the
// chain ring order, board.cc:429-482; region ring order follows ClassifyGroups, board.cc:2109-2163).
// It is a plain trivially-copyable value (about 6 KB): a playout forks it with one memcpy, no heap.
#pragma once

#include <cstdint>
#include <cstring>

#include "go_base.h"

namespace sayuri_go {

class Position {
public:
    Position() { Reset(kMaxBoard); }
    void Reset(int board_size); // board.cc:10-77

    // ---- geometry
    int BoardSize() const { return size_; }
    int Letter() const { return letter_; }
    int NumPoints() const { return points_; }
    int NumVertices() const { return vertices_; }
    int Vertex(int x, int y) const { return (y + 1) * letter_ + x + 1; }
    int Index(int x, int y) const { return y * size_ + x; }
    int X(int v) const { return v % letter_ - 1; }
    int Y(int v) const { return v / letter_ - 1; }
    int IndexToVertex(int i) const { return i2v_[i]; }  // tables of this board size (go_base.h IndexTables)
    int VertexToIndex(int v) const { return v2i_[v]; }
    int IndexToVertexOrPass(int i) const { return i == points_ ? kPassMove : IndexToVertex(i); }
    int VertexToIndexOrPass(int v) const { return v == kPassMove ? points_ : VertexToIndex(v); }
    int Step(int k) const { return dir_[k]; } // 0..3 orthogonal, 4..7 diagonal

    // ---- state
    int At(int v) const { return cell_[v]; }
    int ToMove() const { return to_move_; }
    int LastMove() const { return last_move_; }
    int LastMove2() const { return last_move2_; }
    int KoMove() const { return ko_move_; }
    int Passes() const { return passes_; }
    int Prisoners(int c) const { return prisoners_[c]; }
    int PlayedStones(int c) const { return played_[c]; }
    std::uint64_t Hash() const { return hash_; }
    std::uint64_t KoHash() const { return ko_hash_; }
    std::uint64_t SymmetryHash(int symm) const;   // board.cc:355-359
    std::uint64_t SymmetryKoHash(int symm) const; // board.cc:365-369
    std::uint64_t MoveHash(int v, int c) const;   // board.h:607-613

    // ---- chains
    int ChainHead(int v) const { return head_[v]; }
    int ChainNext(int v) const { return next_[v]; }
    int Liberties(int v) const { return libs_[head_[v]]; }
    int Stones(int v) const { return stones_[head_[v]]; }
    int EmptyNeighbours(int v) const { return (nbr_[v] >> 8) & 0xf; }

    // ---- rules
    bool IsSuicide(int v, int c) const; // board.cc:940-959
    bool IsLegal(int v, int c) const;   // board.cc:203-231 (pass / resign are legal)
    void Play(int v, int c);            // PlayMoveAssumeLegal, board.cc:1484-1508
    void SetToMove(int c);
    void SetLastMove(int a, int b) { last_move_ = a; last_move2_ = b; }
    void RemoveMarked(const int* vertices, int n); // RemoveMarkedStrings, board.cc:251-262

    // ---- local tactics
    bool IsSimpleEye(int v, int c) const;       // board.cc:901-903
    bool IsRealEye(int v, int c) const;         // board.cc:905-938
    bool IsCaptureMove(int v, int c) const;     // board.cc:870-882
    bool IsAtariMove(int v, int c) const;       // board.cc:851-868
    bool IsEscapeMove(int v, int c) const;      // board.cc:884-890
    bool IsSelfAtariMove(int v, int c) const;   // board.cc:822-849
    bool IsNeighbourColor(int v, int c) const;  // board.cc:1078-1084
    bool IsBorder(int v) const { return IsNeighbourColor(v, kWall); }
    bool IsAdjacent(int a, int b) const;
    bool IsSeki(int v) const;                   // board.cc:961-1072

    // ---- ladder reading, board.cc:484-820, 1618-1691
    bool IsLadder(int v, int* vital, int* num_vital) const;
    void LadderMap(std::uint8_t* out /*[NumPoints] LadderMark*/) const;

    // ---- area analyses (outputs are indexed by intersection index, values are Color codes)
    void ReachArea(int* out) const;                                            // board.cc:1547-1579
    void ScoreArea(int* out, int scoring, const int* territory_helper) const;  // board.cc:1581-1616
    void SafeArea(bool* out, bool mark_seki) const;                            // board.cc:1706-1718
    // Area-scoring owner map and safe area in one pass (each colour's Benson analysis runs once; the two
    // separate calls above would run it twice each).  Same results as ScoreArea(kAreaScoring) + SafeArea(false).
    void ScoreAndSafeArea(int* owner, bool* safe) const;
    int ScoreOnBoard(int color, int scoring, const int* territory_helper) const; // board.cc:1526-1545
    void PassAliveArea(bool* out, int color, bool mark_vitals, bool mark_pass_dead) const; // board.cc:1720-1901
    int ReachGroup(int start, int spread, bool* seen /*[NumVertices]*/) const;  // board.cc:264-300

    int ChainMembers(int v, int* out) const; // GetStringList, board.cc:1510-1524
    // the stones in index order (out[y * size + x] = At(vertex)): one row copy per board row
    void CopyStones(std::uint8_t* out) const {
        for (int y = 0; y < size_; ++y) std::memcpy(out + y * size_, cell_ + (y + 1) * letter_ + 1, static_cast<size_t>(size_));
    }

private:
    // cells
    std::uint8_t cell_[kMaxVertices];
    // neighbour counters: bits 0-3 black, 4-7 white, 8-11 empty orthogonal neighbours (a wall counts as black and white)
    std::uint16_t nbr_[kMaxVertices];
    // chains: circular list of stones, head, and per-head liberty / stone counts; slot kMaxVertices is the
    // "no chain" sentinel every non-stone cell points to
    std::uint16_t next_[kMaxVertices + 1];
    std::uint16_t head_[kMaxVertices + 1];
    std::uint16_t libs_[kMaxVertices + 1];
    std::uint16_t stones_[kMaxVertices + 1];

    std::uint64_t hash_, ko_hash_;
    const std::int16_t* i2v_;  // static tables: a Position stays trivially copyable
    const std::int16_t* v2i_;
    std::int16_t dir_[8];
    std::int16_t size_, letter_, points_, vertices_;
    std::int16_t to_move_, last_move_, last_move2_, ko_move_, passes_;
    std::int32_t prisoners_[2];
    std::int32_t played_[2];

    void PlaceStone(int v, int c);
    void LiftStone(int v, int c);
    void Merge(int keep, int gone);
    int RemoveChain(int v);
    void AddPrisoners(int c, int n);
    int PutAndResolve(int v, int c); // UpdateBoard; returns the new ko vertex or kNoVertex

    int ChainLiberties(int v, int* buf, int& n) const;       // FindStringLiberties: append unseen liberties
    int CaptureGainLiberties(int v, int* buf, int& n) const; // FindStringLibertiesGainingCaptures
    void LadderLibertyBounds(int v, int c, int& lo, int& hi) const;
    int PreyCandidates(int prey, int target, int* sel, int& n, bool think_ko) const;
    int HunterCandidates(int prey, int target, int* sel, int& n) const;
    int PreyTurn(Position& b, int hunter_move, int prey, int target, int& nodes) const;
    int HunterTurn(Position& b, int prey_move, int prey, int target, int& nodes) const;
    void ChainSurround(int v, int c, int* libbuf, int& nl, int* chainbuf, int& nc) const;
    bool KillableSekiEye(int v, int eye_size, const std::uint16_t* eye_next) const;

    struct Groups; // scratch for the region / chain classification of the Benson pass
    void Classify(int target, const std::uint8_t* feat, Groups& g) const;
    struct Labels; // provisional labels of the raster labelling pass behind Classify
    void LinkGroups(const Labels& labels, Groups& g) const;
    void PassAliveFromLabels(bool* out, int color, bool mark_vitals, bool mark_pass_dead, const Labels& L, std::uint8_t* spoilt,
                             const std::uint16_t* chain_head, int nchains) const;
    bool AtariEscapesAtOnce(int atari, int extend, int prey) const;
    bool RegionPassDead(int v, int c, const std::uint8_t* feat, const Groups& regions) const;
    void InnerRegions(int v, int c, const Groups& regions, bool* inner) const;
};

} // namespace sayuri_go
