#include "game_state.h"

#include <algorithm>
#include <cmath>

namespace sayuri_go {

// ---------------------------------------------------------------------------------------------
void FrameLog::truncate(int n) {
    if (n <= base_len_) {
        base_len_ = n;
        tail_.clear();
    } else {
        tail_.resize(static_cast<size_t>(n - base_len_));
    }
}

void FrameLog::clear() {
    base_.reset();
    base_len_ = 0;
    tail_.clear();
}

void FrameLog::Freeze() {
    if (tail_.empty() && (!base_ || base_len_ == static_cast<int>(base_->size()))) return;
    auto merged = std::make_shared<std::vector<Frame>>();
    merged->reserve(static_cast<size_t>(size()) + 64);
    if (base_) merged->insert(merged->end(), base_->begin(), base_->begin() + base_len_);
    merged->insert(merged->end(), tail_.begin(), tail_.end());
    base_len_ = static_cast<int>(merged->size());
    base_ = std::move(merged);
    tail_.clear();
}

// ---------------------------------------------------------------------------------------------
void GameState::PushFrame(int vtx, int color) {
    Frame f;
    board_.CopyStones(f.stones);
    f.last_move = static_cast<std::int16_t>(board_.LastMove());
    f.move_vertex = static_cast<std::int16_t>(vtx);
    f.move_color = static_cast<std::int8_t>(color);
    f.ko_hash = board_.KoHash();
    log_.push(f);
}

void GameState::Reset(int boardsize, float komi, int scoring) {
    board_.Reset(boardsize);
    SetKomi(komi);
    SetRule(scoring);
    log_.clear();
    setup_.clear();
    PushFrame(kNoVertex, kBlack);
    winner_ = kUndecided;
    handicap_ = 0;
    move_number_ = 0;
    territory_helper_.fill(kEmpty);
}

void GameState::SetBoardSize(int boardsize) { Reset(boardsize, GetKomi(), GetScoringRule()); }
void GameState::ClearBoard() { Reset(GetBoardSize(), GetKomi(), GetScoringRule()); }

bool GameState::AppendMove(int vtx, int color) {
    if (vtx == kResignMove || vtx == kPassMove) return false;
    if (move_number_ != 0) ClearBoard(); // set-up stones come before the first move
    if (!IsLegalMove(vtx, color)) return false;
    board_.Play(vtx, color);
    board_.SetToMove(kBlack);
    board_.SetLastMove(kNoVertex, kNoVertex);
    move_number_ = 0;
    log_.clear();
    PushFrame(kNoVertex, kBlack);
    setup_.emplace_back(vtx, color);
    return true;
}

bool GameState::PlayMove(int vtx, int color) {
    if (vtx == kResignMove) {
        winner_ = (color == kBlack) ? kWhiteWon : kBlackWon;
        return true;
    }
    if (!IsLegalMove(vtx, color)) return false;
    board_.Play(vtx, color);
    move_number_++;
    log_.truncate(move_number_);
    PushFrame(vtx, color);
    return true;
}

bool GameState::UndoMove() {
    if (move_number_ < 1) return false;
    // rebuild the previous board by replaying the log: chain ring order and heads come out exactly as they were
    Position b;
    b.Reset(GetBoardSize());
    for (const auto& s : setup_) {
        b.Play(s.first, s.second);
        b.SetToMove(kBlack);
        b.SetLastMove(kNoVertex, kNoVertex);
    }
    for (int i = 1; i < move_number_; ++i) b.Play(log_[i].move_vertex, log_[i].move_color);
    board_ = b;
    log_.truncate(move_number_);
    winner_ = kUndecided;
    move_number_--;
    return true;
}

void GameState::SetKomi(float komi) {
    const bool negative = komi < 0.f;
    if (negative) komi = -komi;
    const int integer_part = static_cast<int>(komi);
    const float frac = komi - static_cast<float>(integer_part);
    bool half;
    if (std::abs(frac - 0.f) < 1e-4f) half = false;
    else if (std::abs(frac - 0.5f) < 1e-4f) half = true;
    else return; // only integer and half komi are representable
    komi_half_ = half;
    komi_negative_ = negative;
    komi_integer_ = integer_part;
    const ZobristKeys& z = ZobristKeys::Get();
    komi_hash_ = z.komi[komi_integer_];
    if (komi_negative_) komi_hash_ ^= ZobristKeys::kNegativeKomi;
    if (komi_half_) komi_hash_ ^= ZobristKeys::kHalfKomi;
}

float GameState::GetKomi() const {
    float komi = static_cast<float>(komi_integer_) + static_cast<float>(komi_half_) * 0.5f;
    return komi_negative_ ? -komi : komi;
}

void GameState::SetRule(int scoring) {
    if (scoring != kAreaScoring && scoring != kTerritoryScoring) return;
    scoring_ = static_cast<std::uint8_t>(scoring);
    scoring_hash_ = ZobristKeys::Get().rule[scoring];
}

bool GameState::IsSuperko() const {
    const std::uint64_t now = GetKoHash();
    for (int i = log_.size() - 2; i >= 0; --i)
        if (log_[i].ko_hash == now) return true;
    return false;
}

bool GameState::SetFixedHandicap(int handicap) {
    const int n = GetBoardSize();
    if (handicap < 2 || handicap > 9) return false;
    if (n % 2 == 0 && handicap > 4) return false;
    if (n == 7 && handicap > 4) return false;
    if (n < 7 && handicap > 0) return false;
    const int high = n >= 13 ? 3 : 2, mid = n / 2, low = n - 1 - high;
    std::vector<int> v;
    if (handicap >= 2) {
        v.push_back(GetVertex(low, low));
        v.push_back(GetVertex(high, high));
    }
    if (handicap >= 3) v.push_back(GetVertex(high, low));
    if (handicap >= 4) v.push_back(GetVertex(low, high));
    if (handicap >= 5 && handicap % 2 == 1) v.push_back(GetVertex(mid, mid));
    if (handicap >= 6) {
        v.push_back(GetVertex(low, mid));
        v.push_back(GetVertex(high, mid));
    }
    if (handicap >= 8) {
        v.push_back(GetVertex(mid, low));
        v.push_back(GetVertex(mid, high));
    }
    PlayHandicapStones(v, true);
    return true;
}

bool GameState::PlayHandicapStones(const std::vector<int>& vertices, bool kata_like_style) {
    GameState fork = *this;
    fork.ClearBoard();
    const int n = static_cast<int>(vertices.size());
    for (int i = 0; i < n; ++i) {
        if (!fork.IsLegalMove(vertices[i], kBlack)) return false;
        // KataGo-style records: the last handicap stone is an ordinary move, the others are set-up stones
        if (i == n - 1 && kata_like_style) fork.PlayMove(vertices[i], kBlack);
        else fork.AppendMove(vertices[i], kBlack);
    }
    *this = fork;
    SetHandicap(n);
    SetToMove(kWhite);
    return true;
}

float GameState::FinalScoreWith(int color, const int* territory_helper) const {
    const float black = static_cast<float>(board_.ScoreOnBoard(kBlack, scoring_, territory_helper)) - GetKomiWithPenalty();
    return color == kBlack ? black : -black;
}

std::vector<bool> GameState::GetStrictSafeArea() const {
    bool buf[kMaxPoints];
    SafeAreaCached(buf);
    return std::vector<bool>(buf, buf + GetNumIntersections());
}

// The area cache below is written from const methods (mutable members): a GameState may be READ by one thread at a
// time -- every game, fork and search in this engine owns its states, nothing shares one across threads.  The key mixes
// the board size into the position hash, so a state that is Reset() to another size cannot meet a stale entry.
static inline std::uint64_t AreaKey(std::uint64_t ko_hash, int board_size) {
    return ko_hash ^ (static_cast<std::uint64_t>(board_size) * 0x9e3779b97f4a7c15ULL);
}

void GameState::SafeAreaCached(bool* safe) const {
    const int n = GetNumIntersections();
    const std::uint64_t key = AreaKey(board_.KoHash(), GetBoardSize());
    if (area_key_ == key && (area_have_ & 1)) {
        for (int i = 0; i < n; ++i) safe[i] = (area_safe_[i >> 6] >> (i & 63)) & 1;
        return;
    }
    board_.SafeArea(safe, false);
    if (area_key_ != key) area_have_ = 0;
    area_key_ = key;
    for (auto& w : area_safe_) w = 0;
    for (int i = 0; i < n; ++i)
        if (safe[i]) area_safe_[i >> 6] |= std::uint64_t{1} << (i & 63);
    area_have_ |= 1;
}

void GameState::ScoreAndSafeAreaCached(int* owner, bool* safe) const {
    const int n = GetNumIntersections();
    const std::uint64_t key = AreaKey(board_.KoHash(), GetBoardSize());
    if (area_key_ == key && (area_have_ & 3) == 3) {
        for (int i = 0; i < n; ++i) {
            safe[i] = (area_safe_[i >> 6] >> (i & 63)) & 1;
            owner[i] = ((area_black_[i >> 6] >> (i & 63)) & 1) ? kBlack : ((area_white_[i >> 6] >> (i & 63)) & 1) ? kWhite : kEmpty;
        }
        return;
    }
    board_.ScoreAndSafeArea(owner, safe);
    area_key_ = key;
    for (int k = 0; k < 6; ++k) area_safe_[k] = area_black_[k] = area_white_[k] = 0;
    for (int i = 0; i < n; ++i) {
        const std::uint64_t bit = std::uint64_t{1} << (i & 63);
        if (safe[i]) area_safe_[i >> 6] |= bit;
        if (owner[i] == kBlack) area_black_[i >> 6] |= bit;
        else if (owner[i] == kWhite) area_white_[i >> 6] |= bit;
    }
    area_have_ = 3;
}

std::vector<int> GameState::GetOwnership() const {
    std::vector<int> res(static_cast<size_t>(GetNumIntersections()), kWall);
    board_.ScoreArea(res.data(), scoring_, territory_helper_.data());
    return res;
}

std::vector<int> GameState::GetRawOwnership() const {
    std::vector<int> res(static_cast<size_t>(GetNumIntersections()), kWall);
    board_.ReachArea(res.data());
    return res;
}

float GameState::GetPenalty(int scoring) const {
    float penalty = 0.f;
    if (scoring == kTerritoryScoring) {
        penalty += static_cast<float>(board_.PlayedStones(kBlack));
        penalty -= static_cast<float>(board_.PlayedStones(kWhite));
    }
    if (scoring == kAreaScoring) penalty += static_cast<float>(handicap_);
    return penalty;
}

float GameState::GetPenaltyOffset(int new_scoring, int old_scoring) const {
    if (new_scoring != old_scoring) return GetPenalty(old_scoring) - GetPenalty(new_scoring);
    return GetPenalty(new_scoring);
}

float GameState::GetWave() const {
    if (scoring_ == kTerritoryScoring) return 0.f;
    float komi = GetKomiWithPenalty();
    if (GetToMove() == kWhite) komi = 0.f - komi;
    // distance of the komi (from the mover's side) above the nearest drawable komi, as a triangle wave
    float floor_komi;
    if (GetNumIntersections() % 2 == 0) floor_komi = std::floor(komi / 2.0f) * 2.0f;
    else floor_komi = std::floor((komi - 1.0f) / 2.0f) * 2.0f + 1.0f;
    float delta = komi - floor_komi;
    delta = std::max(delta, 0.f);
    delta = std::min(delta, 2.f);
    if (delta < 0.5f) return delta;
    if (delta < 1.5f) return 1.f - delta;
    return delta - 2.f;
}

int GameState::TextToVertex(const std::string& text) const {
    if (text.size() < 2) return kNoVertex;
    if (text == "PASS" || text == "pass") return kPassMove;
    if (text == "RESIGN" || text == "resign") return kResignMove;
    int x = -1, y = -1;
    const char c = text[0];
    if (c >= 'a' && c <= 'z') x = c - 'a' - (c >= 'i' ? 1 : 0);
    else if (c >= 'A' && c <= 'Z') x = c - 'A' - (c >= 'I' ? 1 : 0);
    std::string digits;
    for (size_t i = 1; i < text.size(); ++i) {
        if (text[i] >= '0' && text[i] <= '9') digits += text[i];
        else {
            digits.clear();
            break;
        }
    }
    if (!digits.empty()) y = std::stoi(digits) - 1;
    if (x == -1 || y == -1) return kNoVertex;
    return board_.Vertex(x, y);
}

std::string GameState::VertexToText(int vtx) const {
    if (vtx == kPassMove) return "pass";
    if (vtx == kResignMove) return "resign";
    const int x = GetX(vtx), y = GetY(vtx);
    std::string out;
    out += static_cast<char>(x + ('A' + x >= 'I' ? 1 : 0) + 'A');
    out += std::to_string(y + 1);
    return out;
}

std::string GameState::VertexToSgf(int vtx) const {
    if (vtx == kPassMove || vtx == kResignMove) return std::string{};
    const int x = GetX(vtx), y = GetBoardSize() - GetY(vtx) - 1;
    std::string out;
    out += static_cast<char>(x >= 26 ? x - 26 + 'A' : x + 'a');
    out += static_cast<char>(y >= 26 ? y - 26 + 'A' : y + 'a');
    return out;
}

std::vector<int> GameState::GetAppendMoves(int color) const {
    std::vector<int> out;
    for (const auto& s : setup_)
        if (s.second == color) out.push_back(s.first);
    return out;
}

} // namespace sayuri_go
