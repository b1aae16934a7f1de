#include "position.h"

#include <array>

#include <algorithm>
#include <memory>
#include <vector>

namespace sayuri_go {

namespace {
constexpr int kNoChain = kMaxVertices; // sentinel slot of next_/head_/libs_/stones_
constexpr int kBlackShift = 0, kWhiteShift = 4, kEmptyShift = 8;
constexpr int kLadderNodeLimit = 2000; // reference types.h:73
enum LadderVerdict { kHunterWins = 0, kPreyWins = 1, kLadderOpen = 2 };

inline bool Contains(const int* buf, int n, int v) {
    for (int i = 0; i < n; ++i)
        if (buf[i] == v) return true;
    return false;
}
} // namespace

// ---------------------------------------------------------------------------------------------
void Position::Reset(int board_size) {
    board_size = std::min(std::max(board_size, kMinBoard), kMaxBoard);
    size_ = static_cast<std::int16_t>(board_size);
    letter_ = static_cast<std::int16_t>(board_size + 2);
    points_ = static_cast<std::int16_t>(board_size * board_size);
    vertices_ = static_cast<std::int16_t>(letter_ * letter_);
    i2v_ = IndexTables::Get().i2v[board_size];
    v2i_ = IndexTables::Get().v2i[board_size];

    for (int v = 0; v < kMaxVertices; ++v) {
        cell_[v] = kWall;
        nbr_[v] = 0;
    }
    for (int y = 0; y < size_; ++y) {
        for (int x = 0; x < size_; ++x) {
            const int v = Vertex(x, y);
            cell_[v] = kEmpty;
            // per axis: an edge has one wall (counted as a black and a white neighbour) and one empty
            // neighbour, an interior cell has two empty neighbours
            const bool xe = (x == 0 || x == size_ - 1), ye = (y == 0 || y == size_ - 1);
            const int wall = (1 << kBlackShift) | (1 << kWhiteShift) | (1 << kEmptyShift);
            nbr_[v] = static_cast<std::uint16_t>((xe ? wall : (2 << kEmptyShift)) + (ye ? wall : (2 << kEmptyShift)));
        }
    }
    for (int v = 0; v <= kMaxVertices; ++v) {
        next_[v] = head_[v] = kNoChain;
        libs_[v] = stones_[v] = 0;
    }
    libs_[kNoChain] = 16384;

    prisoners_[0] = prisoners_[1] = 0;
    played_[0] = played_[1] = 0;
    ko_move_ = last_move_ = last_move2_ = kNoVertex;
    to_move_ = kBlack;
    passes_ = 0;
    const int l = letter_;
    const int d[8] = {-l, -1, +1, +l, -l - 1, -l + 1, +l - 1, +l + 1};
    for (int k = 0; k < 8; ++k) dir_[k] = static_cast<std::int16_t>(d[k]);

    const ZobristKeys& z = ZobristKeys::Get();
    ko_hash_ = ZobristKeys::kEmptyBoard;
    for (int v = 0; v < vertices_; ++v)
        if (cell_[v] != kWall) ko_hash_ ^= z.state[cell_[v]][v];
    hash_ = ko_hash_ ^ ZobristKeys::kBlackToMove ^ z.prisoner[kBlack][0] ^ z.prisoner[kWhite][0] ^ z.pass[0] ^
            z.ko[kNoVertex];
}

std::uint64_t Position::SymmetryKoHash(int symm) const {
    // = kEmptyBoard ^ XOR over the board's cells of state[cell][image of the cell].  A symmetry permutes the board's cells, so
    // the empty cells' share is the whole board's XOR of state[kEmpty][.] (one constant per board size) with the stones' cells
    // taken back out: the loop below only does work per STONE -- these hashes are asked for during the first board-size moves
    // of a game (Network::ProbeCache, seven per leaf), when there are a handful.
    static const auto empty_all = [] {
        std::array<std::uint64_t, kMaxBoard + 1> e{};
        const ZobristKeys& z = ZobristKeys::Get();
        for (int bs = 1; bs <= kMaxBoard; ++bs)
            for (int y = 0; y < bs; ++y)
                for (int x = 0; x < bs; ++x) e[static_cast<size_t>(bs)] ^= z.state[kEmpty][(y + 1) * (bs + 2) + x + 1];
        return e;
    }();
    const ZobristKeys& z = ZobristKeys::Get();
    const SymmetryTables& t = SymmetryTables::Get();
    std::uint64_t h = ZobristKeys::kEmptyBoard ^ empty_all[static_cast<size_t>(size_)];
    const int w = size_ + 2;
    for (int y = 1; y <= size_; ++y) {
        const std::uint8_t* row = &cell_[y * w];
        for (int x = 1; x <= size_; ++x) {
            const int c = row[x];
            if (c == kEmpty) continue;
            const int sv = t.Vertex(size_, symm, y * w + x);
            h ^= z.state[kEmpty][sv] ^ z.state[c][sv];
        }
    }
    return h;
}

std::uint64_t Position::SymmetryHash(int symm) const {
    const ZobristKeys& z = ZobristKeys::Get();
    std::uint64_t h = SymmetryKoHash(symm);
    if (to_move_ == kBlack) h ^= ZobristKeys::kBlackToMove;
    h ^= z.prisoner[kBlack][prisoners_[kBlack]] ^ z.prisoner[kWhite][prisoners_[kWhite]] ^ z.pass[passes_];
    h ^= z.ko[SymmetryTables::Get().Vertex(size_, symm, ko_move_)];
    return h;
}

std::uint64_t Position::MoveHash(int v, int c) const {
    std::uint64_t h = ZobristKeys::Get().state[c][v];
    if (c == to_move_) h ^= ZobristKeys::kBlackToMove;
    return h;
}

// ---------------------------------------------------------------------------------------------
void Position::SetToMove(int c) {
    if (c != to_move_) hash_ ^= ZobristKeys::kBlackToMove;
    to_move_ = static_cast<std::int16_t>(c);
}

void Position::PlaceStone(int v, int c) {
    const ZobristKeys& z = ZobristKeys::Get();
    cell_[v] = static_cast<std::uint8_t>(c);
    const std::uint64_t dz = z.state[kEmpty][v] ^ z.state[c][v];
    hash_ ^= dz;
    ko_hash_ ^= dz;
    int seen[4], ns = 0;
    for (int k = 0; k < 4; ++k) {
        const int a = v + dir_[k];
        nbr_[a] = static_cast<std::uint16_t>(nbr_[a] + (1 << (4 * c)) - (1 << kEmptyShift));
        const int h = head_[a];
        if (!Contains(seen, ns, h)) { // every distinct neighbouring chain loses this liberty once
            libs_[h]--;
            seen[ns++] = h;
        }
    }
}

void Position::LiftStone(int v, int c) {
    const ZobristKeys& z = ZobristKeys::Get();
    cell_[v] = kEmpty;
    const std::uint64_t dz = z.state[kEmpty][v] ^ z.state[c][v];
    hash_ ^= dz;
    ko_hash_ ^= dz;
    int seen[4], ns = 0;
    for (int k = 0; k < 4; ++k) {
        const int a = v + dir_[k];
        nbr_[a] = static_cast<std::uint16_t>(nbr_[a] + (1 << kEmptyShift) - (1 << (4 * c)));
        const int h = head_[a];
        if (!Contains(seen, ns, h)) {
            libs_[h]++;
            seen[ns++] = h;
        }
    }
}

void Position::Merge(int keep, int gone) {
    // liberties of `gone` that `keep` does not touch yet, then splice the two rings
    stones_[keep] = static_cast<std::uint16_t>(stones_[keep] + stones_[gone]);
    int p = gone;
    do {
        for (int k = 0; k < 4; ++k) {
            const int a = p + dir_[k];
            if (cell_[a] != kEmpty) continue;
            bool shared = false;
            for (int kk = 0; kk < 4; ++kk) {
                if (head_[a + dir_[kk]] == keep) {
                    shared = true;
                    break;
                }
            }
            if (!shared) libs_[keep]++;
        }
        head_[p] = static_cast<std::uint16_t>(keep);
        p = next_[p];
    } while (p != gone);
    std::swap(next_[gone], next_[keep]);
}

int Position::RemoveChain(int v) {
    const int c = cell_[v];
    int p = v, n = 0;
    do {
        LiftStone(p, c);
        head_[p] = kNoChain;
        ++n;
        p = next_[p];
    } while (p != v);
    return n;
}

void Position::AddPrisoners(int c, int n) {
    const ZobristKeys& z = ZobristKeys::Get();
    hash_ ^= z.prisoner[c][prisoners_[c]];
    prisoners_[c] += n;
    hash_ ^= z.prisoner[c][prisoners_[c]];
}

int Position::PutAndResolve(int v, int c) {
    PlaceStone(v, c);
    next_[v] = head_[v] = static_cast<std::uint16_t>(v);
    libs_[v] = static_cast<std::uint16_t>(EmptyNeighbours(v));
    stones_[v] = 1;

    const bool in_opp_eye = IsSimpleEye(v, Opp(c));
    int captured = 0, captured_at = kNoVertex;
    for (int k = 0; k < 4; ++k) {
        const int a = v + dir_[k];
        const int s = cell_[a];
        if (s == Opp(c)) {
            if (libs_[head_[a]] == 0) {
                captured += RemoveChain(a);
                captured_at = a;
            }
        } else if (s == c) {
            const int mine = head_[v], other = head_[a];
            if (mine != other) {
                if (stones_[mine] >= stones_[other]) Merge(mine, other);
                else Merge(other, mine);
            }
        }
    }
    if (libs_[head_[v]] == 0) AddPrisoners(Opp(c), RemoveChain(v)); // suicide (never legal, kept for parity)
    if (captured) AddPrisoners(c, captured);
    return (captured == 1 && in_opp_eye) ? captured_at : kNoVertex;
}

void Position::Play(int v, int c) {
    const ZobristKeys& z = ZobristKeys::Get();
    SetToMove(c);
    const int old_ko = ko_move_;
    if (v == kPassMove) {
        const int np = std::min(passes_ + 1, 4);
        hash_ ^= z.pass[passes_] ^ z.pass[np];
        passes_ = static_cast<std::int16_t>(np);
        ko_move_ = kNoVertex;
    } else {
        if (passes_ != 0) {
            hash_ ^= z.pass[passes_] ^ z.pass[0];
            passes_ = 0;
        }
        ko_move_ = static_cast<std::int16_t>(PutAndResolve(v, c));
        played_[c] += 1;
    }
    if (ko_move_ != old_ko) hash_ ^= z.ko[old_ko] ^ z.ko[ko_move_];
    last_move2_ = last_move_;
    last_move_ = static_cast<std::int16_t>(v);
    to_move_ ^= 1;
    hash_ ^= ZobristKeys::kBlackToMove;
}

void Position::RemoveMarked(const int* vertices, int n) {
    int removed[2] = {0, 0};
    for (int i = 0; i < n; ++i) {
        const int c = cell_[vertices[i]];
        if (c == kBlack || c == kWhite) removed[c] += RemoveChain(vertices[i]);
    }
    AddPrisoners(kBlack, removed[kWhite]);
    AddPrisoners(kWhite, removed[kBlack]);
}

// ---------------------------------------------------------------------------------------------
bool Position::IsSuicide(int v, int c) const {
    if (EmptyNeighbours(v)) return false;
    for (int k = 0; k < 4; ++k) {
        const int a = v + dir_[k];
        const int l = libs_[head_[a]];
        if (cell_[a] == c && l > 1) return false;       // joins a chain that keeps a liberty
        if (cell_[a] == Opp(c) && l <= 1) return false; // captures
    }
    return true;
}

bool Position::IsLegal(int v, int c) const {
    if (v == kPassMove || v == kResignMove) return true;
    if (cell_[v] != kEmpty) return false;
    if (IsSuicide(v, c)) return false;
    return v != ko_move_;
}

bool Position::IsSimpleEye(int v, int c) const { return (nbr_[v] & (4 << (4 * c))) != 0; }

bool Position::IsRealEye(int v, int c) const {
    if (cell_[v] != kEmpty || !IsSimpleEye(v, c)) return false;
    int cnt[4] = {0, 0, 0, 0};
    for (int k = 4; k < 8; ++k) cnt[cell_[v + dir_[k]]]++;
    return cnt[kWall] == 0 ? cnt[Opp(c)] <= 1 : cnt[Opp(c)] == 0;
}

bool Position::IsCaptureMove(int v, int c) const {
    for (int k = 0; k < 4; ++k) {
        const int a = v + dir_[k];
        if (cell_[a] == Opp(c) && libs_[head_[a]] == 1) return true;
    }
    return false;
}

bool Position::IsAtariMove(int v, int c) const {
    if (IsSuicide(v, c)) return false;
    for (int k = 0; k < 4; ++k) {
        const int a = v + dir_[k];
        if (cell_[a] == Opp(c) && libs_[head_[a]] == 2) return true;
    }
    return false;
}

bool Position::IsEscapeMove(int v, int c) const { return !IsSuicide(v, c) && IsCaptureMove(v, Opp(c)); }

bool Position::IsSelfAtariMove(int v, int c) const {
    int own = EmptyNeighbours(v);
    int buf[kMaxPoints + 1], n = 0;
    buf[n++] = v;
    for (int k = 0; k < 4; ++k) {
        const int a = v + dir_[k];
        if (cell_[a] == c) ChainLiberties(a, buf, n);
        else if (cell_[a] == Opp(c) && libs_[head_[a]] <= 1) own += 1;
    }
    return (n - 1) + own == 1;
}

bool Position::IsNeighbourColor(int v, int c) const {
    for (int k = 0; k < 4; ++k)
        if (cell_[v + dir_[k]] == c) return true;
    return false;
}

bool Position::IsAdjacent(int a, int b) const {
    for (int k = 0; k < 4; ++k)
        if (a + dir_[k] == b) return true;
    return false;
}

int Position::ChainMembers(int v, int* out) const {
    const int start = head_[v];
    int p = start, n = 0;
    do {
        out[n++] = p;
        p = next_[p];
    } while (p != start);
    return n;
}

// ---------------------------------------------------------------------------------------------
// Ladder reading.
int Position::ChainLiberties(int v, int* buf, int& n) const {
    // The chain's liberty count is kept incrementally, so the walk can stop as soon as every liberty has been met (same
    // entries in the same order as the full walk): into an empty list that is `total` new entries; a chain in atari is done
    // at its first empty neighbour whatever the list holds.  Ladders walk a growing chain at every step of the chase.
    const int total = libs_[head_[v]];
    const bool fresh = n == 0;
    int found = 0, p = v;
    do {
        // (a stone whose neighbour counts show no empty point -- the inside of a chain -- is passed without looking around it;
        // on the edge the wall counts as one, those stones are looked at as before)
        if (((nbr_[p] >> kEmptyShift) & 0xf) == 0) {
            p = next_[p];
            continue;
        }
        for (int k = 0; k < 4; ++k) {
            const int a = p + dir_[k];
            if (cell_[a] != kEmpty) continue;
            if (!Contains(buf, n, a)) {
                buf[n++] = a;
                ++found;
            }
            if (total == 1 || (fresh && found == total)) return found;
        }
        p = next_[p];
    } while (p != v);
    return found;
}

int Position::CaptureGainLiberties(int v, int* buf, int& n) const {
    const int opp = Opp(cell_[v]);
    const int opp_shift = opp == kBlack ? kBlackShift : kWhiteShift;
    int found = 0, p = v;
    do {
        if (((nbr_[p] >> opp_shift) & 0xf) != 0) {  // (stones without an enemy neighbour are passed)
            for (int k = 0; k < 4; ++k) {
                const int a = p + dir_[k];
                if (cell_[a] == opp && libs_[head_[a]] == 1) found += ChainLiberties(a, buf, n);
            }
        }
        p = next_[p];
    } while (p != v);
    return found;
}

void Position::LadderLibertyBounds(int v, int c, int& lo, int& hi) const {
    const int own = EmptyNeighbours(v);
    int captures = 0, capture_gain = 0, connect_sum = 0, connect_max = own;
    for (int k = 0; k < 4; ++k) {
        const int a = v + dir_[k];
        if (cell_[a] == c) {
            const int l = libs_[head_[a]] - 1;
            connect_sum += l;
            connect_max = std::max(connect_max, l);
        } else if (cell_[a] == Opp(c)) {
            if (libs_[head_[a]] == 1) {
                ++captures;
                capture_gain += stones_[head_[a]];
            }
        }
    }
    lo = captures + connect_max;
    hi = own + capture_gain + connect_sum;
}

int Position::PreyCandidates(int prey, int target, int* sel, int& n, bool think_ko) const {
    n = 0;
    // two liberties: escaped. A pending simple ko after the hunter's move also counts as escaped, so that
    // ko-dependent ladders never read as working (and the recursion cannot loop).
    if (libs_[head_[target]] >= 2 || (ko_move_ != kNoVertex && think_ko)) return kPreyWins;
    ChainLiberties(target, sel, n);
    const int extend = sel[0];
    CaptureGainLiberties(target, sel, n);
    int m = 0;
    for (int i = 0; i < n; ++i)
        if (IsLegal(sel[i], prey)) sel[m++] = sel[i];
    n = m;
    if (n == 0) return kHunterWins;
    if (Contains(sel, n, extend)) {
        int lo, hi;
        LadderLibertyBounds(extend, prey, lo, hi);
        if (lo >= 3) return kPreyWins;
        if (n == 1 && hi == 1) return kHunterWins;
    }
    return kLadderOpen;
}

int Position::HunterCandidates(int prey, int target, int* sel, int& n) const {
    n = 0;
    const int l = libs_[head_[target]];
    if (l >= 3) return kPreyWins;
    if (l <= 1) return kHunterWins;
    int lib[kMaxPoints], nl = 0;
    ChainLiberties(target, lib, nl);
    const int a = lib[0], b = lib[1], hunter = Opp(prey);
    if (!IsAdjacent(a, b)) {
        const int la = EmptyNeighbours(a), lb = EmptyNeighbours(b);
        if (la >= 3 && lb >= 3) return kPreyWins;
        if (la >= 3) {
            if (IsLegal(a, hunter)) sel[n++] = a;
        } else if (lb >= 3) {
            if (IsLegal(b, hunter)) sel[n++] = b;
        } else {
            if (IsLegal(a, hunter)) sel[n++] = a;
            if (IsLegal(b, hunter)) sel[n++] = b;
        }
    } else {
        sel[n++] = a;
        sel[n++] = b;
    }
    return n == 0 ? kPreyWins : kLadderOpen;
}

namespace {
// A board for one branch of the ladder reading.  The boards of a thread are recycled: a reading never leaves its thread (no
// network call inside), and the allocator was visible in the profile of a self-play rank (a 5 KB malloc / free per fork).
class ForkBoard {
public:
    explicit ForkBoard(const Position& src) {
        auto& pool = Pool();
        if (pool.empty()) {
            p_ = std::make_unique<Position>(src);
        } else {
            p_ = std::move(pool.back());
            pool.pop_back();
            *p_ = src;
        }
    }
    ~ForkBoard() { Pool().push_back(std::move(p_)); }
    ForkBoard(const ForkBoard&) = delete;
    ForkBoard& operator=(const ForkBoard&) = delete;
    Position& operator*() { return *p_; }

private:
    static std::vector<std::unique_ptr<Position>>& Pool() {
        static thread_local std::vector<std::unique_ptr<Position>> pool;
        return pool;
    }
    std::unique_ptr<Position> p_;
};
}  // namespace

int Position::PreyTurn(Position& b, int hunter_move, int prey, int target, int& nodes) const {
    if (++nodes >= kLadderNodeLimit) return kPreyWins;
    if (hunter_move != kNoVertex) b.Play(hunter_move, Opp(prey));
    int sel[kMaxPoints], n;
    int verdict = b.PreyCandidates(prey, target, sel, n, hunter_move != kNoVertex);
    if (verdict != kLadderOpen) return verdict;
    for (int i = 0; i < n; ++i) {
        if (i == n - 1) {
            // the last (or only) candidate continues in place: nobody looks at this board again -- the caller handed over
            // either a fork of its own or, by this same rule, the board of ITS last candidate
            verdict = HunterTurn(b, sel[i], prey, target, nodes);
        } else {
            ForkBoard fork(b);
            verdict = HunterTurn(*fork, sel[i], prey, target, nodes);
        }
        if (verdict == kPreyWins) break;
    }
    return verdict;
}

int Position::HunterTurn(Position& b, int prey_move, int prey, int target, int& nodes) const {
    if (++nodes >= kLadderNodeLimit) return kPreyWins;
    if (prey_move != kNoVertex) b.Play(prey_move, prey);
    int sel[4], n;
    int verdict = b.HunterCandidates(prey, target, sel, n);
    if (verdict != kLadderOpen) return verdict;
    for (int i = 0; i < n; ++i) {
        if (i == n - 1) {
            verdict = PreyTurn(b, sel[i], prey, target, nodes);  // the last candidate in place (see PreyTurn)
        } else {
            ForkBoard fork(b);
            verdict = PreyTurn(*fork, sel[i], prey, target, nodes);
        }
        if (verdict == kHunterWins) break;
    }
    return verdict;
}

bool Position::IsLadder(int v, int* vital, int* num_vital) const {
    *num_vital = 0;
    if (v == kPassMove) return false;
    const int prey = cell_[v];
    if (prey == kEmpty || prey == kWall) return false;
    // only chains with one or two liberties are read; the chain's liberty count is kept incrementally, so the others
    // leave before their liberties are enumerated (a walk over the whole chain with a quadratic de-duplication: 9 % of
    // all host time went there, most of it for large safe chains)
    if (libs_[head_[v]] > 2) return false;
    int lib[kMaxPoints], nl = 0;
    ChainLiberties(v, lib, nl);
    int nodes = 0;
    if (nl == 1) {
        // PreyTurn with no hunter move in front: its first look is at the board as it stands, so the working copy (5 KB) is
        // only made when the prey has a move to try
        int sel[kMaxPoints], n, verdict = kPreyWins;
        if (++nodes < kLadderNodeLimit) {
            verdict = PreyCandidates(prey, v, sel, n, false);
            if (verdict == kLadderOpen) {
                for (int i = 0; i < n; ++i) {
                    Position work = *this;
                    verdict = HunterTurn(work, sel[i], prey, v, nodes);
                    if (verdict == kPreyWins) break;
                }
            }
        }
        if (verdict == kHunterWins) vital[(*num_vital)++] = lib[0];
    } else if (nl == 2) {
        for (int i = 0; i < 2; ++i) {
            if (!IsLegal(lib[i], Opp(prey))) continue;
            // the hunter ataris first
            if (AtariEscapesAtOnce(lib[i], lib[1 - i], prey)) {
                ++nodes;  // the node PreyTurn would have counted before it found the same
                continue;
            }
            Position work = *this;
            if (PreyTurn(work, lib[i], prey, v, nodes) == kHunterWins) vital[(*num_vital)++] = lib[i];
        }
    }
    return *num_vital > 0;
}

bool Position::AtariEscapesAtOnce(int atari, int extend, int prey) const {
    // The hunter ataris a two-liberty chain at `atari`; the prey's answer is to extend at `extend`.  PreyCandidates calls the
    // ladder off when that extension is legal and reaches three liberties by itself or by the chain it connects to
    // (LadderLibertyBounds' lower bound without its capture term, which only adds).  If the hunter's stone captures nothing,
    // the board after it is this board plus one stone: the bound can be read here, without the copy and the move.  Most
    // two-liberty chains of a position are not in a ladder and leave through this door.
    for (int k = 0; k < 4; ++k) {
        const int a = atari + dir_[k];
        if (cell_[a] == prey && libs_[head_[a]] == 1) return false;  // a capture: the board changes by more than the stone
    }
    int best = EmptyNeighbours(extend) - (IsAdjacent(atari, extend) ? 1 : 0);
    for (int k = 0; k < 4; ++k) {
        const int a = extend + dir_[k];
        if (cell_[a] != prey) continue;
        const int h = head_[a];
        int l = libs_[h];
        for (int kk = 0; kk < 4; ++kk) {
            const int t = atari + dir_[kk];
            if (cell_[t] == prey && head_[t] == h) {
                l -= 1;  // the hunter's stone takes this liberty
                break;
            }
        }
        best = std::max(best, l - 1);
    }
    return best >= 3;
}

void Position::LadderMap(std::uint8_t* out) const {
    std::memset(out, kLadderNone, points_);
    std::uint8_t verdict[kMaxVertices + 1]; // per chain head: 0 unknown, 1 ladder, 2 not a ladder
    std::memset(verdict, 0, sizeof(verdict));
    for (int idx = 0; idx < points_; ++idx) {
        const int v = IndexToVertex(idx);
        if (cell_[v] == kEmpty) continue;
        const int h = head_[v];
        int vital[2], nv = 0;
        bool first = false;
        if (verdict[h] == 0) {
            if (IsLadder(v, vital, &nv)) {
                verdict[h] = 1;
                first = true;
            } else {
                verdict[h] = 2;
            }
        }
        if (verdict[h] != 1) continue;
        const int l = libs_[h];
        out[idx] = (l == 1) ? kLadderDeath : kLadderEscapable;
        if (first)
            for (int i = 0; i < nv; ++i) out[VertexToIndex(vital[i])] = (l == 1) ? kLadderTake : kLadderAtari;
    }
}

// ---------------------------------------------------------------------------------------------
// Areas.
int Position::ReachGroup(int start, int spread, bool* seen) const {
    int queue[kMaxVertices], qh = 0, qt = 0, n = 1;
    seen[start] = true;
    queue[qt++] = start;
    while (qh < qt) {
        const int v = queue[qh++];
        for (int k = 0; k < 4; ++k) {
            const int a = v + dir_[k];
            if (!seen[a] && cell_[a] == spread) {
                seen[a] = true;
                queue[qt++] = a;
                ++n;
            }
        }
    }
    return n;
}

// Provisional labels of a raster labelling pass: lab[v] (from 1; 0 = not a cell of the labelled kind, or off the board), and a
// union-find forest over the labels in which a label's parent is always a smaller label.
struct Position::Labels {
    std::uint16_t lab[kMaxVertices];
    std::uint16_t parent[kMaxPoints + 2];
    int count;
    // the label of the cell at v, given the labels above and to the left of it
    int Join(int v, int letter) {
        int a = lab[v - letter], b = lab[v - 1];
        int mine;
        if (a == 0 && b == 0) {
            mine = ++count;
            parent[mine] = static_cast<std::uint16_t>(mine);
        } else if (a == 0 || b == 0 || a == b) {
            mine = a ? a : b;
        } else {
            while (parent[a] != a) a = parent[a];
            while (parent[b] != b) b = parent[b];
            mine = a < b ? a : b;
            parent[a < b ? b : a] = static_cast<std::uint16_t>(mine);
        }
        lab[v] = static_cast<std::uint16_t>(mine);
        return mine;
    }
};

void Position::ReachArea(int* out) const {
    // Tromp-Taylor: a point belongs to a colour when it is that colour or reaches only that colour through empties --
    // i.e. an empty region belongs to the one colour it borders (reference board.cc:1547-1579 does two breadth-first
    // searches from the stones; labelling the empty regions gives the same map).  The regions are labelled in one raster
    // pass (a cell joins the label of its upper / left neighbour, two labels that meet are united, the smaller one
    // staying the root) with the colours each label touches OR-ed along: no stack, and branches that follow the board's
    // rows instead of a flood fill's frontier.  This runs once per evaluated position.
    std::uint16_t lab[kMaxVertices];
    std::memset(lab, 0, sizeof(lab));
    std::uint16_t parent[kMaxPoints + 2];
    std::uint8_t touch[kMaxPoints + 2];
    int labels = 0;
    const int l = letter_;
    auto find = [&parent](int x) {
        while (parent[x] != x) {
            parent[x] = parent[parent[x]];
            x = parent[x];
        }
        return x;
    };
    for (int i = 0; i < points_; ++i) {
        const int v = IndexToVertex(i);
        if (cell_[v] != kEmpty) continue;
        int a = lab[v - l], b = lab[v - 1];
        int mine;
        if (a == 0 && b == 0) {
            mine = ++labels;
            parent[mine] = static_cast<std::uint16_t>(mine);
            touch[mine] = 0;
        } else if (a == 0 || b == 0 || a == b) {
            mine = a ? a : b;
        } else {
            a = find(a);
            b = find(b);
            mine = a < b ? a : b;
            parent[a < b ? b : a] = static_cast<std::uint16_t>(mine);
        }
        lab[v] = static_cast<std::uint16_t>(mine);
        unsigned border = 0;
        for (int k = 0; k < 4; ++k) {
            const int ca = cell_[v + dir_[k]];
            if (ca == kBlack || ca == kWhite) border |= 1u << ca;
        }
        touch[mine] |= static_cast<std::uint8_t>(border);
    }
    // a label's parent is always a smaller label: fold the masks towards the roots from the top, then hand every label its root's
    for (int x = labels; x >= 1; --x)
        if (parent[x] != x) touch[parent[x]] |= touch[x];
    for (int x = 1; x <= labels; ++x)
        if (parent[x] != x) touch[x] = touch[parent[x]];
    for (int i = 0; i < points_; ++i) {
        const int v = IndexToVertex(i);
        const int c = cell_[v];
        if (c == kBlack || c == kWhite) {
            out[i] = c;
        } else {
            const unsigned border = touch[lab[v]];
            out[i] = border == (1u << kBlack) ? kBlack : border == (1u << kWhite) ? kWhite : kEmpty;
        }
    }
}

void Position::ScoreArea(int* out, int scoring, const int* helper) const {
    if (scoring == kTerritoryScoring) {
        // stones the helper assigns to the other side are dead: lift them, then score by area
        Position fork = *this;
        int dead[kMaxPoints], nd = 0;
        for (int i = 0; i < points_; ++i) {
            const int v = IndexToVertex(i);
            if ((helper[i] == kBlack && cell_[v] == kWhite) || (helper[i] == kWhite && cell_[v] == kBlack)) dead[nd++] = v;
        }
        fork.RemoveMarked(dead, nd);
        fork.ScoreArea(out, kAreaScoring, helper);
        return;
    }
    ReachArea(out);
    bool alive[kMaxPoints];
    for (int c = 0; c < 2; ++c) {
        std::memset(alive, 0, sizeof(alive));
        PassAliveArea(alive, c, true, true);
        for (int i = 0; i < points_; ++i)
            if (alive[i]) out[i] = c;
    }
}

void Position::ScoreAndSafeArea(int* owner, bool* safe) const {
    // ReachArea + PassAliveArea for both colours, with ONE pass over the board for their three labellings (the empty regions,
    // the regions of non-black cells, the regions of non-white cells): once per evaluated position, and each labelling pass of its
    // own was a loop of 361 hard-to-predict branches.  Same labels, same flags, same results as the three separate functions.
    Labels E, NB, NW;  // empty regions | regions without black stones | regions without white stones
    std::memset(E.lab, 0, sizeof(E.lab));
    std::memset(NB.lab, 0, sizeof(NB.lab));
    std::memset(NW.lab, 0, sizeof(NW.lab));
    E.count = NB.count = NW.count = 0;
    std::uint8_t touch[kMaxPoints + 2], spoilt_b[kMaxPoints + 2], spoilt_w[kMaxPoints + 2];
    std::uint16_t heads_b[kMaxPoints], heads_w[kMaxPoints];
    int nb = 0, nw = 0;
    const int l = letter_;
    for (int i = 0; i < points_; ++i) {
        const int v = IndexToVertex(i);
        const int c = cell_[v];
        if (c == kBlack) {
            if (head_[v] == v) heads_b[nb++] = static_cast<std::uint16_t>(v);
            const int before = NW.count;
            const int mine = NW.Join(v, l);
            if (NW.count != before) spoilt_w[mine] = 0;
        } else if (c == kWhite) {
            if (head_[v] == v) heads_w[nw++] = static_cast<std::uint16_t>(v);
            const int before = NB.count;
            const int mine = NB.Join(v, l);
            if (NB.count != before) spoilt_b[mine] = 0;
        } else {
            unsigned border = 0;
            for (int k = 0; k < 4; ++k) {
                const int ca = cell_[v + dir_[k]];
                if (ca == kBlack || ca == kWhite) border |= 1u << ca;
            }
            int before = E.count;
            int mine = E.Join(v, l);
            if (E.count != before) touch[mine] = 0;
            touch[mine] |= static_cast<std::uint8_t>(border);
            before = NB.count;
            mine = NB.Join(v, l);
            if (NB.count != before) spoilt_b[mine] = 0;
            if (!(border & (1u << kBlack))) spoilt_b[mine] = 1;
            before = NW.count;
            mine = NW.Join(v, l);
            if (NW.count != before) spoilt_w[mine] = 0;
            if (!(border & (1u << kWhite))) spoilt_w[mine] = 1;
        }
    }
    // Tromp-Taylor owner of the empty regions (ReachArea)
    for (int x = E.count; x >= 1; --x)
        if (E.parent[x] != x) touch[E.parent[x]] |= touch[x];
    for (int x = 1; x <= E.count; ++x)
        if (E.parent[x] != x) touch[x] = touch[E.parent[x]];
    for (int i = 0; i < points_; ++i) {
        const int v = IndexToVertex(i);
        const int c = cell_[v];
        if (c == kBlack || c == kWhite) {
            owner[i] = c;
        } else {
            const unsigned border = touch[E.lab[v]];
            owner[i] = border == (1u << kBlack) ? kBlack : border == (1u << kWhite) ? kWhite : kEmpty;
        }
    }
    std::memset(safe, 0, points_ * sizeof(bool));
    bool alive[kMaxPoints];
    for (int c = 0; c < 2; ++c) {
        std::memset(alive, 0, sizeof(alive));
        if (c == kBlack) PassAliveFromLabels(alive, kBlack, true, true, NB, spoilt_b, heads_b, nb);
        else PassAliveFromLabels(alive, kWhite, true, true, NW, spoilt_w, heads_w, nw);
        for (int i = 0; i < points_; ++i) {
            if (alive[i]) {
                owner[i] = c;
                safe[i] = true;
            }
        }
    }
}

int Position::ScoreOnBoard(int color, int scoring, const int* helper) const {
    int area[kMaxPoints], lead = 0;
    ScoreArea(area, scoring, helper);
    for (int i = 0; i < points_; ++i) lead += (area[i] == kBlack) - (area[i] == kWhite);
    return color == kBlack ? lead : -lead;
}

void Position::SafeArea(bool* out, bool mark_seki) const {
    std::memset(out, 0, points_ * sizeof(bool));
    PassAliveArea(out, kBlack, true, true);
    PassAliveArea(out, kWhite, true, true);
    if (mark_seki)
        for (int i = 0; i < points_; ++i)
            if (IsSeki(IndexToVertex(i))) out[i] = true;
}

// Connected groups of one feature value: ring links, group ids (from 1, 0 = on board but other value,
// -1 = off board), heads in scan order.  Ring order is fixed: members ascending, each linking to its
// predecessor, the lowest vertex linking to the highest (the order the reference's classification yields).
struct Position::Groups {
    std::int16_t id[kMaxVertices];
    std::uint16_t next[kMaxVertices];
    std::uint16_t heads[kMaxPoints];
    int count;
};

void Position::Classify(int target, const std::uint8_t* feat, Groups& g) const {
    // One raster pass with union-find instead of a flood fill per group (this is called two to four times per evaluated
    // position and was the largest single item of the host profile on the GPU box): a cell takes the label of its upper /
    // left neighbour, two labels that meet are united with the smaller one as the root.
    Labels L;
    std::memset(L.lab, 0, sizeof(L.lab));
    L.count = 0;
    for (int i = 0; i < points_; ++i) {
        const int v = IndexToVertex(i);
        if (feat[v] == target) L.Join(v, letter_);
    }
    LinkGroups(L, g);
}

void Position::LinkGroups(const Labels& L, Groups& g) const {
    // A group's smallest label is the one its first cell in scan order opened, so numbering the roots in label order is
    // numbering the groups by their first cell -- the order a flood fill from every unvisited cell finds them in.
    static_assert(kNoVertex == 0, "next[] is cleared with memset");
    std::memset(g.id, 0xff, sizeof(g.id));   // -1: off the board
    std::memset(g.next, 0, sizeof(g.next));  // kNoVertex
    // group ids: roots in label order; a label's parent is smaller than the label, so its id is already there
    std::uint16_t gid[kMaxPoints + 2];
    g.count = 0;
    for (int x = 1; x <= L.count; ++x) gid[x] = L.parent[x] == x ? static_cast<std::uint16_t>(++g.count) : gid[L.parent[x]];
    // members in ascending vertex order, each linking to its predecessor, the lowest (the head) to the highest
    std::uint16_t last[kMaxPoints + 1];
    std::memset(last, 0, sizeof(last[0]) * static_cast<size_t>(g.count + 1));
    for (int i = 0; i < points_; ++i) {
        const int v = IndexToVertex(i);
        if (!L.lab[v]) {
            g.id[v] = 0;
            g.next[v] = static_cast<std::uint16_t>(v);
            continue;
        }
        const int id = gid[L.lab[v]];
        g.id[v] = static_cast<std::int16_t>(id);
        if (last[id]) g.next[v] = last[id];
        else g.heads[id - 1] = static_cast<std::uint16_t>(v);
        last[id] = static_cast<std::uint16_t>(v);
    }
    for (int k = 0; k < g.count; ++k) g.next[g.heads[k]] = last[k + 1];
}

void Position::InnerRegions(int v, int c, const Groups& regions, bool* inner) const {
    // groups of the complement of this region that do not touch the edge are enclosed by it
    std::uint8_t surround[kMaxVertices];
    std::memset(surround, kWall, sizeof(surround));
    std::memset(inner, 0, kMaxVertices * sizeof(bool));
    for (int i = 0; i < points_; ++i) surround[IndexToVertex(i)] = kEmpty;
    int p = v;
    do {
        surround[p] = static_cast<std::uint8_t>(Opp(c));
        p = regions.next[p];
    } while (p != v);

    Groups rest_store;  // ~2.5 KB of scratch on the stack (Classify writes every field it later reads)
    Groups* const rest = &rest_store;
    Classify(kEmpty, surround, *rest);
    int cnt = rest->count;
    // NOTE: after dropping an edge-touching group the scan advances past the group that slid into its slot;
    // that group is kept without being tested (reference board.cc:2078-2097).  Kept for identical planes.
    for (int i = 0; i < cnt; ++i) {
        const int h = rest->heads[i];
        bool edge = false;
        p = h;
        do {
            for (int k = 0; k < 4; ++k) {
                if (surround[p + dir_[k]] == kWall) {
                    edge = true;
                    break;
                }
            }
            if (edge) break;
            p = rest->next[p];
        } while (p != h);
        if (edge) {
            for (int j = i; j + 1 < rest->count; ++j) rest->heads[j] = rest->heads[j + 1];
            cnt -= 1;
        }
    }
    for (int i = 0; i < cnt; ++i) {
        const int h = rest->heads[i];
        p = h;
        do {
            inner[p] = true;
            p = rest->next[p];
        } while (p != h);
    }
}

bool Position::RegionPassDead(int v, int c, const std::uint8_t* feat, const Groups& regions) const {
    // `c` tries to live inside this region; it needs two separate potential eyes.
    // Shortcut (same answer): an enclosed group only ever turns a diagonal neighbour into a friendly one, so a point that is
    // an eye without that knowledge is an eye with it; three such points settle the question (the adjacency rule below
    // only concerns exactly two) and the enclosed groups -- a classification of the whole board per region -- are not needed.
    // Open regions of ordinary positions leave here after a handful of points.
    {
        int quick = 0;
        int p = v;
        do {
            bool eye = cell_[p] != c;
            if (eye) {
                for (int k = 0; k < 4 && eye; ++k)
                    if (feat[p + dir_[k]] == Opp(c)) eye = false;
            }
            if (eye) {
                int wall = 0, opp = 0;
                for (int k = 4; k < 8; ++k) {
                    const int f = feat[p + dir_[k]];
                    wall += f == kWall;
                    opp += f == Opp(c);
                }
                if (wall == 0 ? opp > 1 : opp > 0) eye = false;
            }
            if (eye && ++quick >= 3) return false;
            p = regions.next[p];
        } while (p != v);
    }
    bool inner[kMaxVertices];
    InnerRegions(v, c, regions, inner);
    int eyes[kMaxPoints], ne = 0;
    int p = v;
    do {
        bool eye = cell_[p] != c; // own stones cannot become own eyes (suicide is forbidden)
        if (eye) {
            int side[4] = {0, 0, 0, 0};
            for (int k = 0; k < 4; ++k) side[feat[p + dir_[k]]]++;
            if (side[Opp(c)] != 0) eye = false;
        }
        if (eye) {
            int corner[4] = {0, 0, 0, 0};
            for (int k = 4; k < 8; ++k) {
                const int a = p + dir_[k];
                corner[inner[a] ? c : feat[a]]++;
            }
            if (corner[kWall] == 0 ? corner[Opp(c)] > 1 : corner[Opp(c)] > 0) eye = false;
        }
        if (eye) eyes[ne++] = p;
        p = regions.next[p];
    } while (p != v);
    if (ne == 2 && IsAdjacent(eyes[0], eyes[1])) ne = 1;
    return ne < 2;
}

void Position::PassAliveArea(bool* out, int color, bool mark_vitals, bool mark_pass_dead) const {
    // Regions of non-`color` cells, labelled in one raster pass; a label is spoilt when it holds an empty cell that touches no
    // `color` stone.  Benson's potentially vital regions are the unspoilt ones.
    Labels L;
    std::memset(L.lab, 0, sizeof(L.lab));
    L.count = 0;
    std::uint8_t spoilt[kMaxPoints + 2];
    std::uint16_t chain_head[kMaxPoints];  // the chains of `color`, by their head stones, in scan order
    int nchains = 0;
    const int l = letter_;
    for (int i = 0; i < points_; ++i) {
        const int v = IndexToVertex(i);
        const int c = cell_[v];
        if (c == color) {
            if (head_[v] == v) chain_head[nchains++] = static_cast<std::uint16_t>(v);
            continue;
        }
        const int before = L.count;
        const int mine = L.Join(v, l);
        if (L.count != before) spoilt[mine] = 0;
        if (c == kEmpty && cell_[v - l] != color && cell_[v - 1] != color && cell_[v + 1] != color && cell_[v + l] != color) spoilt[mine] = 1;
    }
    PassAliveFromLabels(out, color, mark_vitals, mark_pass_dead, L, spoilt, chain_head, nchains);
}

void Position::PassAliveFromLabels(bool* out, int color, bool mark_vitals, bool mark_pass_dead, const Labels& L, std::uint8_t* spoilt,
                                   const std::uint16_t* chain_head, int nchains) const {
    // fold the flags towards the roots (a parent is a smaller label), then hand every label its root's
    bool any_vital = false;
    for (int x = L.count; x >= 1; --x)
        if (L.parent[x] != x) spoilt[L.parent[x]] |= spoilt[x];
    for (int x = 1; x <= L.count; ++x) {
        if (L.parent[x] != x) spoilt[x] = spoilt[L.parent[x]];
        else any_vital |= !spoilt[x];
    }
    // The usual position of a game's first half has no potentially vital region at all.  Then no chain is pass-alive, no
    // region is marked, and the pass-dead step below looks at one region holding the whole board, in which the other side has
    // its two eyes as soon as three points are not its own stones (RegionPassDead's shortcut): nothing is written.
    if (!any_vital) {
        int free_points = 0;
        for (int i = 0; i < points_ && free_points < 3; ++i) free_points += cell_[IndexToVertex(i)] != Opp(color);
        if (free_points >= 3 || !mark_pass_dead) return;
    }
    // Benson's fixpoint over (region, chain) pairs instead of ring walks.  For every potentially vital region r and every
    // `color` chain c that touches it (through an empty cell or through an enemy stone of r): near[r, c] = the empty cells of
    // r next to c.  r is an eye of c when near[r, c] is ALL of r's empty cells; a chain with fewer than two eyes is dropped,
    // and every region a dropped chain touches stops being vital.  Chains that fail are dropped together -- a chain below two
    // eyes never gets one back, so the fixpoint is the one the one-at-a-time loop reaches (reference board.cc:1720-1901).
    std::uint16_t root[kMaxPoints + 2], empties[kMaxPoints + 2];
    for (int x = 1; x <= L.count; ++x) {
        root[x] = L.parent[x] == x ? static_cast<std::uint16_t>(x) : root[L.parent[x]];
        empties[x] = 0;
    }
    struct Pair { std::uint16_t r, c, near; };
    Pair pairs[4 * kMaxPoints];
    int npairs = 0;
    constexpr int kSlots = 2048;  // open addressing over (r, c); <= 4 pairs per cell of a vital region
    std::uint16_t slot_of[kSlots];
    std::memset(slot_of, 0, sizeof(slot_of));
    for (int i = 0; i < points_; ++i) {
        const int v = IndexToVertex(i);
        const int lb = L.lab[v];
        if (!lb || spoilt[lb]) continue;
        const int r = root[lb];
        const bool empty = cell_[v] == kEmpty;
        if (empty) empties[r]++;
        int seen[4], ns = 0;
        for (int k = 0; k < 4; ++k) {
            const int a = v + dir_[k];
            if (cell_[a] != color) continue;
            const int c = head_[a];
            if (Contains(seen, ns, c)) continue;
            seen[ns++] = c;
            unsigned hpos = (static_cast<unsigned>(r) * 0x9e5u + static_cast<unsigned>(c) * 0x2bu) & (kSlots - 1);
            for (;;) {
                const int idx = slot_of[hpos];
                if (idx == 0) {
                    pairs[npairs] = Pair{static_cast<std::uint16_t>(r), static_cast<std::uint16_t>(c), static_cast<std::uint16_t>(empty ? 1 : 0)};
                    slot_of[hpos] = static_cast<std::uint16_t>(++npairs);
                    break;
                }
                Pair& pr = pairs[idx - 1];
                if (pr.r == r && pr.c == c) {
                    pr.near = static_cast<std::uint16_t>(pr.near + (empty ? 1 : 0));
                    break;
                }
                hpos = (hpos + 1) & (kSlots - 1);
            }
        }
    }
    bool region_vital[kMaxPoints + 2];
    for (int x = 1; x <= L.count; ++x) region_vital[x] = L.parent[x] == x && !spoilt[x];
    std::uint8_t chain_state[kMaxVertices + 1];  // by head vertex: 1 = alive so far, 2 = dropped (read for heads of `color` chains only)
    std::uint8_t eyes[kMaxVertices + 1];
    for (int k = 0; k < nchains; ++k) chain_state[chain_head[k]] = 1;
    int alive_count = nchains;
    for (bool changed = alive_count > 0; changed;) {
        changed = false;
        for (int k = 0; k < nchains; ++k) eyes[chain_head[k]] = 0;
        for (int k = 0; k < npairs; ++k) {
            const Pair& pr = pairs[k];
            if (region_vital[pr.r] && pr.near == empties[pr.r] && eyes[pr.c] < 2) eyes[pr.c]++;
        }
        for (int k = 0; k < nchains; ++k) {
            const int h = chain_head[k];
            if (chain_state[h] == 1 && eyes[h] < 2) {
                chain_state[h] = 2;
                --alive_count;
                changed = true;
            }
        }
        if (changed)
            for (int k = 0; k < npairs; ++k)
                if (chain_state[pairs[k].c] == 2) region_vital[pairs[k].r] = false;
    }

    std::uint8_t occ[kMaxVertices];
    std::memset(occ, kWall, sizeof(occ));
    for (int i = 0; i < points_; ++i) {
        const int v = IndexToVertex(i);
        std::uint8_t o = kEmpty;
        if (cell_[v] == color) {
            if (chain_state[head_[v]] == 1) {
                out[i] = true;
                o = static_cast<std::uint8_t>(color);
            }
        } else if (mark_vitals && L.lab[v] && region_vital[root[L.lab[v]]]) {
            out[i] = true;
            o = static_cast<std::uint8_t>(color);
        }
        occ[v] = o;
    }
    Groups regions_store;
    Groups* const regions = &regions_store;
    if (mark_pass_dead) {
        bool all_empty = alive_count == 0;
        for (int i = 0; i < points_ && all_empty; ++i) all_empty = occ[IndexToVertex(i)] == kEmpty;
        if (all_empty) {
            // nothing of `color` is pass-alive (the usual case): every chain was dropped and no region stayed vital, occ is
            // empty all over -- the classification is one region holding every point (members ascending, each linking to
            // its predecessor, the lowest to the highest, as Classify links them)
            for (int v = 0; v < kMaxVertices; ++v) regions->id[v] = -1;
            std::memset(regions->next, 0, sizeof(regions->next));
            int prev = -1, first = -1;
            for (int i = 0; i < points_; ++i) {
                const int v = IndexToVertex(i);
                regions->id[v] = 1;
                if (prev >= 0) regions->next[v] = static_cast<std::uint16_t>(prev);
                else first = v;
                prev = v;
            }
            regions->next[first] = static_cast<std::uint16_t>(prev);
            regions->heads[0] = static_cast<std::uint16_t>(first);
            regions->count = 1;
        } else {
            Classify(kEmpty, occ, *regions);
        }
        for (int r = 0; r < regions->count; ++r) {
            const int h = regions->heads[r];
            if (!RegionPassDead(h, Opp(color), occ, *regions)) continue;
            int p = h;
            do {
                out[VertexToIndex(p)] = true;
                p = regions->next[p];
            } while (p != h);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Seki.
void Position::ChainSurround(int v, int c, int* libbuf, int& nl, int* chainbuf, int& nc) const {
    int p = v;
    do {
        for (int k = 0; k < 4; ++k) {
            const int a = p + dir_[k];
            if (cell_[a] == kEmpty) {
                if (!Contains(libbuf, nl, a)) libbuf[nl++] = a;
            } else if (cell_[a] == Opp(c)) {
                if (!Contains(chainbuf, nc, head_[a])) chainbuf[nc++] = head_[a];
            }
        }
        p = next_[p];
    } while (p != v);
}

bool Position::IsSeki(int v) const {
    if (cell_[v] != kEmpty) return false;
    int chain[2] = {kNoVertex, kNoVertex};
    for (int k = 0; k < 4; ++k) {
        const int a = v + dir_[k];
        const int s = cell_[a];
        if (s == kBlack || s == kWhite) {
            if (chain[s] == kNoVertex) chain[s] = head_[a];
            else if (chain[s] != head_[a]) return false; // one chain per colour
        }
    }
    for (int c = 0; c < 2; ++c)
        if (chain[c] == kNoVertex || libs_[chain[c]] != 2) return false;

    int lib[kMaxPoints], nl = 0, around_b[kMaxPoints], nb = 0, around_w[kMaxPoints], nw = 0;
    ChainSurround(chain[kBlack], kBlack, lib, nl, around_b, nb);
    ChainSurround(chain[kWhite], kWhite, lib, nl, around_w, nw);
    if (nl == 3) return true; // shared liberty plus one private liberty each (includes false seki)

    int inner = kWall;
    if (nb == 1) inner = kBlack;
    else if (nw == 1) inner = kWhite;
    if (inner == kWall) return true; // plain two-liberty seki without eyes

    // the inner chain plus this point form an eye space of the outer side: seki unless that shape is killable
    std::uint16_t eye_next[kMaxVertices];
    std::memset(eye_next, 0, sizeof(eye_next));
    int eye_size = 1;
    int nxt = chain[inner], pos;
    do {
        pos = nxt;
        nxt = next_[nxt];
        eye_next[pos] = static_cast<std::uint16_t>(nxt);
        ++eye_size;
    } while (nxt != chain[inner]);
    eye_next[pos] = static_cast<std::uint16_t>(v);
    eye_next[v] = static_cast<std::uint16_t>(nxt);
    return !KillableSekiEye(v, eye_size, eye_next);
}

bool Position::KillableSekiEye(int v, int eye_size, const std::uint16_t* eye_next) const {
    if (eye_size <= 3) return true;
    if (eye_size >= 7) return false;
    bool in_eye[kMaxVertices];
    std::memset(in_eye, 0, sizeof(in_eye));
    int border = 0, p = v;
    do {
        in_eye[p] = true;
        if (IsBorder(p)) ++border;
        p = eye_next[p];
    } while (p != v);

    // nakade points: cells that touch (8-neighbourhood) every other cell of the eye and leave diagonal eyes
    int nakade = 0;
    int first_eye_cnt = 0, first_eye_vtx = kNoVertex;
    p = v;
    do {
        int touch = 0, diag[4], nd = 0;
        for (int k = 0; k < 8; ++k) {
            const int a = p + dir_[k];
            if (in_eye[a]) {
                ++touch;
                if (k >= 4) diag[nd++] = a;
            }
        }
        if (touch + 1 == eye_size && nd > 0) {
            if (nakade == 0) {
                first_eye_cnt = nd;
                first_eye_vtx = diag[0];
            }
            ++nakade;
        }
        p = eye_next[p];
    } while (p != v);
    if (nakade == 0) return false;

    auto sides_in_eye = [&](int e) {
        int n = 0;
        for (int k = 0; k < 4; ++k) n += in_eye[e + dir_[k]] ? 1 : 0;
        return n;
    };
    const bool bulky = first_eye_cnt == 1 && sides_in_eye(first_eye_vtx) == 2;
    if (eye_size == 4) return border == 4 || bulky;      // bent four in the corner, squared four
    if (eye_size == 5) return bulky;                      // bulky five
    if (nakade == 1) return bulky;                        // rabbitty six
    if (nakade == 2) return border == 4;                  // rectangular six in the corner
    return false;
}

} // namespace sayuri_go
