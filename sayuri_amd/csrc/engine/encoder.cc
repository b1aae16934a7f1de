#include "encoder.h"

#include <emmintrin.h>

#include <algorithm>
#include <cstring>

namespace sayuri_go {

void Encoder::Planes(const GameState& state, int symmetry, int weights_version, float* out) {
    const int n = state.GetNumIntersections();
    const int version = EncoderVersion(weights_version);
    const int channels = InputChannels(weights_version);
    const int me = state.GetToMove(), you = Opp(me);
    const Position& b = state.board_;

    // build un-rotated planes in `raw` only when a symmetry is requested; otherwise write in place
    float scratch[43 * kMaxPoints];
    float* raw = (symmetry == SymmetryTables::kIdentity) ? out : scratch;
    std::memset(raw, 0, sizeof(float) * static_cast<size_t>(channels) * n);

    // 1-24: the last eight boards, (mover's stones, opponent's stones, the move just played)
    const int past = std::min(state.GetMoveNumber() + 1, kHistory);
    for (int p = 0; p < past; ++p) {
        const Frame& f = state.Past(p);
        float* mine = raw + (3 * p + 0) * n;
        float* yours = raw + (3 * p + 1) * n;
        float* move = raw + (3 * p + 2) * n;
        for (int i = 0; i < n; ++i) {
            if (f.stones[i] == me) mine[i] = 1.f;
            else if (f.stones[i] == you) yours[i] = 1.f;
        }
        const int lm = f.last_move;
        if (lm != kNoVertex && lm != kPassMove && lm != kResignMove) move[b.VertexToIndex(lm)] = 1.f;
    }

    float* feat = raw + 3 * kHistory * n;
    // ko point
    if (b.KoMove() != kNoVertex) feat[b.VertexToIndex(b.KoMove())] = 1.f;
    float* area = feat + n;
    float* libs;
    if (version == 1) {
        bool safe[kMaxPoints];
        state.SafeAreaCached(safe);
        for (int i = 0; i < n; ++i)
            if (safe[i]) area[i] = 1.f;
        libs = area + n;
    } else {
        // pass-alive area by owner, then Tromp-Taylor area by owner; nothing under territory scoring
        if (state.GetScoringRule() != kTerritoryScoring) {
            int owner[kMaxPoints];
            bool safe[kMaxPoints];
            state.ScoreAndSafeAreaCached(owner, safe);
            for (int i = 0; i < n; ++i) {
                if (safe[i]) {
                    if (owner[i] == me) area[i] = 1.f;
                    else if (owner[i] == you) area[i + n] = 1.f;
                }
                if (owner[i] == me) area[i + 2 * n] = 1.f;
                else if (owner[i] == you) area[i + 3 * n] = 1.f;
            }
        }
        libs = area + 4 * n;
    }
    // chains with 1..4 liberties
    for (int i = 0; i < n; ++i) {
        const int v = b.IndexToVertex(i);
        const int s = b.At(v);
        if (s == kBlack || s == kWhite) {
            const int l = b.Liberties(v);
            if (l >= 1 && l <= 4) libs[i + (l - 1) * n] = 1.f;
        }
    }
    // ladders: dead / escapable chain stones, atari / capture points
    float* ladder = libs + 4 * n;
    std::uint8_t marks[kMaxPoints];
    b.LadderMap(marks);
    for (int i = 0; i < n; ++i) {
        switch (marks[i]) {
            case kLadderDeath: ladder[i] = 1.f; break;
            case kLadderEscapable: ladder[i + n] = 1.f; break;
            case kLadderAtari: ladder[i + 2 * n] = 1.f; break;
            case kLadderTake: ladder[i + 3 * n] = 1.f; break;
            default: break;
        }
    }
    // scalars broadcast over the board
    float* misc = ladder + 4 * n;
    float komi = state.GetKomiWithPenalty();
    if (me == kWhite) komi = 0.0f - komi;
    auto fill = [&](int plane, float v) { std::fill(misc + plane * n, misc + (plane + 1) * n, v); };
    // komi/20 and N/361 as reciprocal multiplies: that is what the reference binary computes (its build uses
    // -ffast-math, CMakeLists.txt:192/199, which rewrites the divisions of encoder.cc:277-316), and it keeps
    // the planes bit-identical rather than 1 ulp apart.
    const float komi_plane = komi * 0.05f;
    const float size_plane = static_cast<float>(n) * (1.f / 361.f);
    if (version == 1) {
        fill(0, komi_plane);
        fill(1, -komi_plane);
        fill(2, size_plane);
        fill(3, 1.f);
    } else {
        fill(0, state.GetScoringRule() == kAreaScoring ? 0.f : 1.f);
        fill(1, state.GetWave());
        fill(2, komi_plane);
        fill(3, -komi_plane);
        fill(4, size_plane);
        fill(5, 1.f);
    }

    if (raw != out) {
        const SymmetryTables& t = SymmetryTables::Get();
        const int bs = state.GetBoardSize();
        for (int c = 0; c < channels; ++c) {
            const float* src = raw + c * n;
            float* dst = out + c * n;
            for (int i = 0; i < n; ++i) dst[i] = src[t.Index(bs, symmetry, i)];
        }
    }
}

// The compact encoder: every 0/1 feature is first noted as a bit of a per-cell plane mask (indexed by the raw cell), then
// one pass over the output cells sets the plane bits through the symmetry map -- work proportional to the stones and
// marks on the board, not to 43 x 361 floats.  Feature semantics as in Planes() above (reference encoder.cc:101-368).
void Encoder::Packed(const GameState& state, int symmetry, int weights_version, sayuri_host::PackedPlanes* out) {
    const int n = state.GetNumIntersections();
    const int version = EncoderVersion(weights_version);
    const int channels = InputChannels(weights_version);
    const int binary = sayuri_host::PackedPlanes::BinaryPlanes(channels);
    const int me = state.GetToMove(), you = Opp(me);
    const Position& b = state.board_;
    out->Clear(binary);
    out->board_size = state.GetBoardSize();
    out->side_to_move = me;
    out->komi = state.GetKomi();

    std::uint64_t mask[kMaxPoints];
    auto bit = [](int plane) { return std::uint64_t{1} << plane; };
    const int past = std::min(state.GetMoveNumber() + 1, kHistory);
    std::fill(mask, mask + n, std::uint64_t{0});
    {
        // The 16 stone planes of the history hold most of the record's bits.  They do not go through the per-cell masks: a
        // frame's stones, taken through the symmetry map once (output cell d shows raw cell Index(symmetry, d)), are compared
        // with the colour 16 cells at a time and the compare mask IS the plane's next 16 bits (SSE2, part of x86-64).
        const SymmetryTables& st = SymmetryTables::Get();
        const int bs0 = state.GetBoardSize();
        const bool ident = symmetry == SymmetryTables::kIdentity;
        alignas(16) std::uint8_t turned[kMaxPoints + 16];
        const __m128i vme = _mm_set1_epi8(static_cast<char>(me)), vyou = _mm_set1_epi8(static_cast<char>(you));
        for (int p = 0; p < past; ++p) {
            const Frame& f = state.Past(p);
            const std::uint8_t* src = f.stones;  // (reading 16 bytes from the last chunk's start stays inside the Frame)
            static_assert(sizeof(Frame) >= ((kMaxPoints + 15) / 16) * 16, "the last 16-byte chunk of a frame's stones must lie inside the frame");
            if (!ident) {
                for (int d = 0; d < n; ++d) turned[d] = f.stones[st.Index(bs0, symmetry, d)];
                src = turned;
            }
            std::uint32_t* mine = out->bits[3 * p];
            std::uint32_t* yours = out->bits[3 * p + 1];
            for (int c16 = 0; c16 * 16 < n; ++c16) {
                const __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + c16 * 16));
                std::uint32_t a = static_cast<std::uint32_t>(_mm_movemask_epi8(_mm_cmpeq_epi8(v, vme)));
                std::uint32_t y = static_cast<std::uint32_t>(_mm_movemask_epi8(_mm_cmpeq_epi8(v, vyou)));
                const int left = n - c16 * 16;
                if (left < 16) {
                    a &= (1u << left) - 1;
                    y &= (1u << left) - 1;
                }
                mine[c16 >> 1] |= a << (16 * (c16 & 1));
                yours[c16 >> 1] |= y << (16 * (c16 & 1));
            }
            const int lm = f.last_move;
            if (lm != kNoVertex && lm != kPassMove && lm != kResignMove) mask[b.VertexToIndex(lm)] |= bit(3 * p + 2);
        }
    }
    int plane = 3 * kHistory;
    if (b.KoMove() != kNoVertex) mask[b.VertexToIndex(b.KoMove())] |= bit(plane);
    ++plane;
    if (version == 1) {
        bool safe[kMaxPoints];
        state.SafeAreaCached(safe);
        for (int i = 0; i < n; ++i)
            if (safe[i]) mask[i] |= bit(plane);
        plane += 1;
    } else {
        if (state.GetScoringRule() != kTerritoryScoring) {
            int owner[kMaxPoints];
            bool safe[kMaxPoints];
            state.ScoreAndSafeAreaCached(owner, safe);
            std::uint64_t own_bits[4] = {0, 0, 0, 0}, safe_bits[4] = {0, 0, 0, 0};
            own_bits[me] = bit(plane + 2);
            own_bits[you] = bit(plane + 3);
            safe_bits[me] = bit(plane);
            safe_bits[you] = bit(plane + 1);
            for (int i = 0; i < n; ++i) mask[i] |= own_bits[owner[i] & 3] | (safe[i] ? safe_bits[owner[i] & 3] : std::uint64_t{0});
        }
        plane += 4;
    }
    // liberties (1-4 of a stone's chain) and ladder marks: one pass, table lookups instead of branches per cell
    std::uint8_t marks[kMaxPoints];
    b.LadderMap(marks);
    {
        const int lib_plane = plane, ladder_plane = plane + 4;
        std::uint64_t by_libs[6] = {0, bit(lib_plane), bit(lib_plane + 1), bit(lib_plane + 2), bit(lib_plane + 3), 0};
        std::uint64_t by_mark[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        by_mark[kLadderDeath] = bit(ladder_plane);
        by_mark[kLadderEscapable] = bit(ladder_plane + 1);
        by_mark[kLadderAtari] = bit(ladder_plane + 2);
        by_mark[kLadderTake] = bit(ladder_plane + 3);
        for (int i = 0; i < n; ++i) {
            const int v = b.IndexToVertex(i);
            const int s = b.At(v);
            const int l = (s == kBlack || s == kWhite) ? std::min(b.Liberties(v), 5) : 0;
            mask[i] |= by_libs[l] | by_mark[marks[i] & 7];
        }
    }
    plane += 4;
    // output cell d shows raw cell Index(symmetry, d)
    const SymmetryTables& t = SymmetryTables::Get();
    const int bs = state.GetBoardSize();
    const bool identity = symmetry == SymmetryTables::kIdentity;
    for (int d = 0; d < n; ++d) {
        std::uint64_t m = mask[identity ? d : t.Index(bs, symmetry, d)];
        while (m) {
            out->Set(__builtin_ctzll(m), d);
            m &= m - 1;
        }
    }
    // scalars: the same arithmetic as Planes()
    float komi = state.GetKomiWithPenalty();
    if (me == kWhite) komi = 0.0f - komi;
    const float komi_plane = komi * 0.05f;
    const float size_plane = static_cast<float>(n) * (1.f / 361.f);
    if (version == 1) {
        out->scalars[0] = komi_plane;
        out->scalars[1] = -komi_plane;
        out->scalars[2] = size_plane;
        out->scalars[3] = 1.f;
    } else {
        out->scalars[0] = state.GetScoringRule() == kAreaScoring ? 0.f : 1.f;
        out->scalars[1] = state.GetWave();
        out->scalars[2] = komi_plane;
        out->scalars[3] = -komi_plane;
        out->scalars[4] = size_plane;
        out->scalars[5] = 1.f;
    }
}

} // namespace sayuri_go
