#include "encoder.h"

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
        b.SafeArea(safe, false);
        for (int i = 0; i < n; ++i)
            if (safe[i]) area[i] = 1.f;
        libs = area + n;
    } else {
        // pass-alive area by owner, then Tromp-Taylor area by owner; nothing under territory scoring
        if (state.GetScoringRule() != kTerritoryScoring) {
            int owner[kMaxPoints];
            bool safe[kMaxPoints];
            b.ScoreAndSafeArea(owner, safe);
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

} // namespace sayuri_go
