// Network input planes from a GameState.
//
// Same plane semantics, order and values as the reference encoder (src/neural/encoder.h:24-61,
// src/neural/encoder.cc:14-49,101-368):  v3+ nets use encoder v2 (43 planes), v1/v2 nets encoder v1 (38).
// Writes straight into a caller-provided buffer (e.g. a pinned staging slot of the HIP pipe) instead of
// building and copying std::vectors.
#pragma once

#include "game_state.h"
#include "packed_planes.h"

namespace sayuri_go {

struct Encoder {
    static constexpr int kHistory = 8;
    static constexpr int EncoderVersion(int weights_version) { return (weights_version == 1 || weights_version == 2) ? 1 : 2; }
    static constexpr int InputChannels(int weights_version) { return EncoderVersion(weights_version) == 1 ? 38 : 43; }

    // planes: [InputChannels][board*board] of the state's own board size, already symmetry-transformed.
    static void Planes(const GameState& state, int symmetry, int weights_version, float* planes);
    // The same planes in compact form (packed_planes.h): bits for the 0/1 planes, one float per broadcast plane; the
    // symmetry is applied while the bits are set.  Expand() of the result equals Planes() bit for bit.
    static void Packed(const GameState& state, int symmetry, int weights_version, sayuri_host::PackedPlanes* out);
};

} // namespace sayuri_go
