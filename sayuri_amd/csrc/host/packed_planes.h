// PackedPlanes -- the compact form of InputData::planes (the encoder on the critical path).
//
// The reference encoder (src/neural/encoder.cc:101-368) fills 43 fp32 planes per evaluation: 62 KB written by the
// encoder, copied into the batch (batch_forward_pipe.cc:15-33), sent over PCIe and read again by the first kernel.
// 37 of those planes are 0/1 (24 history, ko, 4 area, 4 liberties, 4 ladder) and the other six are one value
// broadcast over the board (rule, wave, komi/20, -komi/20, N/361, 1).  PackedPlanes carries exactly that: one bit per
// cell for the binary planes, in the SAMPLE's own cell order (bit y*board_size + x of plane c is word (y*bs+x)/32,
// bit (y*bs+x)%32), and one float per scalar plane -- 1.8 KB instead of 62 KB.  The first kernel of the forward
// (pack_bits_kernel, small_ops.h) expands it straight into the fp16 NHWC activations, so the values the network sees
// are bit-identical to the fp32 route (tests/test_gpu_net.py::test_packed_planes_give_identical_outputs).
#pragma once

#include <cstdint>
#include <cstring>

namespace sayuri_host {

struct PackedPlanes {
    static constexpr int kWords = 12;       // ceil(19*19 / 32)
    static constexpr int kMaxBinary = 40;
    static constexpr int kMaxScalars = 8;
    // binary / scalar split of the two encoder versions (encoder.h:64-77: 38 planes for v1/v2 nets, 43 for v3+)
    static constexpr int BinaryPlanes(int input_channels) { return input_channels == 38 ? 34 : input_channels - 6; }
    // 32-bit words of one record in a batch buffer: bits[binary][kWords], then kMaxScalars floats
    static constexpr int RecordWords(int binary_planes) { return binary_planes * kWords + kMaxScalars; }

    float komi{0.f};
    int board_size{-1};
    int side_to_move{-1};
    int offset{0};         // PolicyBufferOffset
    int binary_planes{0};  // planes [0, binary_planes) are bit planes, the rest scalars
    std::uint32_t bits[kMaxBinary][kWords];
    float scalars[kMaxScalars];

    void Clear(int binary) {
        binary_planes = binary;
        std::memset(bits, 0, sizeof(std::uint32_t) * kWords * static_cast<size_t>(binary));
        std::memset(scalars, 0, sizeof(scalars));
    }
    void Set(int plane, int cell) { bits[plane][cell >> 5] |= 1u << (cell & 31); }
    bool Get(int plane, int cell) const { return (bits[plane][cell >> 5] >> (cell & 31)) & 1u; }
    // the record as it travels: bits of the net's binary planes, then the scalars
    void Store(std::uint32_t* record) const {
        std::memcpy(record, bits, sizeof(std::uint32_t) * kWords * static_cast<size_t>(binary_planes));
        std::memcpy(record + binary_planes * kWords, scalars, sizeof(scalars));
    }
    // fp32 planes [channels][board_size^2] -- what InputData::planes would hold
    void Expand(int channels, float* planes) const {
        const int n = board_size * board_size;
        for (int c = 0; c < channels; ++c)
            for (int i = 0; i < n; ++i) planes[c * n + i] = c < binary_planes ? (Get(c, i) ? 1.f : 0.f) : scalars[c - binary_planes];
    }
};

}  // namespace sayuri_host
