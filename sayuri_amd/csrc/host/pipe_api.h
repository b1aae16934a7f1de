// pipe_api.h -- the plugin interface of the NN hot path, as the host side of this backend sees it.
//
// These are the request/reply PODs and the abstract pipe of the reference
// (src/neural/network_basic.h:13-161), restated so that this library builds stand-alone.
// Field names, meanings and defaults are the reference's: a HipForwardPipe compiled inside the
// reference tree includes the reference header instead and nothing else
// in this directory changes.
#pragma once

#include <array>
#include <memory>
#include <stdexcept>
#include <string>

#include "packed_planes.h"

#ifndef MAX_BOARD_SIZE
#define MAX_BOARD_SIZE (19)  // reference src/game/types.h:5-7
#endif

namespace sayuri_host {

constexpr int kBoardSize = MAX_BOARD_SIZE;
constexpr int kNumIntersections = kBoardSize * kBoardSize;
constexpr int kInputChannels = 43;  // network_basic.h:11
constexpr int kInvalidColor = 3;    // game/types.h kInvalid

// network_basic.h:15-22: which of the five policy planes a query wants
enum class PolicyBufferOffset : int {
    kNormal = 0, kOpponent = 1, kSoft = 2, kSoftOpponent = 3, kOptimistic = 4, kDefault
};

// network_basic.h:23-34.  planes = [channel][y*board_size + x], packed with the SAMPLE's board
// size; the batch collector re-pads it into the NN grid (batch_forward_pipe.cc:15-33).
struct InputData {
    InputData() { planes.fill(0.f); }
    float komi{0.f};
    int board_size{-1};
    int side_to_move{kInvalidColor};
    PolicyBufferOffset offset{PolicyBufferOffset::kDefault};
    std::array<float, kInputChannels * kNumIntersections> planes;
};

// network_basic.h:36-63.  Everything is RAW network output (pre softmax / tanh / softplus);
// Network::TransformResult post-processes it (network.cc:361-411).
struct OutputResult {
    OutputResult() {
        wdl.fill(0.f);
        probabilities.fill(0.f);
        ownership.fill(0.f);
    }
    void ImportQueryInfo(const OutputResult& other) {
        fp16 = other.fp16;
        board_size = other.board_size;
        komi = other.komi;
    }
    bool fp16{false};
    int board_size{-1};
    float komi{0.f};
    float pass_probability{0.f};
    float wdl_winrate{0.f};
    float stm_winrate{0.f};
    float final_score{0.f};
    float q_error{0.f};
    float score_error{0.f};
    PolicyBufferOffset offset{PolicyBufferOffset::kDefault};
    std::array<float, 3> wdl;
    std::array<float, kNumIntersections> probabilities;
    std::array<float, kNumIntersections> ownership;
};

// network_basic.h:104-130
struct ForwardPipeOption {
    static ForwardPipeOption Get() { return ForwardPipeOption{}; }
    ForwardPipeOption SetBoardSize(int size) { board_size = size; return *this; }
    ForwardPipeOption SetBatchSize(int size) { batch_size = size; return *this; }
    bool IsValidBoardSize() const { return board_size > 0; }
    bool IsValidBatchSize() const { return batch_size > 0; }
    int board_size{-1};
    int batch_size{-1};
};

class DNNWeights;

// network_basic.h:132-161
class NetworkForwardPipe {
public:
    virtual ~NetworkForwardPipe() = default;
    virtual void Initialize(std::shared_ptr<DNNWeights> weights) = 0;
    virtual OutputResult Forward(const InputData& input) = 0;  // blocking, re-entrant
    virtual void Construct(ForwardPipeOption option, std::shared_ptr<DNNWeights> weights) = 0;
    virtual void Release() = 0;
    virtual void Destroy() = 0;
    virtual bool Valid() const = 0;
    virtual int GetNumWorkers() const { return 0; }
    // Extension over network_basic.h:132-161: a pipe that takes the compact planes of
    // packed_planes.h says so, and the engine's encoder then never materialises the 43 fp32 planes.
    virtual bool AcceptsPacked() const { return false; }
    virtual OutputResult ForwardPacked(const PackedPlanes&) { throw std::runtime_error("this pipe does not take packed planes"); }
    std::string GetName() const;
    int GetVersion() const;
    std::shared_ptr<DNNWeights> weights_{nullptr};
};

}  // namespace sayuri_host
