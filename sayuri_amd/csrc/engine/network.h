// Network -- the evaluation facade between the search and a NetworkForwardPipe backend.
//
// Same contract as the reference's `class Network` (src/neural/network.h:17-98, network.cc:141-498):
// pick a symmetry (direct / random / average over 8), probe and fill the position cache keyed by
// GameState::GetHash(), encode, call the pipe, undo the symmetry and post-process the raw outputs
// (tanh ownership, softmax wdl, score x20, softplus error heads), then the policy softmax at the
// query's temperature.  The backend is any sayuri_host::NetworkForwardPipe: HipForwardPipe in the
// product; parity tests plug the reference's CPU pipe in through a function pointer.
//
// Differences in mechanism, not in results:
//   * the cache is sharded (one small lock per shard of 8-way clusters) instead of one global spin
//     lock (utils/cache.h:52), and entries live inline in one flat table instead of one heap block each;
//   * random numbers come from the caller's Rng stream (one per game / search thread) instead of a
//     hidden thread_local generator.
#pragma once

#include <atomic>
#include <cstdint>
#include <memory>
#include <mutex>
#include <vector>

#include "game_state.h"
#include "pipe_api.h"

namespace sayuri_engine {

using sayuri_go::GameState;
using sayuri_go::Rng;
using sayuri_host::InputData;
using sayuri_host::NetworkForwardPipe;
using sayuri_host::OutputResult;
using sayuri_host::PolicyBufferOffset;

// network_basic.h:65-102
struct ForwardQuery {
    static ForwardQuery Get() { return ForwardQuery{}; }
    ForwardQuery SetTemperature(float t) { temperature = t; return *this; }
    ForwardQuery SetSymmetry(int s) { symmetry = s; return *this; }
    ForwardQuery SetCache(bool use) { read_cache = write_cache = use; return *this; }
    ForwardQuery SetOffset(PolicyBufferOffset o) {
        read_cache = write_cache = (o == PolicyBufferOffset::kDefault);
        offset = o;
        return *this;
    }
    float temperature{1.0f};
    int symmetry{-1};
    bool read_cache{true};
    bool write_cache{true};
    PolicyBufferOffset offset{PolicyBufferOffset::kDefault};
};

// Position cache: 8-way clusters, least-recently-inserted eviction (utils/cache.h).
class ResultCache {
public:
    static constexpr size_t kClusterSize = 8;
    static constexpr size_t kEntrySize = 24 + sizeof(OutputResult); // what the reference charges per entry
    void SetCapacity(size_t entries);
    void Insert(std::uint64_t key, const OutputResult& value);
    bool Lookup(std::uint64_t key, OutputResult& value);
    void Clear();
    size_t Capacity() const { return table_.size(); }
    size_t hits() const { return hits_.load(std::memory_order_relaxed); }
    size_t lookups() const { return lookups_.load(std::memory_order_relaxed); }

private:
    struct Entry {
        std::uint64_t key{0};
        std::uint64_t generation{0};
        OutputResult value;
    };
    struct alignas(64) Shard {
        std::mutex mu;
        std::uint64_t generation{0};
    };
    static constexpr size_t kShards = 256;
    std::vector<Entry> table_;
    size_t blocks_{0};
    Shard shards_[kShards];
    std::atomic<size_t> hits_{0}, lookups_{0};
};

struct NetworkOptions {
    PolicyBufferOffset default_policy_offset{PolicyBufferOffset::kNormal}; // "policy_buffer_offset"
    bool no_cache{false};
    bool early_symm_cache{false};
    size_t cache_memory_mib{400};
    bool packed_inputs{true};  // compact planes (packed_planes.h) when the pipe takes them; false = 43 fp32 planes as the reference
};

class Network {
public:
    enum Ensemble { kDirect, kRandom, kAverage };
    using Result = OutputResult;
    using Query = ForwardQuery;

    void Initialize(std::shared_ptr<NetworkForwardPipe> pipe, int weights_version, const NetworkOptions& opt);
    bool Valid() const { return pipe_ && pipe_->Valid(); }
    Result GetOutput(const GameState& state, Ensemble ensemble, Query query, Rng& rng); // network.cc:237-291
    int GetVertexWithPolicy(const GameState& state, float temperature, bool allow_pass, Rng& rng); // :431-468
    size_t SetCacheSize(size_t mib); // network.cc:102-122
    void ClearCache() { cache_.Clear(); }
    void ResetNumQueries(size_t q = 0) { num_queries_.store(q, std::memory_order_relaxed); }
    size_t GetNumQueries() const { return num_queries_.load(std::memory_order_relaxed); }
    PolicyBufferOffset GetDefaultPolicyOffset() const { return opt_.default_policy_offset; }
    int GetVersion() const { return version_; }
    const ResultCache& cache() const { return cache_; }
    NetworkForwardPipe* pipe() const { return pipe_.get(); }

    static void TransformResult(Result& result, int symmetry);          // network.cc:361-411
    static void ActivatePolicy(Result& result, float temperature);      // network.cc:413-429

private:
    Result GetOutputInternal(const GameState& state, int symmetry, PolicyBufferOffset offset, Rng& rng);
    bool ProbeCache(const GameState& state, Result& result); // network.cc:197-235
    Result DummyForward(const InputData& inputs, Rng& rng) const; // network.cc:144-165

    std::shared_ptr<NetworkForwardPipe> pipe_;
    ResultCache cache_;
    NetworkOptions opt_;
    int version_{4};
    std::atomic<size_t> num_queries_{0};
};

// y_i = exp((x_i - max)/temp) / sum, double accumulator (utils/logits.h:22-39).
std::vector<float> Softmax(const std::vector<float>& logits, double temp);
void SoftmaxInPlace(float* x, int n, double temp);

} // namespace sayuri_engine
