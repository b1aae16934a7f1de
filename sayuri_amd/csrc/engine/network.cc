#include "network.h"

#include <algorithm>
#include <cmath>
#include <random>

#include "encoder.h"

namespace sayuri_engine {

using sayuri_go::Encoder;
using sayuri_go::SymmetryTables;

// Kept in the shape of the reference's Softmax (utils/logits.h:22-39): values are pushed one by one into a
// fresh vector, which also keeps the compiler from vectorising the exp() loop with libmvec -- the vector exp
// rounds differently from the scalar one the reference binary ends up calling.
std::vector<float> Softmax(const std::vector<float>& logits, double temp) {
    std::vector<float> output;
    output.reserve(logits.size());
    const float alpha = *std::max_element(logits.begin(), logits.end());
    double denom = 0.0;
    for (const float logit : logits) {
        const double val = std::exp((logit - alpha) / temp);
        denom += val;
        output.emplace_back(static_cast<float>(val));
    }
    for (auto& out : output) out /= denom;
    return output;
}

void SoftmaxInPlace(float* x, int n, double temp) {
    const std::vector<float> in(x, x + n);
    const std::vector<float> out = Softmax(in, temp);
    std::copy(out.begin(), out.end(), x);
}

// ---------------------------------------------------------------------------------------------
void ResultCache::SetCapacity(size_t size) {
    if (size % kClusterSize) size = (size / kClusterSize + 1) * kClusterSize;
    blocks_ = size / kClusterSize;
    table_.assign(size, Entry{});
    table_.shrink_to_fit();
}

void ResultCache::Clear() {
    for (auto& e : table_) e.generation = 0;
    for (auto& s : shards_) s.generation = 0;
}

void ResultCache::Insert(std::uint64_t key, const OutputResult& value) {
    if (blocks_ == 0) return;
    const size_t block = key % blocks_;
    Entry* cluster = table_.data() + block * kClusterSize;
    Shard& sh = shards_[block % kShards];
    std::lock_guard<std::mutex> lock(sh.mu);
    size_t victim = 0;
    std::uint64_t oldest = cluster[0].generation;
    for (size_t i = 1; i < kClusterSize; ++i) {
        if (oldest > cluster[i].generation) {
            oldest = cluster[i].generation;
            victim = i;
        }
    }
    Entry& e = cluster[victim];
    e.key = key;
    e.generation = ++sh.generation;
    e.value = value;
}

bool ResultCache::Lookup(std::uint64_t key, OutputResult& value) {
    if (blocks_ == 0) return false;
    lookups_.fetch_add(1, std::memory_order_relaxed);
    const size_t block = key % blocks_;
    Entry* cluster = table_.data() + block * kClusterSize;
    Shard& sh = shards_[block % kShards];
    std::lock_guard<std::mutex> lock(sh.mu);
    for (size_t i = 0; i < kClusterSize; ++i) {
        if (cluster[i].key == key && cluster[i].generation != 0) {
            value = cluster[i].value;
            hits_.fetch_add(1, std::memory_order_relaxed);
            return true;
        }
    }
    return false;
}

// ---------------------------------------------------------------------------------------------
void Network::Initialize(std::shared_ptr<NetworkForwardPipe> pipe, int weights_version, const NetworkOptions& opt) {
    pipe_ = std::move(pipe);
    version_ = weights_version;
    opt_ = opt;
    if (!Valid()) opt_.no_cache = false; // network.cc:84-88: the cache has no effect on the dummy pipe
    SetCacheSize(opt_.cache_memory_mib);
    ResetNumQueries();
}

size_t Network::SetCacheSize(size_t mib) {
    const size_t mem_mib = std::min(std::max<size_t>(5, mib), size_t{128} * 1024);
    const size_t entries = mem_mib * 1024 * 1024 / ResultCache::kEntrySize + 1;
    opt_.cache_memory_mib = mem_mib;
    cache_.SetCapacity(entries);
    return entries;
}

Network::Result Network::DummyForward(const InputData& inputs, Rng& rng) const {
    Result r{};
    std::uniform_real_distribution<float> dist(0, 1);
    const int n = inputs.board_size * inputs.board_size;
    r.board_size = inputs.board_size;
    for (int i = 0; i < 3; ++i) r.wdl[i] = dist(rng);
    r.stm_winrate = 0.f;
    for (int i = 0; i < n; ++i) r.probabilities[i] = dist(rng);
    r.pass_probability = dist(rng);
    return r;
}

Network::Result Network::GetOutputInternal(const GameState& state, int symmetry, PolicyBufferOffset offset, Rng& rng) {
    if (Valid() && opt_.packed_inputs && pipe_->AcceptsPacked()) {
        // the compact route (packed_planes.h): 37 bit planes + 6 scalars straight into the pipe's pinned staging, expanded
        // on the GPU -- the 43 fp32 planes (62 KB written here, copied, sent, read) never exist
        sayuri_host::PackedPlanes pk;
        Encoder::Packed(state, symmetry, version_, &pk);
        pk.offset = static_cast<int>((offset == PolicyBufferOffset::kDefault) ? opt_.default_policy_offset : offset);
        num_queries_.fetch_add(1, std::memory_order_relaxed);
        Result result = pipe_->ForwardPacked(pk);
        TransformResult(result, symmetry);
        return result;
    }
    InputData in; // 62 KB of planes on the caller's stack: no sharing between threads (or fibers)
    in.board_size = state.GetBoardSize();
    in.side_to_move = state.GetToMove();
    in.komi = state.GetKomi();
    Encoder::Planes(state, symmetry, version_, in.planes.data());
    in.offset = (offset == PolicyBufferOffset::kDefault) ? opt_.default_policy_offset : offset;

    Result result;
    if (Valid()) {
        num_queries_.fetch_add(1, std::memory_order_relaxed);
        result = pipe_->Forward(in);
    } else {
        result = DummyForward(in, rng);
    }
    TransformResult(result, symmetry);
    return result;
}

bool Network::ProbeCache(const GameState& state, Result& result) {
    if (cache_.Lookup(state.GetHash(), result)) {
        if (result.board_size == state.GetBoardSize()) return true;
    }
    if (state.GetBoardSize() >= state.GetMoveNumber() && opt_.early_symm_cache) {
        // early in the game a rotated / mirrored twin may already be cached
        const SymmetryTables& t = SymmetryTables::Get();
        for (int symm = SymmetryTables::kIdentity + 1; symm < SymmetryTables::kCount; ++symm) {
            if (!cache_.Lookup(state.ComputeSymmetryHash(symm), result)) continue;
            if (result.board_size != state.GetBoardSize()) break;
            const int bs = result.board_size, n = state.GetNumIntersections();
            float prob[sayuri_go::kMaxPoints], own[sayuri_go::kMaxPoints];
            std::copy(result.probabilities.begin(), result.probabilities.begin() + n, prob);
            std::copy(result.ownership.begin(), result.ownership.begin() + n, own);
            for (int i = 0; i < n; ++i) {
                const int j = t.Index(bs, symm, i);
                result.probabilities[i] = prob[j];
                result.ownership[i] = own[j];
            }
            return true;
        }
    }
    return false;
}

Network::Result Network::GetOutput(const GameState& state, Ensemble ensemble, Query query, Rng& rng) {
    if (ensemble == kDirect) {
        if (query.symmetry < 0 || query.symmetry >= SymmetryTables::kCount) query.symmetry = SymmetryTables::kIdentity;
    } else if (ensemble == kRandom) {
        query.symmetry = static_cast<int>(rng.Below(SymmetryTables::kCount));
    }
    if (opt_.no_cache || ensemble == kAverage) query.read_cache = query.write_cache = false;

    Result result;
    if (query.read_cache && ProbeCache(state, result)) {
        ActivatePolicy(result, query.temperature);
        return result;
    }
    if (ensemble == kAverage) {
        constexpr int k = SymmetryTables::kCount;
        for (int symm = 0; symm < k; ++symm) {
            Result one = GetOutputInternal(state, symm, query.offset, rng);
            const int n = one.board_size * one.board_size;
            ActivatePolicy(one, query.temperature);
            result.pass_probability += one.pass_probability / k;
            result.wdl_winrate += one.wdl_winrate / k;
            result.stm_winrate += one.stm_winrate / k;
            result.final_score += one.final_score / k;
            result.q_error += one.q_error / k;
            result.score_error += one.score_error / k;
            for (int i = 0; i < 3; ++i) result.wdl[i] += one.wdl[i] / k;
            for (int i = 0; i < n; ++i) {
                result.probabilities[i] += one.probabilities[i] / k;
                result.ownership[i] += one.ownership[i] / k;
            }
            if (symm == SymmetryTables::kIdentity) result.ImportQueryInfo(one);
        }
    } else {
        result = GetOutputInternal(state, query.symmetry, query.offset, rng);
        if (query.write_cache) cache_.Insert(state.GetHash(), result); // stored before the policy softmax
        ActivatePolicy(result, query.temperature);
    }
    return result;
}

void Network::TransformResult(Result& result, int symmetry) {
    const int bs = result.board_size, n = bs * bs;
    const SymmetryTables& t = SymmetryTables::Get();
    float prob[sayuri_go::kMaxPoints], own[sayuri_go::kMaxPoints];
    std::copy(result.probabilities.begin(), result.probabilities.begin() + n, prob);
    std::copy(result.ownership.begin(), result.ownership.begin() + n, own);
    for (int i = 0; i < n; ++i) {
        const int j = t.Index(bs, symmetry, i);
        result.probabilities[j] = prob[i];
        result.ownership[j] = std::tanh(own[i]);
    }
    result.final_score = 20 * result.final_score;
    float wdl[3] = {result.wdl[0], result.wdl[1], result.wdl[2]};
    SoftmaxInPlace(wdl, 3, 1);
    result.wdl[0] = wdl[0];
    result.wdl[1] = wdl[1];
    result.wdl[2] = wdl[2];
    result.wdl_winrate = (wdl[0] - wdl[2] + 1.f) / 2;
    result.stm_winrate = (std::tanh(result.stm_winrate) + 1.f) / 2;
    auto softplus_square = [](float x) -> float {
        if (x <= 20.f) x = std::log(1.f + std::exp(x));
        return (x * x) / 4.f;
    };
    result.q_error = static_cast<float>(0.25 * softplus_square(result.q_error));
    result.score_error = 150 * softplus_square(result.score_error);
}

void Network::ActivatePolicy(Result& result, float temperature) {
    const int n = result.board_size * result.board_size;
    float buf[sayuri_go::kMaxPoints + 1];
    std::copy(result.probabilities.begin(), result.probabilities.begin() + n, buf);
    buf[n] = result.pass_probability;
    SoftmaxInPlace(buf, n + 1, temperature);
    std::copy(buf, buf + n, result.probabilities.begin());
    result.pass_probability = buf[n];
}

int Network::GetVertexWithPolicy(const GameState& state, float temperature, bool allow_pass, Rng& rng) {
    const Result result = GetOutput(state, kRandom, Query::Get().SetTemperature(temperature), rng);
    const int n = result.board_size * result.board_size;
    float accum = 0.0f;
    std::vector<std::pair<float, int>> table;
    for (int i = 0; i < n; ++i) {
        const int vtx = state.IndexToVertex(i);
        if (state.IsLegalMove(vtx)) {
            accum += result.probabilities[i];
            table.emplace_back(accum, vtx);
        }
    }
    if (table.empty() || allow_pass) {
        accum += result.pass_probability;
        table.emplace_back(accum, sayuri_go::kPassMove);
    }
    std::uniform_real_distribution<float> dist(0.0f, accum);
    const float pick = dist(rng);
    for (const auto& e : table)
        if (pick < e.first) return e.second;
    return sayuri_go::kNoVertex;
}

} // namespace sayuri_engine
