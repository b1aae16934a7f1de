// The search benchmark (reference --mode benchmark, src/benchmark/benchmark.cc:78-161): a set of positions reached by
// policy-sampled openings, then a timed Search::Computation(playouts, kThinking) on each with a fresh tree and no
// help from the evaluation cache; playouts per second, averaged over the positions, plus KataGo's ad-hoc Elo estimate
// (benchmark.cc:14-28).
//
// One difference in kind: the reference raises throughput with threads inside ONE tree (its "tbg:threads:batch:games"
// queries); this engine runs one playout per tree at a time and gets its batches from CONCURRENT searches, so the
// benchmark's knob is how many positions are searched at once (`concurrent`).  concurrent = 1 is the latency figure
// of a single search; concurrent = 2 x batch size is the throughput figure that corresponds to self-play.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <mutex>
#include <random>
#include <thread>
#include <vector>

#include "selfplay.h"

namespace sayuri_engine {

double ComputeEloEffect(double playouts_per_move, double playouts_per_second, int threads) {  // benchmark.cc:14-28
    const double cost = threads * 7.0 * std::pow(1600.0 / (800.0 + playouts_per_move), 0.85);
    const double gain = 250.0 * std::log(playouts_per_second) / std::log(2.0);
    return gain - cost;
}

SearchBenchmarkResult RunSearchBenchmark(std::shared_ptr<NetworkForwardPipe> pipe, int weights_version, const EngineOptions& opt_in,
                                         int positions, int concurrent) {
    EngineOptions opt = opt_in;
    opt.network.no_cache = true;  // the reference clears the cache before every timed search
    Network network;
    network.Initialize(std::move(pipe), weights_version, opt.network);
    positions = std::max(1, positions);
    concurrent = std::max(1, std::min(concurrent, positions));
    std::uint64_t seed = opt.selfplay.seed ? opt.selfplay.seed : 0x5a79757269ULL;

    // ---- the test set (GenerateTestSet, benchmark.cc:78-107): N(0, bs/4) + 8 % of the board policy-sampled moves
    std::vector<GameState> set(static_cast<size_t>(positions));
    {
        std::vector<std::thread> th;
        std::atomic<int> next{0};
        for (int t = 0; t < concurrent; ++t)
            th.emplace_back([&, t] {
                Rng rng(seed + 7919ULL * static_cast<std::uint64_t>(t + 1));
                for (int n; (n = next.fetch_add(1)) < positions;) {
                    GameState& s = set[static_cast<size_t>(n)];
                    s.Reset(opt.selfplay.default_boardsize, opt.selfplay.default_komi, opt.selfplay.scoring_rule);
                    std::normal_distribution<float> dist(0.f, static_cast<float>(s.GetBoardSize()) / 4);
                    const int moves = static_cast<int>(dist(rng) + 0.08f * s.GetNumIntersections());
                    for (int i = 0; i < moves; ++i) {
                        if (s.GetPasses() >= 2) break;
                        s.PlayMove(network.GetVertexWithPolicy(s, 0.95f, false, rng));
                    }
                }
            });
        for (auto& t : th) t.join();
    }

    // ---- timed searches
    SearchBenchmarkResult res;
    res.positions = positions;
    res.concurrent = concurrent;
    network.ResetNumQueries();
    std::atomic<int> next{0};
    std::mutex mu;
    double sum_rate = 0, sum_playouts = 0, sum_elapsed = 0;
    std::string error;
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int t = 0; t < concurrent; ++t)
        th.emplace_back([&, t] {
            try {
                for (int n; (n = next.fetch_add(1)) < positions;) {
                    GameState state = set[static_cast<size_t>(n)];
                    Search search(state, network, opt.search);
                    search.Seed(seed + 2 * static_cast<std::uint64_t>(n) + 1, seed + 2 * static_cast<std::uint64_t>(n) + 2);
                    const auto a = std::chrono::steady_clock::now();
                    const ComputationResult r = search.Computation(opt.search.playouts, Search::kThinking | Search::kNoBuffer);
                    const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count();
                    std::lock_guard<std::mutex> lk(mu);
                    sum_rate += r.playouts / std::max(el, 1e-9);
                    sum_playouts += r.playouts;
                    sum_elapsed += el;
                }
            } catch (const std::exception& e) {
                std::lock_guard<std::mutex> lk(mu);
                if (error.empty()) error = e.what();
            }
            (void)t;
        });
    for (auto& t : th) t.join();
    if (!error.empty()) throw std::runtime_error("search benchmark failed: " + error);
    res.wall_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    res.playouts_per_move = sum_playouts / positions;
    res.playouts_per_second_per_search = res.playouts_per_move / (sum_elapsed / positions);  // the reference's figure (avg / avg)
    res.playouts_per_second_total = sum_playouts / res.wall_seconds;
    res.nn_queries = network.GetNumQueries();
    res.nn_evals_per_second = res.nn_queries / res.wall_seconds;
    res.elo = ComputeEloEffect(res.playouts_per_move, res.playouts_per_second_per_search, 1);
    (void)sum_rate;
    return res;
}

} // namespace sayuri_engine
