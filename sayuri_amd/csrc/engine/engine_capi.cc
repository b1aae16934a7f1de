// C entry points of the evaluation facade and the tree search for the Python face and the parity tests.
// Result layouts match the oracle taps of oracle/ref_search_driver.cc field for field.
#include <cstdint>
#include <cstring>
#include <memory>
#include <sstream>
#include <string>

#include "../../../include/sayuri_engine.h"
#include "engine_options.h"
#include "search.h"
#include "selfplay.h"

using namespace sayuri_engine;
using sayuri_go::kNoVertex;
using sayuri_go::kPassMove;
using sayuri_go::kResignMove;

namespace {

// A NetworkForwardPipe that forwards to a C function: parity tests plug a CPU pipe in here (tests only -- the
// product backend is HipForwardPipe).  Two shapes are accepted:
//   kind 0: fn(board_size, komi, side_to_move, offset, planes, out)        e.g. the oracle tap ref_forward
//   kind 1: fn(user, board_size, komi, offset, planes, out)                e.g. the oracle port so_forward
// out: prob[N], own[N], pass, wdl[3], stm, score, q_err, score_err, offset   (N = board_size^2, raw outputs)
using ForwardFn0 = int (*)(int, float, int, int, const float*, float*);
using ForwardFn1 = int (*)(const void*, int, float, int, const float*, float*);

class CallbackPipe : public sayuri_host::NetworkForwardPipe {
public:
    CallbackPipe(void* fn, int kind, const void* user) : fn_(fn), kind_(kind), user_(user) {}
    void Initialize(std::shared_ptr<sayuri_host::DNNWeights>) override {}
    OutputResult Forward(const InputData& in) override {
        const int n = in.board_size * in.board_size;
        float out[2 * sayuri_go::kMaxPoints + 16];
        const int offset = static_cast<int>(in.offset);
        if (kind_ == 0) reinterpret_cast<ForwardFn0>(fn_)(in.board_size, in.komi, in.side_to_move, offset, in.planes.data(), out);
        else reinterpret_cast<ForwardFn1>(fn_)(user_, in.board_size, in.komi, offset, in.planes.data(), out);
        OutputResult r;
        r.board_size = in.board_size;
        r.komi = in.komi;
        r.offset = in.offset;
        std::memcpy(r.probabilities.data(), out, sizeof(float) * static_cast<size_t>(n));
        std::memcpy(r.ownership.data(), out + n, sizeof(float) * static_cast<size_t>(n));
        const float* s = out + 2 * n;
        r.pass_probability = s[0];
        r.wdl = {s[1], s[2], s[3]};
        r.stm_winrate = s[4];
        r.final_score = s[5];
        r.q_error = s[6];
        r.score_error = s[7];
        return r;
    }
    void Construct(sayuri_host::ForwardPipeOption, std::shared_ptr<sayuri_host::DNNWeights>) override {}
    void Release() override {}
    void Destroy() override {}
    bool Valid() const override { return fn_ != nullptr; }

private:
    void* fn_;
    int kind_;
    const void* user_;
};

int MoveToIndex(const GameState& g, int v) {
    if (v == kNoVertex) return -3;
    if (v == kPassMove) return g.GetNumIntersections();
    if (v == kResignMove) return -1;
    return g.VertexToIndex(v);
}

void Export(const GameState& g, const ComputationResult& r, int* ints, float* floats, int* visits, float* estq,
            float* target, float* own) {
    const int n = g.GetNumIntersections();
    ints[0] = MoveToIndex(g, r.best_move);
    ints[1] = MoveToIndex(g, r.best_no_pass_move);
    ints[2] = MoveToIndex(g, r.random_move);
    ints[3] = MoveToIndex(g, r.gumbel_move);
    ints[4] = MoveToIndex(g, r.gumbel_no_pass_move);
    ints[5] = MoveToIndex(g, r.capture_all_dead_move);
    ints[6] = MoveToIndex(g, r.high_priority_move);
    ints[7] = r.visits;
    ints[8] = r.playouts;
    ints[9] = r.to_move;
    ints[10] = r.side_resign;
    floats[0] = r.root_eval;
    floats[1] = r.root_score_lead;
    floats[2] = r.best_eval;
    floats[3] = r.root_score_stddev;
    floats[4] = r.root_eval_stddev;
    floats[5] = r.policy_kld;
    if (static_cast<int>(r.root_searched_visits.size()) == n + 1) {
        for (int i = 0; i <= n; ++i) {
            visits[i] = r.root_searched_visits[static_cast<size_t>(i)];
            estq[i] = r.root_estimated_q[static_cast<size_t>(i)];
            target[i] = r.target_policy_dist[static_cast<size_t>(i)];
        }
        for (int i = 0; i < n; ++i) own[i] = r.root_ownership[static_cast<size_t>(i)];
    }
}

} // namespace

extern "C" {

// options: "key=value key=value ..." (names of the reference's option map, engine_options.h)
void* sayuri_engine_net_new_callback(void* forward_fn, int kind, const void* user, int weights_version, const char* options) {
    EngineOptions opt;
    opt.Parse(options ? options : "");
    auto* net = new Network();
    std::shared_ptr<NetworkForwardPipe> pipe;
    if (forward_fn) pipe = std::make_shared<CallbackPipe>(forward_fn, kind, user);
    net->Initialize(pipe, weights_version, opt.network);
    return net;
}
void sayuri_engine_net_free(void* n) { delete static_cast<Network*>(n); }
unsigned long sayuri_engine_net_queries(void* n) { return static_cast<Network*>(n)->GetNumQueries(); }

void sayuri_engine_net_output(void* n, void* game, int ensemble, int symmetry, float temperature, int use_cache,
                              std::uint64_t seed, float* out) {
    auto* g = static_cast<GameState*>(game);
    Rng rng(seed);
    auto q = Network::Query::Get().SetTemperature(temperature).SetSymmetry(symmetry).SetCache(use_cache != 0);
    auto r = static_cast<Network*>(n)->GetOutput(*g, static_cast<Network::Ensemble>(ensemble), q, rng);
    const int N = g->GetNumIntersections();
    std::memcpy(out, r.probabilities.data(), sizeof(float) * static_cast<size_t>(N));
    std::memcpy(out + N, r.ownership.data(), sizeof(float) * static_cast<size_t>(N));
    float* s = out + 2 * N;
    s[0] = r.pass_probability;
    s[1] = r.wdl[0];
    s[2] = r.wdl[1];
    s[3] = r.wdl[2];
    s[4] = r.wdl_winrate;
    s[5] = r.stm_winrate;
    s[6] = r.final_score;
    s[7] = r.q_error;
    s[8] = r.score_error;
}

void* sayuri_engine_search_new(void* game, void* net, const char* options) {
    EngineOptions opt;
    opt.Parse(options ? options : "");
    return new Search(*static_cast<GameState*>(game), *static_cast<Network*>(net), opt.search);
}
void sayuri_engine_search_free(void* s) { delete static_cast<Search*>(s); }
void sayuri_engine_search_seed(void* s, std::uint64_t caller_seed, std::uint64_t playout_seed) {
    static_cast<Search*>(s)->Seed(caller_seed, playout_seed);
}
void sayuri_engine_search_computation(void* s, void* game, int playouts, int tag, int* ints, float* floats, int* visits,
                                      float* estq, float* target, float* own) {
    auto r = static_cast<Search*>(s)->Computation(playouts, tag);
    Export(*static_cast<GameState*>(game), r, ints, floats, visits, estq, target, own);
}
int sayuri_engine_search_selfplay_move(void* s, void* game, int tag) {
    return MoveToIndex(*static_cast<GameState*>(game), static_cast<Search*>(s)->GetSelfPlayMove(tag));
}
int sayuri_engine_search_think(void* s, void* game) {
    return MoveToIndex(*static_cast<GameState*>(game), static_cast<Search*>(s)->ThinkBestMove());
}
// indices of the buffered training samples that came from single-candidate searches (see search.h); returns the count
// (call it BEFORE sayuri_engine_search_gather, which empties the list; cap = 0 queries the count without copying)
int sayuri_engine_search_single_candidate(void* s, int* out, int cap) {
    const auto& v = static_cast<Search*>(s)->single_candidate_records();
    for (size_t i = 0; i < v.size() && static_cast<int>(i) < cap; ++i) out[i] = v[i];
    return static_cast<int>(v.size());
}
void sayuri_engine_search_update_territory_helper(void* s) { static_cast<Search*>(s)->UpdateTerritoryHelper(); }
// Self-play on a forward pipe (raw = sayuri_pipe_raw(handle); NULL = the dummy random-output backend).
// stats[10] = games_started, games_done, moves, playouts, nn_queries, cache_lookups, cache_hits, records,
// chunks_saved, 0.  Returns 0, or -1 with the message in sayuri_engine_last_error().
static thread_local std::string g_engine_err;
const char* sayuri_engine_last_error() { return g_engine_err.c_str(); }
namespace {
constexpr int kStatSlots = 20;
void PackStats(const SelfplayStats& st, std::uint64_t* v) {  // the 20 slots of sayuri_selfplay_run_ex
    const std::uint64_t a[kStatSlots] = {st.games_started, st.games_done, st.moves, st.playouts, st.nn_queries,
                                         st.cache_lookups, st.cache_hits, st.records, st.chunks_saved, 0,
                                         st.finished_moves, st.prerolled_moves,
                                         st.chunks_saved_window, st.writer_cpu_ns, st.writer_cpu_ns_window, st.bytes_written,
                                         st.text_bytes, st.flush_ns, 0, 0};
    std::memcpy(v, a, sizeof(a));
}
struct StatsHook {
    sayuri_selfplay_stats_fn fn;
    void* user;
};
int StatsTrampoline(const SelfplayStats* st, int local_halt, void* user) {
    const StatsHook* h = static_cast<const StatsHook*>(user);
    std::uint64_t v[kStatSlots];
    PackStats(*st, v);
    return h->fn(v, st->elapsed, local_halt, h->user);
}
} // namespace

int sayuri_selfplay_run_ex(void* raw_pipe, int weights_version, const char* options, const char* name_suffix, double seconds,
                           int move_cap, sayuri_selfplay_stats_fn on_stats, void* user, double interval_seconds,
                           std::uint64_t* stats, double* elapsed) {
    try {
        EngineOptions opt;
        opt.Parse(options ? options : "");
        std::shared_ptr<NetworkForwardPipe> pipe;
        if (raw_pipe) pipe = std::shared_ptr<NetworkForwardPipe>(static_cast<NetworkForwardPipe*>(raw_pipe), [](NetworkForwardPipe*) {});
        SelfplayPipe sp(pipe, weights_version, opt, name_suffix ? name_suffix : "");
        sp.engine().move_cap = move_cap;
        StatsHook hook{on_stats, user};
        if (on_stats) sp.SetStatsCallback(&StatsTrampoline, &hook, interval_seconds > 0 ? interval_seconds : 2.0);
        const SelfplayStats st = sp.Run(seconds);
        PackStats(st, stats);
        stats[9] = static_cast<std::uint64_t>(sp.max_games());
        *elapsed = st.elapsed;
        return 0;
    } catch (const std::exception& e) {
        g_engine_err = e.what();
        return -1;
    }
}

int sayuri_selfplay_run(void* raw_pipe, int weights_version, const char* options, const char* name_suffix, double seconds,
                        int move_cap, std::uint64_t* stats, double* elapsed) {
    std::uint64_t v[kStatSlots];
    const int rc = sayuri_selfplay_run_ex(raw_pipe, weights_version, options, name_suffix, seconds, move_cap, nullptr, nullptr, 0.0, v, elapsed);
    if (rc == 0) std::memcpy(stats, v, sizeof(std::uint64_t) * 10);
    return rc;
}

int sayuri_engine_benchmark(void* raw_pipe, int weights_version, const char* options, int positions, int concurrent, double* out8) {
    try {
        EngineOptions opt;
        opt.Parse(options ? options : "");
        std::shared_ptr<NetworkForwardPipe> pipe;
        if (raw_pipe) pipe = std::shared_ptr<NetworkForwardPipe>(static_cast<NetworkForwardPipe*>(raw_pipe), [](NetworkForwardPipe*) {});
        const SearchBenchmarkResult r = RunSearchBenchmark(pipe, weights_version, opt, positions, concurrent);
        const double v[8] = {r.playouts_per_move, r.playouts_per_second_per_search, r.playouts_per_second_total, r.nn_evals_per_second,
                             r.wall_seconds, r.elo, static_cast<double>(r.nn_queries), static_cast<double>(r.positions)};
        std::memcpy(out8, v, sizeof(v));
        return 0;
    } catch (const std::exception& e) {
        g_engine_err = e.what();
        return -1;
    }
}

// The network evaluation facade on a forward pipe (for GPU parity tests of search moves).
void* sayuri_engine_net_new_pipe(void* raw_pipe, int weights_version, const char* options) {
    EngineOptions opt;
    opt.Parse(options ? options : "");
    auto* net = new Network();
    std::shared_ptr<NetworkForwardPipe> pipe(static_cast<NetworkForwardPipe*>(raw_pipe), [](NetworkForwardPipe*) {});
    net->Initialize(pipe, weights_version, opt.network);
    return net;
}

// Returns the size of the serialized records.  Gathering empties the search's training buffer, so the text is parked
// per calling thread until a call with a large enough `cap` has copied it out: a short buffer loses nothing, the
// caller retries with the returned size (a call for a different search handle drops a parked text).
long sayuri_engine_search_gather(void* s, char* buf, long cap) {
    static thread_local void* parked_for = nullptr;
    static thread_local std::string parked;
    if (parked_for != s) {
        std::vector<TrainingData> chunk;
        static_cast<Search*>(s)->GatherTrainingBuffer(chunk);
        std::ostringstream oss;
        for (auto& d : chunk) d.StreamOut(oss);
        parked = oss.str();
        parked_for = s;
    }
    const long size = static_cast<long>(parked.size());
    if (size <= cap && (buf || size == 0)) {
        if (size > 0) std::memcpy(buf, parked.data(), parked.size());
        parked.clear();
        parked.shrink_to_fit();
        parked_for = nullptr;
    }
    return size;
}

} // extern "C"
