// oracle/ref_search_driver.cc -- TEST INFRASTRUCTURE ONLY (never part of the product path).
//
// extern "C" taps on the *unmodified* reference Network facade (src/neural/network.h), Search
// (src/mcts/search.h) and training-record writer, compiled in THIS container only into
// oracle/_ref/libsayuri_ref.so.  The reference seeds its generators from thread ids
// (utils/random.cc:21-27); ref_seed() pins them: every Random<> instance of a thread shares one
// thread_local state (utils/random.h:57), so constructing a temporary generator with a seed re-seeds
// the calling thread's stream, and the same is done on the single pool worker that runs playouts.
#include <cstdint>
#include <cstring>
#include <sstream>
#include <string>
#include <vector>

#include "config.h"
#include "game/game_state.h"
#include "mcts/search.h"
#include "neural/network.h"
#include "neural/training_data.h"
#include "utils/option.h"
#include "utils/random.h"
#include "utils/threadpool.h"

extern "C" int ref_ensure_args(int winograd);

namespace {
int MoveToIndex(const GameState& g, int v) {
    if (v == kNullVertex) return -3;
    if (v == kPass) return g.GetNumIntersections();
    if (v == kResign) return -1;
    return g.VertexToIndex(v);
}
} // namespace

extern "C" {

int ref_opt_int(const char* k, int v) { ref_ensure_args(1); return SetOption<int>(k, v) ? 0 : -1; }
int ref_opt_float(const char* k, float v) { ref_ensure_args(1); return SetOption<float>(k, v) ? 0 : -1; }
int ref_opt_double(const char* k, double v) { ref_ensure_args(1); return SetOption<double>(k, v) ? 0 : -1; }
int ref_opt_bool(const char* k, int v) { ref_ensure_args(1); return SetOption<bool>(k, v != 0) ? 0 : -1; }

void ref_seed(std::uint64_t caller_seed, std::uint64_t worker_seed) {
    // touch the lazily-built thread_local generator first: its constructor seeds from the thread id and would
    // otherwise overwrite the pinned state at its first use
    Random<>::Get();
    { Random<kXoroShiro128Plus> r(caller_seed); }
    ThreadPool::Get().AddTask([worker_seed]() {
        Random<>::Get();
        Random<kXoroShiro128Plus> r(worker_seed);
    }).get();
}

// weights == "" -> the dummy (random outputs) backend
void* ref_net_new(const char* weights) {
    ref_ensure_args(1);
    auto* n = new Network();
    n->Initialize(weights);
    return n;
}
void ref_net_free(void* n) {
    static_cast<Network*>(n)->Destroy();
    delete static_cast<Network*>(n);
}
unsigned long ref_net_queries(void* n) { return static_cast<Network*>(n)->GetNumQueries(); }

// out: prob[N], own[N], pass, wdl[3], wdl_winrate, stm_winrate, final_score, q_error, score_error
void ref_net_output(void* n, void* game, int ensemble, int symmetry, float temperature, int use_cache, float* out) {
    auto* g = static_cast<GameState*>(game);
    auto q = Network::Query::Get().SetTemperature(temperature).SetSymmetry(symmetry).SetCache(use_cache != 0);
    auto r = static_cast<Network*>(n)->GetOutput(*g, static_cast<Network::Ensemble>(ensemble), q);
    const int N = g->GetNumIntersections();
    std::memcpy(out, r.probabilities.data(), sizeof(float) * N);
    std::memcpy(out + N, r.ownership.data(), sizeof(float) * N);
    float* s = out + 2 * N;
    s[0] = r.pass_probability;
    s[1] = r.wdl[0]; s[2] = r.wdl[1]; s[3] = r.wdl[2];
    s[4] = r.wdl_winrate; s[5] = r.stm_winrate; s[6] = r.final_score; s[7] = r.q_error; s[8] = r.score_error;
}

void* ref_search_new(void* game, void* net) {
    ref_ensure_args(1);
    return new Search(*static_cast<GameState*>(game), *static_cast<Network*>(net));
}
void ref_search_free(void* s) { delete static_cast<Search*>(s); }

// ints: best, best_no_pass, random, gumbel, gumbel_no_pass, capture_all_dead, high_priority, visits, playouts,
//       to_move, side_resign;  floats: root_eval, root_score_lead, best_eval, root_score_stddev, root_eval_stddev,
//       policy_kld;  arrays: visits[N+1], estimated_q[N+1], target[N+1], ownership[N]
static void Export(const GameState& g, const ComputationResult& r, int* ints, float* floats, int* visits, float* estq,
                   float* target, float* own) {
    const int N = g.GetNumIntersections();
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
    floats[0] = r.root_eval; floats[1] = r.root_score_lead; floats[2] = r.best_eval;
    floats[3] = r.root_score_stddev; floats[4] = r.root_eval_stddev; floats[5] = r.policy_kld;
    if (static_cast<int>(r.root_searched_visits.size()) == N + 1) {
        for (int i = 0; i <= N; ++i) {
            visits[i] = r.root_searched_visits[i];
            estq[i] = r.root_estimated_q[i];
            target[i] = r.target_policy_dist[i];
        }
        for (int i = 0; i < N; ++i) own[i] = r.root_ownership[i];
    }
}

void ref_search_computation(void* s, void* game, int playouts, int tag, int* ints, float* floats, int* visits,
                            float* estq, float* target, float* own) {
    auto r = static_cast<Search*>(s)->Computation(playouts, static_cast<Search::OptionTag>(tag));
    Export(*static_cast<GameState*>(game), r, ints, floats, visits, estq, target, own);
}
int ref_search_selfplay_move(void* s, void* game, int tag) {
    auto* g = static_cast<GameState*>(game);
    return MoveToIndex(*g, static_cast<Search*>(s)->GetSelfPlayMove(static_cast<Search::OptionTag>(tag)));
}
int ref_search_think(void* s, void* game) {
    auto* g = static_cast<GameState*>(game);
    return MoveToIndex(*g, static_cast<Search*>(s)->ThinkBestMove());
}
void ref_search_update_territory_helper(void* s) { static_cast<Search*>(s)->UpdateTerritoryHelper(); }
// The game's training records as the reference writes them (training_data.cc:63-95); returns the byte count.
long ref_search_gather(void* s, char* buf, long cap) {
    std::vector<TrainingData> chunk;
    static_cast<Search*>(s)->GatherTrainingBuffer(chunk);
    std::ostringstream oss;
    for (auto& d : chunk) d.StreamOut(oss);
    const std::string str = oss.str();
    if (static_cast<long>(str.size()) <= cap) std::memcpy(buf, str.data(), str.size());
    return static_cast<long>(str.size());
}

} // extern "C"
