// oracle/ref_hip_callers.cc -- TEST INFRASTRUCTURE ONLY (never part of the product path).
//
// The drop-in under the REFERENCE'S OWN CALLERS.  Compiled, with the unmodified reference
// sources, only into oracle/_ref/libsayuri_ref_hip.so and libsayuri_ref_hip_sc.so (the same with -DSELF_CHECK: the
// reference's Network then evaluates every position on its BlasForwardPipe too and throws on an L2 above 0.2,
// src/neural/network.cc:188-193, 333-359).  What runs here is the reference's code --
//   Network::GetOutput / GetOutputInternal (network.cc:167-291): encoder, symmetry, cache, pipe_->Forward(inputs)
//   Search::Computation with `threads` playout threads (mcts/search.cc:252-436), ThreadPool / ThreadGroup
// -- over THIS repo's HipForwardPipe (csrc/host/hip_forward_pipe.cc compiled against the reference's headers), i.e.
// the plugin interface's contract "blocking, re-entrant, called concurrently from every search thread" 
// exercised from the reference side.
//
// The backend is chosen at compile time in the reference (network.cc:61-67, `using Backend = ...`); INTEGRATION.md shows
// the three-line USE_HIP patch.  This driver does not patch or shadow anything: it lets Network::Initialize build its
// default pipe and then puts a HipForwardPipe, initialised with the SAME DNNWeights, into Network::pipe_ (a private
// member, reached through an explicit template instantiation, which may name private members).
#include <atomic>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "config.h"
#include "game/game_state.h"
#include "mcts/search.h"
#include "neural/network.h"
#include "utils/option.h"
#include "utils/threadpool.h"
#include "utils/time.h"

#include "hip_forward_pipe.h"

extern "C" int ref_ensure_args(int winograd);

namespace {
template <typename Tag, typename Tag::type M> struct Rob {
    friend typename Tag::type get(Tag) { return M; }
};
struct NetPipe {
    typedef std::unique_ptr<NetworkForwardPipe> Network::*type;
    friend type get(NetPipe);
};
template struct Rob<NetPipe, &Network::pipe_>;

std::string g_err2;
HipForwardPipe* g_hip_raw = nullptr;  // owned by the Network it was put into
}  // namespace

extern "C" {

const char* ref_hip_callers_error() { return g_err2.c_str(); }

// A reference Network on `weights` whose forward pipe is a HipForwardPipe (board / batch / fp16 / device as given).
void* ref_hip_net_new(const char* weights, int board, int batch, int fp16, int device, int cache_mib) {
    try {
        ref_ensure_args(1);
        SetOption("defualt_boardsize", board);
        SetOption("batch_size", batch);
        SetOption("cache_memory_mib", cache_mib);
        auto* n = new Network();
        n->Initialize(weights);  // loader + the default (CPU) pipe; with -DSELF_CHECK also cpu_pipe_
        auto& slot = (*n).*get(NetPipe());
        if (!slot || !slot->Valid()) { g_err2 = "the reference loader rejected the weights file"; delete n; return nullptr; }
        std::shared_ptr<DNNWeights> w = slot->weights_;
        HipPipeConfig cfg;
        cfg.batch_size = batch;
        cfg.fp16 = fp16 != 0;
        cfg.default_boardsize = board;
        if (device >= 0) cfg.gpus = {device};
        auto hip = std::make_unique<HipForwardPipe>(cfg);
        hip->Initialize(w);
        g_hip_raw = hip.get();
        slot->Destroy();
        slot = std::move(hip);
        return n;
    } catch (const std::exception& e) {
        g_err2 = e.what();
        return nullptr;
    }
}
void ref_hip_net_free(void* n) {
    if (!n) return;
    static_cast<Network*>(n)->Destroy();
    delete static_cast<Network*>(n);
    g_hip_raw = nullptr;
}

// The reference's `netbench` loop (GtpLoop::NetBench, game/gtp.cc:1516-1557; a private member of the GTP front-end, so
// its body is restated): `threads` workers of the reference's ThreadPool call Network::GetOutput(state, kRandom,
// cache off) until the time limit.  out = {evals, seconds, evals/s, batches, mean batch}
int ref_hip_netbench(void* net, int board, int threads, float timelimit, double* out) {
    try {
        auto* n = static_cast<Network*>(net);
        GameState state;
        state.Reset(board, 7.5f, kArea);
        n->ResetNumQueries();
        const size_t b0 = g_hip_raw ? g_hip_raw->num_batches() : 0, e0 = g_hip_raw ? g_hip_raw->num_evals() : 0;
        std::atomic<bool> running{true};
        std::string err;
        std::mutex mu;
        const auto Worker = [&]() -> void {
            try {
                while (running.load(std::memory_order_relaxed))
                    n->GetOutput(state, Network::kRandom, Network::Query::Get().SetCache(false));
            } catch (const std::exception& e) {
                std::lock_guard<std::mutex> lk(mu);
                err = e.what();
                running.store(false);
            }
        };
        Timer timer;
        timer.Clock();
        auto group = ThreadGroup<void>(&ThreadPool::Get("search", threads));
        for (int i = 0; i < threads; ++i) group.AddTask(Worker);
        while (timer.GetDuration() < timelimit && running.load()) std::this_thread::yield();
        running.store(false, std::memory_order_relaxed);
        group.WaitToJoin();
        const double el = timer.GetDuration();
        if (!err.empty()) { g_err2 = err; return -1; }
        out[0] = static_cast<double>(n->GetNumQueries());
        out[1] = el;
        out[2] = out[0] / el;
        out[3] = g_hip_raw ? static_cast<double>(g_hip_raw->num_batches() - b0) : 0;
        out[4] = out[3] > 0 ? static_cast<double>(g_hip_raw->num_evals() - e0) / out[3] : 0;
        return 0;
    } catch (const std::exception& e) {
        g_err2 = e.what();
        return -1;
    }
}

// The reference's Search::Computation with `threads` playout threads on the position reached by `moves` (vertex
// indices y*board+x, board*board = pass).  out = {best move index, root visits, playouts, seconds, nn queries}
int ref_hip_search(void* net, int board, float komi, const int* moves, int nmoves, int threads, int playouts, double* out) {
    try {
        auto* n = static_cast<Network*>(net);
        SetOption("threads", threads);
        SetOption("playouts", playouts);
        GameState state;
        state.Reset(board, komi, kArea);
        for (int i = 0; i < nmoves; ++i) {
            const int m = moves[i];
            const int vtx = m == board * board ? kPass : state.GetVertex(m % board, m / board);
            if (!state.PlayMove(vtx)) { g_err2 = "illegal move in the position"; return -1; }
        }
        n->ResetNumQueries();
        Search search(state, *n);
        Timer timer;
        timer.Clock();
        const auto result = search.Computation(playouts, Search::kNullTag);
        out[3] = timer.GetDuration();
        const int v = result.best_move;
        out[0] = v == kPass ? board * board : v == kResign ? -1 : state.VertexToIndex(v);
        out[1] = result.visits;
        out[2] = result.playouts;
        out[4] = static_cast<double>(n->GetNumQueries());
        return 0;
    } catch (const std::exception& e) {
        g_err2 = e.what();
        return -1;
    }
}

}  // extern "C"
