// Self-play: game set-up (board / komi / handicap / random-opening queries), the game loop, SGF and
// training-chunk output, and the worker pool that keeps many games in flight against one evaluation queue.
//
// Behaviour follows the reference's self-play mode (src/selfplay/engine.{h,cc}, src/selfplay/pipe.{h,cc},
// src/game/sgf.cc:513-588, src/neural/training_data.cc): same query grammar ("bkp:19:7.5:0.2", "bhp:9:2:0.1",
// "srs:area:territory"), same set-up randomisation, same 53-line records, same tdata/ vdata/ sgf/ net_queries/
// directory layout (chunks gzip'ed with zlib).  Each game owns its two random streams, so a seed fixes a game.
#pragma once

#include <atomic>
#include <cstdint>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "engine_options.h"
#include "search.h"

namespace sayuri_engine {

struct SelfplayStats {
    std::uint64_t games_started{0}, games_done{0}, moves{0}, playouts{0};
    std::uint64_t nn_queries{0}, cache_lookups{0}, cache_hits{0}, records{0}, chunks_saved{0};
    std::uint64_t finished_moves{0};   // sum of the move numbers of the finished games (their full length)
    std::uint64_t prerolled_moves{0};  // policy-sampled moves played by the stagger_moves option (not searched, not recorded)
    // the data writer (pipe.cc:116-159,181-233).  `*_window` = as of the moment the time window ended (or the last game
    // finished); the others include the flush of the writer's pool behind it
    std::uint64_t chunks_saved_window{0};
    std::uint64_t writer_cpu_ns{0}, writer_cpu_ns_window{0};  // CPU time of the writer thread (CLOCK_THREAD_CPUTIME_ID)
    std::uint64_t bytes_written{0};    // on disk: the gzip'ed tdata / vdata chunks, the SGF and net-queries lines
    std::uint64_t text_bytes{0};       // the records' text before gzip
    std::uint64_t flush_ns{0};         // wall time from the workers' end until the writer has emptied its pool
    double elapsed{0};
};

class SelfplayEngine { // engine.h:12-69
public:
    SelfplayEngine(std::shared_ptr<NetworkForwardPipe> pipe, int weights_version, const EngineOptions& opt);
    void PrepareGame(int g);                                       // engine.cc:193-232
    void PlayPolicyMoves(int g, int count);                        // extension: `count` policy-sampled moves, no search, no records
    std::string SelectWeights() const;                             // engine.cc:63-86
    bool ShouldHalt() const;                                       // engine.cc:88-90
    bool Step(int g);                                              // one self-play move; false once the game is over
    void Selfplay(int g);                                          // engine.cc:234-241
    void GatherTrainingData(std::vector<TrainingData>& chunk, int g);
    std::string GatherSgfString(int g);                            // engine.cc:181-186 + sgf.cc:513-588
    int GetParallelGames() const { return static_cast<int>(slots_.size()); }
    Network& network() { return network_; }
    GameState& state(int g) { return slots_[static_cast<size_t>(g)]->state; }
    Search& search(int g) { return *slots_[static_cast<size_t>(g)]->search; }
    std::uint64_t moves_played() const { return moves_.load(std::memory_order_relaxed); }
    std::uint64_t playouts() const;
    int move_cap{0}; // extension: 0 = none; otherwise both sides pass once a game reaches this many moves

private:
    struct BoardQuery { int board_size; float komi; float prob; };
    struct HandicapQuery { int board_size; int handicaps; float prob; };
    struct Slot {
        GameState state;
        std::unique_ptr<Search> search;
        std::vector<std::string> comments; // SGF comment per move number
    };
    void ParseQueries();
    void SetNormalGame(int g);
    void SetHandicapGame(int g, int handicaps);
    void SetRandomOpeningGame(int g);
    void SetUnfairKomi(int g);
    void SetFairKomi(int g);
    int GetHandicaps(int g);
    Slot& At(int g);

    EngineOptions opt_;
    Network network_;
    std::vector<std::unique_ptr<Slot>> slots_;
    std::vector<BoardQuery> board_queries_;
    std::vector<HandicapQuery> handicap_queries_;
    std::vector<int> scoring_set_;
    std::atomic<std::uint64_t> moves_{0};
};

// The worker pool + data writer (pipe.cc:13-341).  One worker per concurrent game; each blocks in the
// evaluation queue of the forward pipe while its leaf is in a batch.
class SelfplayPipe {
public:
    SelfplayPipe(std::shared_ptr<NetworkForwardPipe> pipe, int weights_version, const EngineOptions& opt,
                 const std::string& name_suffix = "");
    // Plays until `num_games` games are complete, or (seconds > 0) until the clock runs out: games still in
    // progress are then abandoned and not written.
    SelfplayStats Run(double seconds = 0);
    // Periodic exchange hook: every `interval` seconds the calling thread of Run() hands a snapshot of the counters and
    // this process's own halt wish (ShouldHalt: newer weights have appeared) to `cb`; a non-zero return = "some
    // process wants to halt", and the loop winds down as pipe.cc:246-258 does (max_games rounded up to the next 25).
    // The multi-GPU driver all-gathers the records there (sayuri_amd/shard.py) -- the path's only exchange.
    using StatsCallback = int (*)(const SelfplayStats* stats, int local_halt, void* user);
    void SetStatsCallback(StatsCallback cb, void* user, double interval) { stats_cb_ = cb; stats_user_ = user; stats_interval_ = interval; }
    int max_games() const { return max_games_.load(); }
    const std::string& filename_hash() const { return hash_; }
    SelfplayEngine& engine() { return engine_; }

private:
    using DataSgf = std::pair<std::vector<TrainingData>, std::string>;
    void WriterLoop();
    bool SaveChunk(int id, float vdata_prob, std::vector<TrainingData>& chunk, Rng& rng);
    bool WriteGzip(const std::string& name, const std::string& text);
    void SaveSgf(const std::string& sgf);
    void SaveNetQueries(int games, const std::string& text);

    EngineOptions opt_;
    SelfplayEngine engine_;
    std::string hash_, tdata_dir_, vdata_dir_, sgf_dir_, queries_dir_;
    std::mutex data_mu_;
    std::deque<std::shared_ptr<DataSgf>> data_queue_;
    std::deque<std::pair<int, std::string>> queries_queue_;
    std::atomic<bool> writer_running_{false};
    std::atomic<int> accumulation_games_{0}, played_games_{0};
    std::atomic<bool> stop_{false};
    std::atomic<std::uint64_t> records_{0}, chunks_{0}, finished_moves_{0}, prerolled_moves_{0};
    std::atomic<std::uint64_t> writer_cpu_ns_{0}, bytes_written_{0}, text_bytes_{0};
    bool window_closed_{false};              // under data_mu_: the time window has ended, finished games are no longer taken
    std::uint64_t flush_helper_cpu_ns_{0};   // CPU time of the threads that share the final flush of the writer's pool
    std::atomic<int> max_games_{0};
    std::atomic<bool> halt_wish_{false};     // set by worker 0 (ShouldHalt) or by the stats callback's verdict
    void WindDown();                         // pipe.cc:248-254
    StatsCallback stats_cb_{nullptr};
    void* stats_user_{nullptr};
    double stats_interval_{2.0};
    std::string error_;
};

float AdjustKomiToHalf(float komi); // utils/komi.cc AdjustKomi<float>

// benchmark.cc -- the reference's --mode benchmark (src/benchmark/benchmark.cc:78-161)
struct SearchBenchmarkResult {
    int positions{0}, concurrent{0};
    double playouts_per_move{0};               // average playouts of a timed search
    double playouts_per_second_per_search{0};  // average playouts / average elapsed: the reference's "p/s" figure
    double playouts_per_second_total{0};       // all playouts / wall clock (what `concurrent` searches deliver together)
    double nn_evals_per_second{0};
    double wall_seconds{0}, elo{0};
    std::uint64_t nn_queries{0};
};
SearchBenchmarkResult RunSearchBenchmark(std::shared_ptr<NetworkForwardPipe> pipe, int weights_version, const EngineOptions& opt,
                                         int positions, int concurrent);
double ComputeEloEffect(double playouts_per_move, double playouts_per_second, int threads);

} // namespace sayuri_engine
