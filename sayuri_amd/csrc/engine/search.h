// Search -- one game's Monte-Carlo tree search: playouts from the root state, root statistics, the
// self-play move chooser and the training records it emits.
//
// Follows the reference's `class Search` (src/mcts/search.h:155-296, src/mcts/search.cc) for everything a
// self-play game or a genmove reads: visit/playout caps, sub-tree reuse, territory-rule handling in playouts,
// policy-target pruning, completed-Q targets, resign / pass policy, KataGo-style value targets.  Out of scope
// here (GTP front-end features): time control, pondering, analysis streams, opening book.
//
// Randomness: the reference draws from two thread-local generators -- the calling thread's (root expansion,
// Dirichlet noise, move pickers) and the search worker's (symmetry of each playout's evaluation).  The same
// split is kept as two explicit streams so that a fixed-seed search reproduces the reference's moves.
#pragma once

#include <atomic>
#include <cstdint>
#include <memory>
#include <ostream>
#include <string>
#include <vector>

#include "tree.h"

namespace sayuri_engine {

// One training sample (reference src/neural/training_data.h:6-66); StreamOut writes the 53-line v2 record.
struct TrainingData {
    int version{2}, mode{0}, board_size{0};
    float komi{0};
    int side_to_move{0};
    // the 37 binary input planes as bit planes, [plane][12 words], bit y*bs+x (what the record writes, 4 cells per hex digit):
    // 1.8 KB per move instead of 62 KB of fp32 planes -- a game holds its ~375 samples until it ends, which was 12 GB of the
    // resident set of a 512-game rank (and as much again of malloc arenas that do not shrink)
    std::vector<std::uint32_t> plane_bits;
    std::vector<float> probabilities, auxiliary_probabilities;
    std::vector<int> ownership;
    int result{0};
    float q_value{0}, avg_q_value{0}, short_avg_q{0}, middle_avg_q{0}, long_avg_q{0};
    float final_score{0}, score_lead{0}, avg_score_lead{0}, short_avg_score{0}, middle_avg_score{0}, long_avg_score{0};
    float q_stddev{0}, score_stddev{0}, kld{0}, rule{0}, wave{0};
    int accum_resign_cnt{0};
    bool discard{false};
    void StreamOut(std::ostream& out) const; // training_data.cc:63-95
};

struct ComputationResult { // search.h:111-153
    int board_size{0};
    int best_move{sayuri_go::kNoVertex}, best_no_pass_move{sayuri_go::kNoVertex}, random_move{sayuri_go::kNoVertex};
    int gumbel_move{sayuri_go::kNoVertex}, gumbel_no_pass_move{sayuri_go::kNoVertex};
    int capture_all_dead_move{sayuri_go::kNoVertex}, high_priority_move{sayuri_go::kNoVertex};
    int to_move{sayuri_go::kBlack};
    float komi{0}, root_eval{0}, root_score_lead{0}, best_eval{0}, root_score_stddev{0}, root_eval_stddev{0};
    std::vector<float> root_ownership;
    std::vector<int> root_searched_visits;
    std::vector<float> root_estimated_q, root_visits_dist, target_policy_dist;
    std::vector<std::vector<int>> alive_strings, dead_strings;
    int movenum{0}, visits{0}, playouts{0};
    float policy_kld{0};
    bool side_resign{false};
};

class Search {
public:
    enum OptionTag : int {
        kNullTag = 0,
        kThinking = 1 << 1,
        kForced = 1 << 4,      // strip trailing double passes before searching
        kUnreused = 1 << 5,    // visit cap instead of playout cap
        kNoExploring = 1 << 6, // no noise / temperature / forced playouts
        kNoBuffer = 1 << 7,    // do not record a training sample
    };

    Search(GameState& state, Network& network, const SearchParams& params);

    // Re-seed the two random streams (caller stream, playout stream).
    void Seed(std::uint64_t caller_seed, std::uint64_t playout_seed);
    Rng& caller_rng() { return caller_rng_; }

    ComputationResult Computation(int playouts, int tag); // search.cc:257-430
    int GetBestMove(int playouts, int tag);               // search.cc:847-881
    int ThinkBestMove();
    int GetSelfPlayMove(int tag = kNullTag);              // search.cc:949-1068
    void UpdateTerritoryHelper();                         // search.cc:1148-1173
    void GatherTrainingBuffer(std::vector<TrainingData>& chunk); // search.cc:1180-1306
    void ClearTrainingBuffer() { training_buffer_.clear(); }
    void ReleaseTree() { root_.reset(); }
    // When set and raised, a running Computation stops after the current playout (used to end a timed self-play run).
    void SetAbortFlag(const std::atomic<bool>* flag) { abort_ = flag; }
    SearchParams* GetParams(bool no_exploring = false) { return no_exploring ? passive_ : active_; }
    const Node* root() const { return root_.get(); }
    const std::string& last_comment() const { return last_comment_; }
    size_t total_playouts() const { return total_playouts_.load(std::memory_order_relaxed); }
    // searches that ended at once because the root had a single candidate (the reference stops those from a
    // polling thread, after a timing-dependent handful of playouts: search.cc:352-386, 1423-1441)
    int single_candidate_searches() const { return single_candidate_searches_; }
    // indices (within the current training buffer, counting discarded samples) of the samples such searches produced
    const std::vector<int>& single_candidate_records() const { return single_candidate_records_; }

private:
    struct PlayoutResult {
        bool valid{false};
        NodeEvals evals{};
    };
    void PlaySimulation(GameState& state, Node* node, int depth, PlayoutResult& result); // search.cc:60-137
    void GameOverEvals(GameState& state, PlayoutResult& result);                        // search.h:33-60
    void TryRecoverOwnershipMap(GameState& state, std::vector<int>& ownership);          // search.h:62-105
    void PrepareRootNode(ComputationResult& result, int tag);                            // search.cc:139-181
    void PrepareParam();
    bool AdvanceToNewRootState(int tag);                                                 // search.cc:1342-1417
    int GetPlayoutsLeft(int cap, int tag) const;
    bool AchieveCap(int cap, int tag) const { return GetPlayoutsLeft(cap, tag) == 0; }
    bool HaveAlternateMoves() const;
    bool StoppedByKldGain(ComputationResult& result, int tag);
    void UpdateComputationResult(ComputationResult& result); // search.cc:467-745
    void GatherData(const GameState& state, ComputationResult& result, bool discard);

    GameState& root_state_;
    GameState last_state_;
    Network& network_;
public:
    std::size_t arena_bytes() const { return arena_.slab_bytes(); }  // memory statistics (SAYURI_MEMSTAT)
    std::size_t arena_live_blocks() const { return arena_.live_blocks(); }
private:
    static constexpr std::size_t kArenaKeepBytes = std::size_t(8) << 20;  // slabs a game keeps across fresh roots
    TreeArena arena_;             // before root_: the tree is destroyed first, its blocks go back into a living arena
    std::unique_ptr<Node> root_;
    NodeEvals root_evals_;
    SearchParams params_[2]; // [0] normal, [1] exploration disabled
    SearchParams *active_, *passive_;
    TQuantiles t_quantiles_;
    SearchShared shared_;
    Rng caller_rng_, playout_rng_;
    std::vector<TrainingData> training_buffer_;
    std::vector<float> root_raw_probabilities_;
    std::vector<double> prev_kld_policy_;
    int prev_kld_visits_{0};
    int playouts_{0};
    std::atomic<size_t> total_playouts_{0};  // read by the self-play statistics while the game thread searches
    int single_candidate_searches_{0};
    bool last_single_candidate_{false};
    std::vector<int> single_candidate_records_;
    std::string last_comment_;
    const std::atomic<bool>* abort_{nullptr};
};

bool ShouldResign(GameState& state, ComputationResult& result, const SearchParams* param); // search.cc:749-793
bool ShouldPass(GameState& state, ComputationResult& result, const SearchParams* param);   // search.cc:795-845
bool ShouldForbidPass(GameState& state, ComputationResult& result, NodeEvals& root_evals);  // search.cc:889-947

} // namespace sayuri_engine
