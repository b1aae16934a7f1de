// Search parameters and the two small lookup tables the selection arithmetic needs.
//
// Field names and defaults are the reference's option map (src/config.cc:21-133, read by
// src/mcts/parameters.h:12-78); only the knobs the search / self-play path consumes are kept.
#pragma once

#include <array>
#include <vector>

#include "go_base.h"

namespace sayuri_engine {

struct SearchParams {
    int threads{1};
    int batch_size{1};
    int playouts{400};
    int virtual_loss_count{1};
    int random_min_visits{1};
    float random_min_ratio{0.f};
    float random_moves_factor{0.f};
    float random_moves_temp{1.f};
    float resign_threshold{0.1f};
    float lcb_reduction{0.02f};
    float fpu_reduction{0.25f};
    float root_fpu_reduction{0.25f};
    float cpuct_init{0.5f};
    float cpuct_base_factor{1.0f};
    float cpuct_base{19652.f};
    bool cpuct_dynamic{true};
    float cpuct_dynamic_k_factor{4.f};
    float cpuct_dynamic_k_base{10000.f};
    float forced_playouts_k{0.f};
    float suppress_pass_factor{0.1667f};
    float gumbel_c_visit{50.f};
    float gumbel_c_scale{1.f};
    int gumbel_prom_visits{1};
    int gumbel_considered_moves{16};
    int gumbel_playouts_threshold{400};
    bool gumbel{false};
    bool always_completed_q_policy{false};
    bool dirichlet_noise{false};
    float dirichlet_epsilon{0.25f};
    float dirichlet_factor{361.f};
    float dirichlet_init{0.03f};
    double kldgain_per_node{0.0};
    int kldgain_interval{0};
    float score_utility_factor{0.4f};
    float score_utility_div{1.f};
    float root_policy_temp{1.f};
    float policy_temp{1.f};
    int resign_playouts{0};
    int fastsearch_playouts{0};
    float fastsearch_playouts_prob{0.f};
    float random_fastsearch_prob{0.f};
    float resign_discard_prob{0.f};
    bool reuse_tree{false};
    bool friendly_pass{false};
    bool first_pass_bonus{false};
    bool symm_pruning{false};
    bool use_stm_winrate{false};
    bool capture_all_dead{false};
    float ci_alpha{1e-5f};

    // search-time state the reference also keeps in this struct (parameters.h:74-77)
    bool no_exploring_phase{false};
    int board_size{sayuri_go::kMaxBoard};
    float recent_expected_black_score{0.0f};
    std::array<float, sayuri_go::kMaxVertices + 10> dirichlet_buffer{};

    // Selection at the root (tree.cc Node::PuctSelectChild): all ~360 root children carry nodes, and looking at every one of
    // them for every playout was most of the selection's cost (a cache line per child).  A child that was never chosen has no
    // visits and scores fpu + cpuct * psa * sqrt(N): monotone in its search policy psa, which is fixed for the whole search.  So
    // the root keeps its children's indices by descending psa and the (sorted) list of the children chosen so far; a call looks
    // at those and at the head of the never-chosen ones.  Built by PrepareRootNode, valid while `root_index_owner` is the root.
    // The index is per-search scratch that lives here only because the selection gets at the parameters and not at the Search:
    // a COPY of the parameters starts with an empty index (no stale owner pointer, no copied vectors) and whoever searches
    // with the copy rebuilds it in PrepareRootNode.  One playout per tree at a time (the self-play rule) is what makes the
    // mutation under a const Node safe; several threads in one tree would need the index on the Search.
    struct RootIndex {
        RootIndex() = default;
        RootIndex(const RootIndex&) {}
        RootIndex& operator=(const RootIndex& o) {
            if (this != &o) *this = RootIndex();
            return *this;
        }
        RootIndex(RootIndex&&) = default;
        RootIndex& operator=(RootIndex&&) = default;
        const void* owner{nullptr};
        std::vector<std::int16_t> by_psa;     // child indices, psa descending, ties: lower index first
        std::vector<std::int16_t> chosen;     // child indices that were returned by the selection, ascending
        std::vector<std::uint8_t> is_chosen;  // per child index
        int cursor{0};                        // by_psa[0 .. cursor) are all chosen
    } root_index;
};

// Expected score utility E[2/pi * atan(x / board)] for x ~ N(mean, stddev), from a table integrated once
// (reference src/mcts/score_value.h:33-134, itself after KataGo's nninputs.cpp).
class ScoreUtility {
public:
    static const ScoreUtility& Get();
    float Expected(float mean, float stddev, float center, float scale, float board_size) const;

private:
    ScoreUtility();
    static constexpr int kExtra = 60;
    static constexpr int kMeanRadius = sayuri_go::kMaxPoints + kExtra;
    static constexpr int kMeanLen = kMeanRadius * 2;
    static constexpr int kStddevLen = sayuri_go::kMaxPoints + kExtra;
    std::vector<float> table_;
};

// Student-t quantiles for the lower confidence bound (reference src/mcts/lcb.h:11-87).
class TQuantiles {
public:
    explicit TQuantiles(float complement_probability);
    float At(int degrees) const;

private:
    std::array<float, 1000> z_;
};

} // namespace sayuri_engine
