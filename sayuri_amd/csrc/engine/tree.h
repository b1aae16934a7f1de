// Search tree node: expansion from a network evaluation, PUCT / Gumbel child selection, value backup,
// and the root statistics the move chooser and the training targets read.
//
// Restates the arithmetic of the reference's `class Node` (src/mcts/node.h:77-338, src/mcts/node.cc) so a
// fixed-seed search walks the same tree: same float/double mix in every formula, same child order
// (stable sort by policy then vertex), same random draws in the same order.  What differs: a tree belongs to
// one game and is walked by one playout at a time (the engine gets its parallelism from thousands of
// concurrent games, not from threads inside a tree), so there are no atomics, spin-waits or tagged
// pointers here, and children hold their node in a plain unique_ptr that is filled on first descent.
#pragma once

#include <array>
#include <cstdint>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "game_state.h"
#include "network.h"
#include "search_params.h"
#include "tree_arena.h"

namespace sayuri_engine {

struct NodeEvals { // node.h:16-21
    float black_final_score{0.0f};
    float black_wl{0.0f};
    float draw{0.0f};
    std::array<float, sayuri_go::kMaxPoints> black_ownership;
};

struct SearchShared { // what every node of one search can reach
    const TQuantiles* t_quantiles{nullptr};
    TreeArena* arena{nullptr}; // the game's own tree memory (tree_arena.h); null = malloc
};

class Node {
public:
    struct Edge {
        Edge(int v, float p) : vertex(static_cast<std::int16_t>(v)), policy(p) {}
        std::int16_t vertex;
        float policy;
        std::unique_ptr<Node> node; // empty until the edge is first descended ("inflated")
        Node* Get() const { return node.get(); }
        int GetVertex() const { return vertex; }
        float GetPolicy() const { return node ? node->GetPolicy() : policy; }
        int GetVisits() const { return node ? node->GetVisits() : 0; }
    };

    using EdgeList = std::vector<Edge, TreeArenaAllocator<Edge>>;

    Node(SearchParams* param, const SearchShared* shared, int vertex, float policy)
        : param_(param), shared_(shared), children_(TreeArenaAllocator<Edge>(shared ? shared->arena : nullptr)), policy_(policy),
          vertex_(static_cast<std::int16_t>(vertex)) {}
    // nodes come out of the game's tree arena: `new (shared) Node(...)`; a plain `new Node` uses malloc behind the same
    // header, so `delete` (the unique_ptrs of the edges) is one function for both
    static void* operator new(std::size_t n, const SearchShared* shared) { return TreeArena::Alloc(shared ? shared->arena : nullptr, n); }
    static void* operator new(std::size_t n) { return TreeArena::Alloc(nullptr, n); }
    static void operator delete(void* p) noexcept { TreeArena::Release(p); }
    static void operator delete(void* p, const SearchShared*) noexcept { TreeArena::Release(p); }

    // node.cc:127-357
    bool ExpandChildren(Network& network, GameState& state, NodeEvals& evals, bool is_root, Rng& rng);
    bool PrepareRootNode(Network& network, GameState& state, NodeEvals& evals, Rng& rng); // node.cc:30-66
    bool SetTerminal(const NodeEvals* evals);

    Node* DescentSelectChild(int color, bool is_root, Rng& rng); // node.cc:373-383
    Node* ProbSelectChild(bool allow_pass);                      // node.cc:385-421
    Node* PuctSelectChild(int color, bool is_root);              // node.cc:505-585
    int GetRandomMoveProportionally(float temp, float min_ratio, int min_visits, Rng& rng); // node.cc:587-643
    int GetRandomMoveWithLogitsQ(GameState& state, float temp, Rng& rng);                   // node.cc:645-706
    void Update(const NodeEvals* evals);                              // node.cc:708-751
    void UpdateScoreBonus(GameState& state, NodeEvals& evals);        // node.cc:68-83
    std::vector<std::pair<float, int>> GetSortedLcbUtilityList(int color);
    std::vector<std::pair<float, int>> GetSortedLcbUtilityList(int color, int children_visits); // node.cc:1148-1187
    float GetLcb(int color) const;      // node.cc:793-813
    int GetBestMove(bool allow_pass);   // node.cc:1189-1218
    int GetGumbelMove(bool allow_pass, Rng& rng); // node.cc:1794-1821
    std::vector<float> GetProbLogitsCompletedQ(GameState& state); // node.cc:1487-1505
    bool ShouldApplyGumbel() const;

    EdgeList& GetChildren() { EnsureSorted(); return children_; }
    const EdgeList& GetChildren() const { EnsureSorted(); return children_; }
    bool HasChildren() const { return expanded_ && color_ != sayuri_go::kWall; }
    Node* GetChild(int vertex);
    std::unique_ptr<Node> PopChild(int vertex);

    int GetVisits() const { return visits_; }
    int GetVertex() const { return vertex_; }
    float GetPolicy() const { return policy_; }
    float GetNetWL(int color) const { return color == sayuri_go::kBlack ? black_wl_ : 1.0f - black_wl_; }
    float GetNetScore(int color) const { return color == sayuri_go::kBlack ? black_fs_ : 0.0f - black_fs_; }
    float GetScoreBonus(int color) const { return color == sayuri_go::kBlack ? black_sb_ : 0.f - black_sb_; }
    float GetFinalScore(int color) const;
    float GetScoreEval(int color) const;
    float GetWL(int color, bool use_virtual_loss = false) const;
    float GetDraw() const { return static_cast<float>(acc_draw_ / GetVisits()); }
    float GetFpu(int color, float total_visited_policy, bool is_root) const;
    float GetCpuct(int children_visits) const;
    int GetForcedVisits(float policy, int children_visits, bool is_root) const;
    std::array<float, sayuri_go::kMaxPoints> GetOwnership(int color) const;
    float GetWLStddev() const;
    float GetScoreStddev() const;
    int GetChildrenVisits() const;

    bool Expandable() const { return !expanded_; }
    bool IsActive() const { return status_ == kActive; }
    bool IsValid() const { return status_ != kInvalidNode; }
    void Invalidate() { status_ = kInvalidNode; }
    size_t CountNodes() const;

private:
    enum Status : std::uint8_t { kInvalidNode, kPruned, kActive };

    Network::Result GetNetOutput(Network& network, GameState& state, bool is_root, Rng& rng);
    void RecomputePolicy(Network& network, GameState& state, NodeEvals& evals, bool is_root, Rng& rng);
    void FillNodeEvalsFromNet(const Network::Result& net, NodeEvals& evals, int color) const;
    void ApplyDirichletNoise(float alpha, Rng& rng);
    void KillRootSuperkos(GameState& state);
    void ComputeScoreBonus(GameState& state, NodeEvals& parent_evals);
    float GetDynamicCpuctFactor(Node* node, int visits, int children_visits) const;
    float GetSearchPolicy(const Edge& child, bool is_root) const;
    float GetScoreVariance(float default_var, int visits) const;
    float GetWLVariance(float default_var, int visits) const;
    Node* Inflate(Edge& e);
    void InflateAllChildren();
    void BuildRootIndex();  // search_params.h RootIndex
    float GetGumbelEval(int color) const;
    float TransformCompletedQ(float completed_q, int max_visits) const;
    bool ProcessGumbelLogits(std::vector<float>& logits, int color, bool only_max_visits, Rng& rng);
    Node* GumbelSelectChild(int color, bool only_max_visits, bool allow_pass, Rng& rng);
    void MixLogitsCompletedQ(GameState& state, std::vector<float>& prob);

    SearchParams* param_;
    const SearchShared* shared_;
    EdgeList children_;
    std::array<float, sayuri_go::kMaxPoints> avg_black_ownership_;
    double sq_eval_diff_{static_cast<double>(1e-4f)};
    double sq_score_diff_{static_cast<double>(1e-4f)};
    double acc_black_fs_{0.0};
    double acc_black_wl_{0.0};
    double acc_draw_{0.0};
    float black_sb_{0.f}; // score bonus (first-pass bonus), black's view
    float black_wl_{0.5f};
    float black_fs_{0.0f};
    float policy_;
    int visits_{0};
    std::int16_t vertex_;
    std::int16_t inflated_hi_{0};  // every child at an index >= this is still a bare edge (Inflate keeps it; selection stops there)
    // The children's order (best policy first, ties: higher vertex first) is established LAZILY: an expansion puts the
    // kSortedAtExpansion best in place and leaves the others behind them in no particular order -- a node has ~360 children, a
    // search descends into a handful of them, and sorting all 360 keys was the largest single item of a self-play rank's host
    // time.  children_[0 .. sorted_n_) are final; EnsureSorted(k) finishes the order when anything wants to look at index >= sorted_n_
    // (the selection asks for inflated_hi_ + 1, everything else for all).  Bare edges only are ever moved (an inflated edge
    // sits below inflated_hi_ <= sorted_n_), and the order reached is the same total order whichever way it is reached.
    static constexpr int kSortedAtExpansion = 16;
    mutable std::int16_t sorted_n_{0};
    void EnsureSorted(int upto = 1 << 14) const {
        if (upto > sorted_n_ && sorted_n_ < static_cast<int>(children_.size())) SortTail();
    }
    void SortTail() const;
    std::uint8_t color_{sayuri_go::kWall}; // side to move here; kWall while there are no children
    Status status_{kActive};
    bool expanded_{false};
};

} // namespace sayuri_engine
