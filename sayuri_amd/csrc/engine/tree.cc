#include "tree.h"

#include <cstdio>
#include <cstdlib>

#include <algorithm>
#include <functional>
#include <cstring>
#include <cmath>
#include <limits>
#include <numeric>
#include <random>

namespace sayuri_engine {

using sayuri_go::kBlack;
using sayuri_go::kMaxPoints;
using sayuri_go::kNoVertex;
using sayuri_go::kPassMove;
using sayuri_go::kWall;
using sayuri_go::kWhite;
using sayuri_go::SymmetryTables;

namespace {
constexpr double kLogitEps = 1e-8;
constexpr float kLogitZero = -1e6f;
inline float SafeLog(float v) { return static_cast<float>(std::log(static_cast<double>(v) + kLogitEps)); }
} // namespace

// ---------------------------------------------------------------------------------------------
// Expansion.
Network::Result Node::GetNetOutput(Network& network, GameState& state, bool is_root, Rng& rng) {
    // The root always reads the normal policy head and bypasses the cache (its temperature may differ);
    // inner nodes use the default head through the cache.  (SetOffset overrides the SetCache flag, as in the
    // reference's builder chain, node.cc:136-146.)
    const bool default_is_normal = network.GetDefaultPolicyOffset() == PolicyBufferOffset::kNormal;
    const auto offset = is_root ? PolicyBufferOffset::kNormal : PolicyBufferOffset::kDefault;
    const float temp = is_root ? param_->root_policy_temp : param_->policy_temp;
    const auto query = Network::Query::Get().SetTemperature(temp).SetCache(!default_is_normal).SetOffset(offset);
    return network.GetOutput(state, Network::kRandom, query, rng);
}

void Node::FillNodeEvalsFromNet(const Network::Result& net, NodeEvals& evals, int color) const {
    float wl = param_->use_stm_winrate ? net.stm_winrate : (net.wdl[0] - net.wdl[2] + 1) / 2;
    float score = net.final_score;
    if (color == kWhite) {
        wl = 1.0f - wl;
        score = 0.0f - score;
    }
    for (int i = 0; i < kMaxPoints; ++i) evals.black_ownership[static_cast<size_t>(i)] = color == kWhite ? 0.f - net.ownership[static_cast<size_t>(i)] : net.ownership[static_cast<size_t>(i)];
    evals.black_wl = wl;
    evals.draw = net.wdl[1];
    evals.black_final_score = score;
}

bool Node::ExpandChildren(Network& network, GameState& state, NodeEvals& evals, bool is_root, Rng& rng) {
    if (expanded_) return false;
    color_ = static_cast<std::uint8_t>(state.GetToMove());
    const Network::Result net = GetNetOutput(network, state, is_root, rng);

    FillNodeEvalsFromNet(net, evals, color_);
    black_wl_ = evals.black_wl;
    black_fs_ = evals.black_final_score;
    avg_black_ownership_.fill(0.f);

    std::pair<float, int> list_buf[kMaxPoints + 1];  // (policy, vertex) of the candidates; no heap traffic per expansion
    int list_n = 0;
    struct ListView {  // the few vector operations the code below uses
        std::pair<float, int>* p;
        int& n;
        void emplace_back(float a, int b) { p[n++] = std::make_pair(a, b); }
        bool empty() const { return n == 0; }
        size_t size() const { return static_cast<size_t>(n); }
        std::pair<float, int>* begin() { return p; }
        std::pair<float, int>* end() { return p + n; }
    } list{list_buf, list_n};
    float legal_sum = 0.0f;
    const int bs = state.GetBoardSize(), n = state.GetNumIntersections();
    bool safe[kMaxPoints];
    state.SafeAreaCached(safe);  // GetStrictSafeArea() without the vector<bool>

    // optional opening-stage pruning of moves that are mirror images of an already listed move
    const bool symm_prune = param_->symm_pruning && bs >= state.GetMoveNumber();
    std::vector<std::uint64_t> seen_hashes;
    std::uint64_t symm_base[SymmetryTables::kCount] = {0};
    for (int s = 0; symm_prune && s < SymmetryTables::kCount; ++s) symm_base[s] = state.ComputeSymmetryHash(s);

    for (int i = 0; i < n; ++i) {
        const int vtx = state.IndexToVertex(i);
        const float policy = net.probabilities[static_cast<size_t>(i)];
        // illegal moves and points inside pass-alive / pass-dead areas are never searched
        if (!state.IsLegalMove(vtx, color_) || safe[static_cast<size_t>(i)]) continue;
        if (symm_prune) {
            bool twin = false;
            for (int s = 1; s < SymmetryTables::kCount && !twin; ++s) {
                const int sv = SymmetryTables::Get().Vertex(bs, s, vtx);
                const std::uint64_t h = symm_base[s] ^ state.GetMoveHash(sv, color_);
                twin = std::find(seen_hashes.begin(), seen_hashes.end(), h) != seen_hashes.end();
            }
            if (twin) {
                legal_sum += policy; // still a legal move: counts in the normalisation
                continue;
            }
            seen_hashes.push_back(state.GetHash() ^ state.GetMoveHash(vtx, color_));
        }
        list.emplace_back(policy, vtx);
        legal_sum += policy;
    }

    // while most of the board is still open, pass is not a candidate
    const int open_threshold = std::max(0, static_cast<int>((1.0f - param_->suppress_pass_factor) * n));
    const bool suppress_pass = !list.empty() && static_cast<int>(list.size()) > open_threshold;
    if (!suppress_pass) {
        list.emplace_back(net.pass_probability, kPassMove);
        legal_sum += net.pass_probability;
    }
    if (legal_sum < 1e-8f) {
        for (auto& e : list) e.first = 1.f / list.size();
    } else {
        for (auto& e : list) e.first /= legal_sum;
    }
    // best policy first (ties: higher vertex first).  The reference stable-sorts the reversed range ascending by (policy,
    // vertex); vertices are unique, so that order is total and a plain descending sort gives the same sequence without the
    // merge buffer stable_sort allocates
    // Non-negative floats order like their bit patterns, so (policy bits << 16 | vertex) sorted as one integer is that same
    // order at a third of the cost of comparing pairs (the sort was the largest single item of an expansion); anything
    // else (a negative or NaN policy: never from a softmax) takes the pair comparison.
    {
        std::uint64_t keys[kMaxPoints + 1];
        bool plain = true;
        for (int i = 0; i < list_n; ++i) {
            const float p = list_buf[i].first;
            std::uint32_t bits;
            std::memcpy(&bits, &p, sizeof(bits));
            plain = plain && p >= 0.0f && bits != 0x80000000u && list_buf[i].second >= 0 && list_buf[i].second < 65536;
            keys[i] = (static_cast<std::uint64_t>(bits) << 16) | static_cast<std::uint64_t>(list_buf[i].second & 0xffff);
        }
        children_.reserve(list.size());
        if (plain) {
            // the best kSortedAtExpansion in place and in order, the others behind them as they fall (tree.h: sorted_n_)
            int sorted = list_n;
            if (list_n > 2 * kSortedAtExpansion && !is_root) {
                sorted = kSortedAtExpansion;
                std::nth_element(keys, keys + sorted, keys + list_n, std::greater<std::uint64_t>());
            }
            std::sort(keys, keys + sorted, std::greater<std::uint64_t>());
            sorted_n_ = static_cast<std::int16_t>(sorted);
            for (int i = 0; i < list_n; ++i) {  // the edges straight from the keys
                const std::uint32_t bits = static_cast<std::uint32_t>(keys[i] >> 16);
                float p;
                std::memcpy(&p, &bits, sizeof(p));
                children_.emplace_back(static_cast<int>(keys[i] & 0xffff), p);
            }
        } else {
            std::sort(list.begin(), list.end(), [](const std::pair<float, int>& a, const std::pair<float, int>& b) { return a > b; });
            for (const auto& e : list) children_.emplace_back(e.second, e.first);
            sorted_n_ = static_cast<std::int16_t>(list_n);
        }
    }
    expanded_ = true;
    return true;
}

// The rest of the children's order (tree.h: sorted_n_): bare edges only, by (policy, vertex) descending -- the keys of
// ExpandChildren's plain path (non-negative policies order like their bit patterns).
void Node::SortTail() const {
    auto* self = const_cast<Node*>(this);
    auto key = [](const Edge& e) {
        std::uint32_t bits;
        const float p = e.policy;
        std::memcpy(&bits, &p, sizeof(bits));
        return (static_cast<std::uint64_t>(bits) << 16) | static_cast<std::uint64_t>(static_cast<std::uint16_t>(e.vertex));
    };
    std::sort(self->children_.begin() + sorted_n_, self->children_.end(), [&key](const Edge& a, const Edge& b) { return key(a) > key(b); });
    sorted_n_ = static_cast<std::int16_t>(children_.size());
}

bool Node::SetTerminal(const NodeEvals* evals) {
    if (expanded_) return false;
    color_ = kWall; // no children
    black_wl_ = evals->black_wl;
    black_fs_ = evals->black_final_score;
    expanded_ = true;
    return true;
}

void Node::RecomputePolicy(Network& network, GameState& state, NodeEvals& evals, bool is_root, Rng& rng) {
    EnsureSorted();  // every child's place (tree.h: the order is finished on demand)
    if (!HasChildren()) return;
    const Network::Result net = GetNetOutput(network, state, is_root, rng);
    FillNodeEvalsFromNet(net, evals, state.GetToMove());
    std::vector<float> buf;
    for (auto& c : children_) {
        const int vtx = c.GetVertex();
        buf.push_back(vtx == kPassMove ? net.pass_probability : net.probabilities[static_cast<size_t>(state.VertexToIndex(vtx))]);
    }
    const float sum = std::accumulate(buf.begin(), buf.end(), 0.0f);
    for (auto& p : buf) p /= sum;
    size_t k = 0;
    for (auto& c : children_) c.Get()->policy_ = buf[k++];
}

bool Node::PrepareRootNode(Network& network, GameState& state, NodeEvals& evals, Rng& rng) {
    const bool fresh = ExpandChildren(network, state, evals, true, rng);
    EnsureSorted();  // a reused subtree's node may still carry its lazily ordered tail: finish it before any edge there gets a node
    InflateAllChildren();
    // a reused root may carry a policy computed with other settings (temperature): refresh it
    if (!fresh) RecomputePolicy(network, state, evals, true, rng);
    if (param_->dirichlet_noise) {
        const float alpha = param_->dirichlet_init * param_->dirichlet_factor / static_cast<float>(children_.size());
        ApplyDirichletNoise(alpha, rng);
    }
    KillRootSuperkos(state);
    UpdateScoreBonus(state, evals);
    BuildRootIndex();
    return fresh;
}

// search_params.h RootIndex: the root's children by descending search policy, and those that already carry visits (a reused
// subtree's) as the first "chosen" ones.  Called last in PrepareRootNode: the policies, the noise and the set of children are final.
void Node::BuildRootIndex() {
    auto& ri = param_->root_index;
    const int size = static_cast<int>(children_.size());
    ri.owner = nullptr;
    if (param_->gumbel || size == 0) return;  // Gumbel's own selection also descends root children: the full loop stays
    std::vector<std::pair<float, std::int16_t>> order(static_cast<size_t>(size));
    ri.chosen.clear();
    ri.is_chosen.assign(static_cast<size_t>(size), 0);
    for (int i = 0; i < size; ++i) {
        const Edge& c = children_[static_cast<size_t>(i)];
        Node* n = c.Get();
        if (!n) return;  // (never: InflateAllChildren ran)
        order[static_cast<size_t>(i)] = {GetSearchPolicy(c, true), static_cast<std::int16_t>(i)};
        if (n->GetVisits() > 0 || !n->IsActive() || n->HasChildren()) {  // anything but a fresh node is looked at individually
            ri.is_chosen[static_cast<size_t>(i)] = 1;
            ri.chosen.push_back(static_cast<std::int16_t>(i));
        }
    }
    std::sort(order.begin(), order.end(), [](const std::pair<float, std::int16_t>& a, const std::pair<float, std::int16_t>& b) {
        return a.first > b.first || (a.first == b.first && a.second < b.second);
    });
    ri.by_psa.resize(static_cast<size_t>(size));
    for (int i = 0; i < size; ++i) ri.by_psa[static_cast<size_t>(i)] = order[static_cast<size_t>(i)].second;
    ri.cursor = 0;
    ri.owner = this;
}

void Node::ApplyDirichletNoise(float alpha, Rng& rng) {
    EnsureSorted();  // every child's place (tree.h: the order is finished on demand)
    const size_t n = children_.size();
    std::vector<float> buf(n);
    std::gamma_distribution<float> gamma(alpha, 1.0f);
    std::generate(buf.begin(), buf.end(), [&]() { return gamma(rng); });
    const float sum = std::accumulate(buf.begin(), buf.end(), 0.0f);
    auto& noise = param_->dirichlet_buffer;
    noise.fill(0.0f);
    if (sum < std::numeric_limits<float>::min()) return; // zero or denormal: no noise
    for (auto& v : buf) v /= sum;
    for (size_t i = 0; i < n; ++i) noise[static_cast<size_t>(children_[i].GetVertex())] = buf[i];
}

void Node::KillRootSuperkos(GameState& state) {
    EnsureSorted();  // every child's place (tree.h: the order is finished on demand)
    for (auto& c : children_) {
        const int vtx = c.GetVertex();
        GameState fork = state;
        fork.PlayMove(vtx);
        if (vtx != kPassMove && fork.IsSuperko()) c.Get()->Invalidate();
    }
    children_.erase(std::remove_if(children_.begin(), children_.end(), [](Edge& e) { return !e.Get()->IsValid(); }),
                    children_.end());
    inflated_hi_ = static_cast<std::int16_t>(children_.size());
}

void Node::UpdateScoreBonus(GameState& state, NodeEvals& evals) {
    EnsureSorted();  // every child's place (tree.h: the order is finished on demand)
    if (!param_->first_pass_bonus) return;
    black_sb_ = 0.0f;
    InflateAllChildren();
    for (auto& c : children_) c.Get()->ComputeScoreBonus(state, evals);
}

void Node::ComputeScoreBonus(GameState& state, NodeEvals& parent) {
    if (!param_->first_pass_bonus || state.GetKoMove() != kNoVertex) {
        black_sb_ = 0.0f;
        return;
    }
    constexpr float kOwnerThreshold = 0.8f, kTail = 1.0f - kOwnerThreshold, kEndBonus = 0.5f;
    const int vtx = GetVertex(), color = state.GetToMove();
    float bonus = 0.0f;
    if (state.GetScoringRule() == sayuri_go::kAreaScoring) {
        // area scoring: nudge towards passing first; filling a seki point or a settled own point next to the
        // opponent costs nothing either
        if (vtx == kPassMove) {
            bonus += kEndBonus;
        } else if (state.IsSeki(vtx)) {
            bonus += kEndBonus;
        } else {
            const float owner = parent.black_ownership[static_cast<size_t>(state.VertexToIndex(vtx))];
            if ((owner > kOwnerThreshold && color == kBlack) || (owner < -kOwnerThreshold && color == kWhite)) {
                if (state.IsNeighborColor(vtx, color ^ 1)) bonus += kEndBonus;
            }
        }
        if (color == kWhite) bonus = 0.0f - bonus;
    } else {
        // territory scoring: discourage passing before dame are filled, and useless moves inside settled areas
        if (vtx == kPassMove) {
            bonus -= (2.f / 3.f) * kEndBonus;
        } else {
            const float owner = parent.black_ownership[static_cast<size_t>(state.VertexToIndex(vtx))];
            float factor = 0.0f;
            if (owner > kOwnerThreshold || owner < -kOwnerThreshold) factor = (std::abs(owner) - kOwnerThreshold) / kTail;
            bonus -= factor * kEndBonus;
        }
        if (color == kWhite) bonus = 0.0f - bonus;
    }
    black_sb_ = bonus;
}

// ---------------------------------------------------------------------------------------------
// Selection.
Node* Node::Inflate(Edge& e) {
    if (!e.node) {
        e.node.reset(new (shared_) Node(param_, shared_, e.vertex, e.policy));
        const int idx = static_cast<int>(&e - children_.data());
        if (idx + 1 > inflated_hi_) inflated_hi_ = static_cast<std::int16_t>(idx + 1);
    }
    return e.node.get();
}

void Node::InflateAllChildren() {
    EnsureSorted();  // every child's place (tree.h: the order is finished on demand)
    for (auto& c : children_) Inflate(c);
}

Node* Node::DescentSelectChild(int color, bool is_root, Rng& rng) {
    if (is_root && param_->gumbel) {
        if (Node* n = GumbelSelectChild(color, false, true, rng)) return n;
    }
    return PuctSelectChild(color, is_root);
}

Node* Node::ProbSelectChild(bool allow_pass) {
    EnsureSorted();  // every child's place (tree.h: the order is finished on demand)
    Edge* best = nullptr;
    float best_prob = std::numeric_limits<float>::lowest();
    for (auto& c : children_) {
        Node* n = c.Get();
        if (n && !n->IsActive()) continue;
        float prob = c.GetPolicy();
        if (!allow_pass && c.GetVertex() == kPassMove) prob = prob - 1e6f;
        if (prob > best_prob) {
            best_prob = prob;
            best = &c;
        }
    }
    return Inflate(*best);
}

float Node::GetFpu(int color, float total_visited_policy, bool is_root) const {
    // first-play urgency: the parent's network value blended towards its search value as more of the policy
    // mass has been tried, minus a reduction growing with that mass
    const float reduction_max = is_root ? param_->root_fpu_reduction : param_->fpu_reduction;
    const float reduction = reduction_max * std::sqrt(total_visited_policy);
    if (GetVisits() <= 0) return GetNetWL(color) - reduction;
    const float w = total_visited_policy * total_visited_policy;
    const float value = (1.0f - w) * GetNetWL(color) + w * GetWL(color, false);
    return value - reduction;
}

float Node::GetDynamicCpuctFactor(Node* node, int visits, int children_visits) const {
    if (!param_->cpuct_dynamic || node == nullptr || visits <= 1) return 1.0f;
    const double k_factor = param_->cpuct_dynamic_k_factor, k_base = param_->cpuct_dynamic_k_base;
    const double variance = node->GetWLVariance(1.0f, visits);
    const double stddev = std::sqrt(variance);
    double k = k_factor * (stddev / visits);
    k = std::max(0.5, k);
    k = std::min(1.4, k);
    const double alpha = 1.0 / (1.0 + std::sqrt(children_visits / k_base));
    k = alpha * k + (1.0 - alpha) * 1.0;
    return static_cast<float>(k);
}

float Node::GetCpuct(int children_visits) const {
    return param_->cpuct_init +
           param_->cpuct_base_factor * std::log((static_cast<float>(children_visits) + param_->cpuct_base + 1) / param_->cpuct_base);
}

int Node::GetForcedVisits(float policy, int children_visits, bool is_root) const {
    const float k = is_root ? param_->forced_playouts_k : 0.f;
    const float f = std::max(1e-4f, k * std::min(0.2f, policy) * static_cast<float>(children_visits));
    return static_cast<int>(std::sqrt(f));
}

float Node::GetSearchPolicy(const Edge& child, bool is_root) const {
    float policy = child.GetPolicy();
    if (is_root && param_->dirichlet_noise) {
        const float eps = param_->dirichlet_epsilon;
        const float eta = param_->dirichlet_buffer[static_cast<size_t>(child.GetVertex())];
        policy = policy * (1 - eps) + eps * eta;
    }
    return policy;
}

Node* Node::PuctSelectChild(int color, bool is_root) {
    // Same arithmetic and the same winner as the reference's loop over all children (node.cc:505-576); what is skipped
    // cannot win.  The children are sorted by policy, best first, and a bare edge (never descended) scores
    // fpu + cpuct * policy * sqrt(N): among bare edges the first in order has the largest value (every step of that
    // expression is monotone in the policy, and the comparison below is strict), so after the first bare edge the others are
    // not looked at, and beyond `inflated_hi_` there are only bare edges.  At the root the search policy carries Dirichlet
    // noise (not monotone in the edge's policy): every child is looked at there.  A node has ~360 children and a handful of
    // descended ones: the two loops were 8 % of all host time of a self-play rank.
    const int size = static_cast<int>(children_.size());
    const int hi = is_root ? size : std::min<int>(inflated_hi_, size);
    EnsureSorted(hi + 1);  // the children looked at below: [0, hi] (tree.h: the order is finished on demand)
    // the root's index (search_params.h RootIndex): the children chosen so far + the head of the others by search policy
    auto& ri = param_->root_index;
    const bool indexed = is_root && ri.owner == this && static_cast<int>(ri.by_psa.size()) == size;
    int children_visits = 0;
    float visited_policy = 0.0f;
    auto tally = [&](int i) {
        Edge& c = children_[static_cast<size_t>(i)];
        Node* n = c.Get();
        if (n && n->IsValid()) {
            const int v = n->GetVisits();
            children_visits += v;
            if (v > 0) visited_policy += c.GetPolicy();
        }
    };
    if (indexed) {
        for (const std::int16_t i : ri.chosen) tally(i);  // ascending: the float sum adds the same terms in the same order
    } else {
        for (int i = 0; i < hi; ++i) tally(i);
    }
    const float raw_cpuct = GetCpuct(children_visits);
    const float numerator = std::sqrt(static_cast<float>(children_visits));
    const float fpu = GetFpu(color, visited_policy, is_root);

    Edge* best = nullptr;
    float best_value = std::numeric_limits<float>::lowest();
    bool bare_seen = false;
    auto consider = [&](Edge& c, Edge*& best, float& best_value) {
        Node* n = c.Get();
        if (n && !n->IsActive()) return;
        float q = fpu;
        const float psa = GetSearchPolicy(c, is_root);
        float cpuct = raw_cpuct, denom = 1.0f;
        if (n) {
            const int visits = n->GetVisits();
            if (visits > 0) {
                // win/loss plus score utility steer the descent
                q = n->GetWL(color, true) + n->GetScoreEval(color);
                const int forced = GetForcedVisits(psa, children_visits, is_root);
                if (forced - visits > 0) q = static_cast<float>(q + (forced - visits) * 1e6);
            }
            cpuct *= GetDynamicCpuctFactor(n, visits, children_visits);
            denom += visits;
        }
        const float puct = cpuct * psa * (numerator / denom);
        const float value = q + puct;
        if (value > best_value) {
            best_value = value;
            best = &c;
        }
    };
    if (indexed) {
        // the chosen children one by one (ascending index: the first of equal values wins, as in the full loop) ...
        for (const std::int16_t i : ri.chosen) consider(children_[static_cast<size_t>(i)], best, best_value);
        // ... and of the never-chosen ones (fresh nodes: no visits, active, value = fpu + cpuct * psa * sqrt(N), monotone in psa)
        // those that share the largest value, i.e. the head of the psa order down to the first strictly smaller value
        while (ri.cursor < size && ri.is_chosen[static_cast<size_t>(ri.by_psa[static_cast<size_t>(ri.cursor)])]) ++ri.cursor;
        Edge* fresh_best = nullptr;
        float fresh_value = std::numeric_limits<float>::lowest();
        for (int k = ri.cursor; k < size; ++k) {
            const int i = ri.by_psa[static_cast<size_t>(k)];
            if (ri.is_chosen[static_cast<size_t>(i)]) continue;
            Edge* e = nullptr;
            float v = std::numeric_limits<float>::lowest();
            consider(children_[static_cast<size_t>(i)], e, v);
            if (!e) continue;
            if (!fresh_best) { fresh_best = e; fresh_value = v; }
            else if (v < fresh_value) break;
            else if (e < fresh_best) fresh_best = e;  // an equal value at a lower index
        }
        if (fresh_best && (fresh_value > best_value || (fresh_value == best_value && fresh_best < best))) {
            best = fresh_best;
            best_value = fresh_value;
        }
        if (best) {
            const int bi = static_cast<int>(best - children_.data());
            if (!ri.is_chosen[static_cast<size_t>(bi)]) {
                ri.is_chosen[static_cast<size_t>(bi)] = 1;
                ri.chosen.insert(std::lower_bound(ri.chosen.begin(), ri.chosen.end(), static_cast<std::int16_t>(bi)), static_cast<std::int16_t>(bi));
            }
        }
    } else {
        for (int i = 0; i < hi; ++i) {
            Edge& c = children_[static_cast<size_t>(i)];
            if (!c.Get() && !is_root) {
                if (bare_seen) continue;
                bare_seen = true;
            }
            consider(c, best, best_value);
        }
        if (!bare_seen && hi < size) consider(children_[static_cast<size_t>(hi)], best, best_value);  // the best of the bare edges beyond
    }
    // SAYURI_PUCT_CHECK=1 (tests): the reference's loop over ALL children must pick the same edge, and the two conditions the
    // pruning rests on must hold -- children sorted by policy, nothing but bare edges beyond inflated_hi_.  For ONE playout per
    // tree at a time (this engine's own search): with several threads in one tree (virtual loss) the statistics may change
    // between the two loops and the comparison can abort a correct search.
    static const bool check = std::getenv("SAYURI_PUCT_CHECK") != nullptr;
    if (check) {
        EnsureSorted();
        Edge* full = nullptr;
        float full_value = std::numeric_limits<float>::lowest();
        for (int i = 0; i < size; ++i) {
            Edge& c = children_[static_cast<size_t>(i)];
            consider(c, full, full_value);
            // (the root is exempt: every child is looked at there, and its policies carry noise)
            const bool sorted = is_root || i == 0 || !(children_[static_cast<size_t>(i - 1)].GetPolicy() < c.GetPolicy());
            if (!sorted || (i >= hi && c.Get())) {
                std::fprintf(stderr, "PuctSelectChild: %s at child %d of %d (inflated_hi %d, root %d)\n",
                             sorted ? "an inflated edge beyond inflated_hi_" : "children not sorted by policy", i, size, static_cast<int>(inflated_hi_), is_root ? 1 : 0);
                std::abort();
            }
        }
        if (full != best) {
            std::fprintf(stderr, "PuctSelectChild: the pruned loop picked child %d, the full loop child %d (of %d, root %d)\n",
                         static_cast<int>(best - children_.data()), static_cast<int>(full - children_.data()), size, is_root ? 1 : 0);
            std::abort();
        }
    }
    return Inflate(*best);
}

// ---------------------------------------------------------------------------------------------
// Statistics.
void Node::Update(const NodeEvals* evals) {
    // running means plus Welford-style squared-difference sums for the variance of value and score
    auto delta = [](double x, double old_acc, int old_n) {
        const double before = old_n > 0 ? x - old_acc / old_n : 0.0f;
        const double after = x - (old_acc + x) / (old_n + 1);
        return before * after;
    };
    const double eval = evals->black_wl, draw = evals->draw, score = evals->black_final_score;
    const int old_visits = visits_;
    const double d_eval = delta(eval, acc_black_wl_, old_visits);
    const double d_score = delta(score, acc_black_fs_, old_visits);
    visits_ += 1;
    acc_black_wl_ += eval;
    acc_draw_ += draw;
    acc_black_fs_ += score;
    sq_eval_diff_ += d_eval;
    sq_score_diff_ += d_score;
    for (size_t i = 0; i < kMaxPoints; ++i) {
        const double e = evals->black_ownership[i], avg = avg_black_ownership_[i];
        const double diff = (e - avg) / (old_visits + 1);
        avg_black_ownership_[i] = static_cast<float>(avg_black_ownership_[i] + diff);
    }
}

std::array<float, kMaxPoints> Node::GetOwnership(int color) const {
    std::array<float, kMaxPoints> out{};
    for (size_t i = 0; i < kMaxPoints; ++i) out[i] = color == kWhite ? 0.f - avg_black_ownership_[i] : avg_black_ownership_[i];
    return out;
}

float Node::GetFinalScore(int color) const {
    const double score = acc_black_fs_ / GetVisits();
    return color == kBlack ? static_cast<float>(score) : static_cast<float>(0.0f - score);
}

float Node::GetWL(int color, bool /*use_virtual_loss*/) const {
    // one playout walks a tree at a time, so no virtual loss is ever pending below the node being scored
    const double eval = acc_black_wl_ / GetVisits();
    return color == kBlack ? static_cast<float>(eval) : static_cast<float>(1.0f - eval);
}

float Node::GetScoreEval(int color) const {
    const float recent = color == kBlack ? param_->recent_expected_black_score : -param_->recent_expected_black_score;
    const float mean = GetFinalScore(color) + GetScoreBonus(color);
    const float stddev = GetScoreStddev();
    const float v = ScoreUtility::Get().Expected(mean, stddev, recent, param_->score_utility_div, static_cast<float>(param_->board_size));
    return v * param_->score_utility_factor;
}

float Node::GetScoreVariance(float default_var, int visits) const {
    return visits > 1 ? static_cast<float>(sq_score_diff_ / (visits - 1)) : default_var;
}
float Node::GetWLVariance(float default_var, int visits) const {
    return visits > 1 ? static_cast<float>(sq_eval_diff_ / (visits - 1)) : default_var;
}
float Node::GetScoreStddev() const { return std::sqrt(GetScoreVariance(1.0f, GetVisits())); }
float Node::GetWLStddev() const { return std::sqrt(GetWLVariance(1.0f, GetVisits())); }

float Node::GetLcb(int color) const {
    const int visits = GetVisits();
    if (visits <= 1) return GetPolicy() - 1e6f; // no variance yet
    const float mean = GetWL(color, false);
    const float stddev = std::sqrt(GetWLVariance(1.0f, visits));
    const float z = shared_->t_quantiles->At(visits - 1);
    return mean - z * (stddev / visits); // dividing by visits (not sqrt) makes the bound shrink more slowly
}

int Node::GetChildrenVisits() const {
    int sum = 0;
    for (const auto& c : children_)
        if (c.Get() && c.Get()->IsActive()) sum += c.Get()->GetVisits();
    return sum;
}

std::vector<std::pair<float, int>> Node::GetSortedLcbUtilityList(int color) {
    return GetSortedLcbUtilityList(color, GetChildrenVisits());
}

std::vector<std::pair<float, int>> Node::GetSortedLcbUtilityList(int color, int children_visits) {
    EnsureSorted();  // every child's place (tree.h: the order is finished on demand)
    const float reduction = std::min(std::max(0.f, param_->lcb_reduction), 1.f);
    std::vector<std::pair<float, int>> list;
    for (const auto& c : children_) {
        Node* n = c.Get();
        if (!n || !n->IsActive()) continue;
        const int visits = n->GetVisits();
        if (visits > 0) {
            const float mixed = n->GetLcb(color) + n->GetScoreEval(color);
            const float rlcb = mixed * (1.0f - reduction) + reduction * (static_cast<float>(visits) / children_visits);
            list.emplace_back(rlcb, n->GetVertex());
        }
    }
    std::stable_sort(list.rbegin(), list.rend());
    return list;
}

int Node::GetBestMove(bool allow_pass) {
    EnsureSorted();  // every child's place (tree.h: the order is finished on demand)
    const auto list = GetSortedLcbUtilityList(color_);
    float best_value = std::numeric_limits<float>::lowest();
    int best = kNoVertex;
    for (const auto& e : list) {
        if (e.first > best_value) {
            if (!allow_pass && e.second == kPassMove) continue;
            best_value = e.first;
            best = e.second;
        }
    }
    if (best == kNoVertex) best = ProbSelectChild(allow_pass)->GetVertex();
    return best;
}

Node* Node::GetChild(int vertex) {
    EnsureSorted();  // an edge inflated in the unsorted tail would be moved beyond inflated_hi_ by the next SortTail
    for (auto& c : children_)
        if (c.GetVertex() == vertex) return Inflate(c);
    return nullptr;
}

std::unique_ptr<Node> Node::PopChild(int vertex) {
    EnsureSorted();  // every child's place (tree.h: the order is finished on demand)
    for (auto it = children_.begin(); it != children_.end(); ++it) {
        if (it->GetVertex() == vertex) {
            Inflate(*it);
            std::unique_ptr<Node> out = std::move(it->node);
            children_.erase(it);
            inflated_hi_ = static_cast<std::int16_t>(children_.size());
            return out;
        }
    }
    return nullptr;
}

size_t Node::CountNodes() const {
    size_t n = 1;
    for (const auto& c : children_)
        if (c.Get()) n += c.Get()->CountNodes();
    return n;
}

// ---------------------------------------------------------------------------------------------
// Random move pickers.
int Node::GetRandomMoveProportionally(float temp, float min_ratio, int min_visits, Rng& rng) {
    EnsureSorted();  // every child's place (tree.h: the order is finished on demand)
    // visit-proportional pick, ignoring children below a relative / absolute visit floor
    double norm = 0, accum = 0;
    std::vector<std::pair<double, int>> table;
    int max_n = 0;
    for (const auto& c : children_) max_n = std::max(max_n, c.GetVisits());
    min_visits = std::max(static_cast<int>(std::round(max_n * min_ratio)), min_visits);
    for (const auto& c : children_) {
        const int visits = c.Get()->GetVisits();
        if (visits > min_visits) {
            if (norm == 0.0) norm = visits;
            const double val = visits / norm;
            accum += std::pow(val, (1.0 / temp));
            table.emplace_back(accum, c.Get()->GetVertex());
        }
    }
    if (table.empty()) return GetBestMove(true);
    std::uniform_real_distribution<double> dist(0.0, accum);
    const double pick = dist(rng);
    for (const auto& e : table)
        if (pick < e.first) return e.second;
    return kNoVertex;
}

int Node::GetRandomMoveWithLogitsQ(GameState& state, float temp, Rng& rng) {
    EnsureSorted();  // every child's place (tree.h: the order is finished on demand)
    const int n = state.GetNumIntersections();
    std::vector<float> prob(static_cast<size_t>(n + 1), 0.f);
    std::vector<int> vertices(static_cast<size_t>(n + 1), kNoVertex);
    int total = 0;
    for (const auto& c : children_) {
        const int visits = c.Get()->GetVisits();
        const int vtx = c.GetVertex();
        const size_t idx = static_cast<size_t>(state.board_.VertexToIndexOrPass(vtx));
        if (visits != 0) {
            total += visits;
            prob[idx] = static_cast<float>(visits);
            vertices[idx] = vtx;
        }
    }
    if (total == 0) return GetBestMove(true);
    for (float& p : prob) p /= static_cast<float>(total);
    MixLogitsCompletedQ(state, prob);
    double accum = 0;
    std::vector<std::pair<double, int>> table;
    for (size_t i = 0; i < prob.size(); ++i) {
        if (vertices[i] != kNoVertex) {
            accum += std::pow(static_cast<double>(prob[i]), (1.0 / temp));
            table.emplace_back(accum, vertices[i]);
        }
    }
    if (table.empty()) return GetRandomMoveProportionally(temp, 0.f, 0, rng);
    std::uniform_real_distribution<double> dist(0.0, accum);
    const double pick = dist(rng);
    for (const auto& e : table)
        if (pick < e.first) return e.second;
    return kNoVertex;
}

// ---------------------------------------------------------------------------------------------
// Gumbel / completed-Q machinery (node.cc:1469-1821).
float Node::GetGumbelEval(int color) const { return GetWL(color, false) + GetScoreEval(color); }

float Node::TransformCompletedQ(float completed_q, int max_visits) const {
    return (param_->gumbel_c_visit + std::min(param_->gumbel_playouts_threshold, max_visits)) * param_->gumbel_c_scale * completed_q;
}

std::vector<float> Node::GetProbLogitsCompletedQ(GameState& state) {
    EnsureSorted();  // every child's place (tree.h: the order is finished on demand)
    const int n = state.GetNumIntersections();
    std::vector<float> prob(static_cast<size_t>(n + 1), 0.f);
    float acc = 0.f;
    for (auto& c : children_) {
        acc += c.GetPolicy();
        prob[static_cast<size_t>(state.board_.VertexToIndexOrPass(c.GetVertex()))] = c.GetPolicy();
    }
    for (auto& v : prob) v /= acc;
    MixLogitsCompletedQ(state, prob);
    return prob;
}

void Node::MixLogitsCompletedQ(GameState& state, std::vector<float>& prob) {
    EnsureSorted();  // every child's place (tree.h: the order is finished on demand)
    const int n = state.GetNumIntersections();
    const int color = state.GetToMove();
    if (n + 1 != static_cast<int>(prob.size())) return;
    std::vector<float> logits(static_cast<size_t>(n + 1), kLogitZero);

    int max_visits = 0, children_visits = 0;
    float weighted_q = 0.f, weighted_pi = 0.f;
    for (auto& c : children_) {
        Node* node = c.Get();
        const int visits = (node && node->IsActive()) ? node->GetVisits() : 0;
        children_visits += visits;
        max_visits = std::max(max_visits, visits);
        if (visits > 0) {
            weighted_q += c.GetPolicy() * node->GetGumbelEval(color);
            weighted_pi += c.GetPolicy();
        }
    }
    // unvisited children get a value interpolated between the raw network value and the visited children's mean
    const float raw_value = GetNetWL(color);
    const float approx_q = (raw_value + (children_visits / weighted_pi) * weighted_q) / (1 + children_visits);
    for (auto& c : children_) {
        Node* node = c.Get();
        const int visits = (node && node->IsActive()) ? node->GetVisits() : 0;
        const float completed_q = visits == 0 ? approx_q : node->GetGumbelEval(color);
        const size_t idx = static_cast<size_t>(state.board_.VertexToIndexOrPass(c.GetVertex()));
        logits[idx] = SafeLog(prob[idx]) + TransformCompletedQ(completed_q, max_visits);
    }
    prob = logits;
    SoftmaxInPlace(prob.data(), n + 1, 1.f);

    // drop negligible entries and renormalise
    const double threshold = 1. / (100. + static_cast<double>(prob.size()));
    double kept = 0.;
    for (auto& v : prob) {
        if (v < threshold) v = 0.;
        else kept += v;
    }
    for (auto& v : prob) v = static_cast<float>(v / kept);
}

bool Node::ShouldApplyGumbel() const { return param_->gumbel && param_->gumbel_playouts_threshold > GetChildrenVisits(); }

bool Node::ProcessGumbelLogits(std::vector<float>& logits, int color, bool only_max_visits, Rng& rng) {
    EnsureSorted();  // every child's place (tree.h: the order is finished on demand)
    // Sequential-halving visit schedule over the top `considered` children; the child(ren) whose visit count
    // equals the schedule's current target get Gumbel(0,1) + log prior + transformed completed Q.
    const int size = static_cast<int>(children_.size());
    std::vector<std::pair<int, int>> table(static_cast<size_t>(size));
    logits.resize(static_cast<size_t>(size), kLogitZero);
    for (int i = 0; i < size; ++i) {
        Node* node = children_[static_cast<size_t>(i)].Get();
        if (node && node->IsValid() && node->IsActive()) {
            table[static_cast<size_t>(i)].first = node->GetVisits();
            table[static_cast<size_t>(i)].second = children_[static_cast<size_t>(i)].GetVertex();
        }
    }
    std::stable_sort(table.rbegin(), table.rend());
    const int max_visits = table[0].first;

    const int considered = std::min(param_->gumbel_considered_moves, size);
    int budget = param_->gumbel_playouts_threshold;
    const int prom = std::max(1, param_->gumbel_prom_visits);
    const int rounds = static_cast<int>(std::log2(std::max(1, considered)) + 1);
    const int top = static_cast<int>(std::pow(2, rounds - 1)); // power of two
    int target = 0, width = top, level = prom;

    if (only_max_visits) {
        budget = std::max(budget, 1);
        target = max_visits;
    } else {
        bool done = false;
        while (!done) {
            for (int i = 0; i < level && !done; ++i) {
                for (int j = 0; j < width; ++j) {
                    if (table[static_cast<size_t>(j)].first <= 0) {
                        target = GetChild(table[static_cast<size_t>(j)].second)->GetVisits();
                        done = true;
                        break;
                    }
                    table[static_cast<size_t>(j)].first -= 1;
                    budget -= 1;
                    if (budget <= 0) {
                        done = true;
                        break;
                    }
                }
            }
            if (done) break;
            if (width == 1) {
                width = top;
                level = prom;
            } else {
                width /= 2;
                level *= 2;
            }
        }
    }
    if (budget <= 0) return false;

    int count = 0;
    std::extreme_value_distribution<float> gumbel(0, 1);
    for (int i = 0; i < size; ++i) {
        auto& c = children_[static_cast<size_t>(i)];
        Node* node = c.Get();
        if (node && !node->IsActive()) continue;
        if (target == c.GetVisits()) {
            const float logit = gumbel(rng) + SafeLog(c.GetPolicy());
            float completed_q = 0.f;
            if (node && target > 0) completed_q = TransformCompletedQ(node->GetGumbelEval(color), max_visits);
            logits[static_cast<size_t>(i)] = logit + completed_q;
            count += 1;
        }
    }
    return count != 0;
}

Node* Node::GumbelSelectChild(int color, bool only_max_visits, bool allow_pass, Rng& rng) {
    EnsureSorted();  // every child's place (tree.h: the order is finished on demand)
    std::vector<float> logits;
    if (!ProcessGumbelLogits(logits, color, only_max_visits, rng)) return nullptr;
    Edge *best = nullptr, *best_no_pass = nullptr;
    float best_value = std::numeric_limits<float>::lowest();
    for (size_t i = 0; i < children_.size(); ++i) {
        if (logits[i] > best_value) {
            best_value = logits[i];
            best = &children_[i];
            if (children_[i].GetVertex() != kPassMove) best_no_pass = &children_[i];
        }
    }
    if (!allow_pass && best_no_pass) return Inflate(*best_no_pass);
    return Inflate(*best);
}

int Node::GetGumbelMove(bool allow_pass, Rng& rng) {
    EnsureSorted();  // every child's place (tree.h: the order is finished on demand)
    int candidates = 0;
    for (auto& c : children_)
        if (c.GetVisits() > 0 && c.Get()->IsValid()) candidates += 1;
    if (!allow_pass && candidates == 1) allow_pass = true; // the only candidate may be the pass
    Node* node = GumbelSelectChild(color_, true, allow_pass, rng);
    if (!node) return GetBestMove(allow_pass);
    return node->GetVertex();
}

} // namespace sayuri_engine
