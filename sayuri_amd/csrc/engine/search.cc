#include "search.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <random>
#include <stack>

#include "encoder.h"

namespace sayuri_engine {

using sayuri_go::Encoder;
using sayuri_go::kAreaScoring;
using sayuri_go::kBlack;
using sayuri_go::kEmpty;
using sayuri_go::kMaxPoints;
using sayuri_go::Position;
using sayuri_go::kNoVertex;
using sayuri_go::kPassMove;
using sayuri_go::kResignMove;
using sayuri_go::kTerritoryScoring;
using sayuri_go::kWhite;

namespace {
constexpr int kMaxPlayouts = std::numeric_limits<int>::max() / 2;

template <typename T> T KlDivergence(const std::vector<T>& p, const std::vector<T>& q, double floor = 1e-8) {
    T out = 0.f;
    if (p.size() != q.size()) return out;
    for (size_t i = 0; i < p.size(); ++i) {
        const double a = std::max(floor, static_cast<double>(p[i])), b = std::max(floor, static_cast<double>(q[i]));
        out += a * std::log(a / b);
    }
    return out;
}
} // namespace

// ---------------------------------------------------------------------------------------------
Search::Search(GameState& state, Network& network, const SearchParams& params)
    : root_state_(state), last_state_(state), network_(network), t_quantiles_(params.ci_alpha) {
    params_[0] = params;
    params_[1] = params; // exploration switched off (search.cc:37-46)
    params_[1].gumbel = false;
    params_[1].dirichlet_noise = false;
    params_[1].root_policy_temp = 1.f;
    params_[1].forced_playouts_k = 0.f;
    params_[1].no_exploring_phase = true;
    params_[1].kldgain_per_node = 0.0;
    params_[1].kldgain_interval = 0;
    active_ = &params_[0];
    passive_ = &params_[1];
    shared_.t_quantiles = &t_quantiles_;
    shared_.arena = &arena_;
}

void Search::Seed(std::uint64_t caller_seed, std::uint64_t playout_seed) {
    caller_rng_.Seed(caller_seed);
    playout_rng_.Seed(playout_seed);
}

// ---------------------------------------------------------------------------------------------
void Search::TryRecoverOwnershipMap(GameState& state, std::vector<int>& ownership) {
    // territory games: fall back to the network's ownership for everything that is not provably settled
    GameState fork = state;
    while (fork.GetPasses() >= 2) fork.UndoMove();
    if (fork.GetScoringRule() == kAreaScoring) return;
    fork.SetRule(kAreaScoring);
    constexpr float kThreshold = 0.8f;
    const auto net = network_.GetOutput(fork, Network::kRandom, Network::Query{}, playout_rng_);
    const int color = fork.GetToMove(), n = fork.GetNumIntersections();
    const auto safe = fork.GetStrictSafeArea();
    for (int i = 0; i < n; ++i) {
        float black_owner = net.ownership[static_cast<size_t>(i)];
        if (color == kWhite) black_owner = 0.f - black_owner;
        if (safe[static_cast<size_t>(i)]) continue;
        ownership[static_cast<size_t>(i)] = black_owner > kThreshold ? kBlack : black_owner < -kThreshold ? kWhite : kEmpty;
    }
}

void Search::GameOverEvals(GameState& state, PlayoutResult& result) {
    result.valid = true;
    auto ownership = state.GetOwnership();
    TryRecoverOwnershipMap(state, ownership);
    for (size_t i = 0; i < ownership.size(); ++i)
        result.evals.black_ownership[i] = ownership[i] == kBlack ? 1.f : ownership[i] == kWhite ? -1.f : 0.f;
    const float score = state.GetFinalScore(kBlack, ownership);
    result.evals.black_final_score = score;
    if (score > 1e-4) {
        result.evals.black_wl = 1.0f;
        result.evals.draw = 0.0f;
    } else if (score < -1e-4) {
        result.evals.black_wl = 0.0f;
        result.evals.draw = 0.0f;
    } else {
        result.evals.black_wl = 0.5f;
        result.evals.draw = 1.0f;
    }
}

void Search::PlaySimulation(GameState& state, Node* node, int depth, PlayoutResult& result) {
    const bool end_by_passes = state.GetPasses() >= 2;
    const int scoring = state.GetScoringRule();
    if (end_by_passes) {
        if (scoring == kAreaScoring) {
            GameOverEvals(state, result);
        } else {
            // territory scoring: the network judges area scoring more reliably, so drop the passes, switch the
            // rule (compensating the komi) and keep playing
            while (state.GetLastMove() == kPassMove) state.UndoMove();
            const float komi = state.GetKomi();
            const float offset = state.GetPenaltyOffset(kAreaScoring, kTerritoryScoring);
            state.SetRule(kAreaScoring);
            state.SetKomi(komi + offset);
        }
    }
    if (node->Expandable()) {
        const int last_move = state.GetLastMove();
        if (end_by_passes && scoring == kAreaScoring) {
            if (result.valid) node->SetTerminal(&result.evals);
        } else if (last_move != kPassMove && state.IsSuperko()) {
            node->Invalidate(); // superko positions are pruned
        } else {
            const bool had_children = node->HasChildren();
            NodeEvals evals{};
            const bool ok = node->ExpandChildren(network_, state, evals, false, playout_rng_);
            if (!had_children && ok) {
                result.valid = true;
                result.evals = evals;
            }
        }
    }
    if (node->HasChildren() && !result.valid) {
        const int color = state.GetToMove();
        Node* next = node->DescentSelectChild(color, depth == 0, playout_rng_);
        state.PlayMove(next->GetVertex(), color);
        PlaySimulation(state, next, depth + 1, result);
    }
    if (result.valid) node->Update(&result.evals);
}

// ---------------------------------------------------------------------------------------------
void Search::PrepareParam() {
    active_->recent_expected_black_score = root_->GetFinalScore(kBlack);
    active_->board_size = root_state_.GetBoardSize();
}

bool Search::AdvanceToNewRootState(int tag) {
    if (!root_) return false;
    const int depth = root_state_.GetMoveNumber() - last_state_.GetMoveNumber();
    if (depth < 0) return false;
    std::stack<int> moves;
    GameState test = root_state_;
    for (int i = 0; i < depth; ++i) {
        moves.push(test.GetLastMove());
        test.UndoMove();
    }
    if (test.GetHash() != last_state_.GetHash() || test.GetBoardSize() != last_state_.GetBoardSize()) return false;
    while (!moves.empty()) {
        const int vtx = moves.top();
        std::unique_ptr<Node> next = root_->PopChild(vtx);
        root_ = std::move(next); // the rest of the old tree is dropped here
        if (!root_) return false;
        last_state_.PlayMove(vtx);
        moves.pop();
    }
    if (root_state_.GetHash() != last_state_.GetHash()) return false;
    if (!root_->HasChildren()) return false; // a bare edge carries nothing worth keeping
    if ((tag & kUnreused) && active_->gumbel) {
        // sequential halving needs most of its budget on a fresh distribution
        const int remaining = active_->playouts - (root_->GetVisits() - 1);
        if (remaining < active_->gumbel_playouts_threshold) return false;
    }
    return true;
}

void Search::PrepareRootNode(ComputationResult& result, int tag) {
    const bool reused = AdvanceToNewRootState(tag);
    if (!reused) {
        // a fresh root (the first move of a game, or a position the old tree does not contain): the old tree goes first, and
        // with the arena empty its slabs above the cap go back to the system (tree_arena.h)
        root_.reset();
        static const std::size_t keep = [] {
            const char* e = std::getenv("SAYURI_AB_ARENA_KEEP_MB");  // measuring aid
            return e ? static_cast<std::size_t>(std::atol(e)) << 20 : kArenaKeepBytes;
        }();
        arena_.Reset(keep);
        root_.reset(new (&shared_) Node(active_, &shared_, kPassMove, 1.0f));
    }
    playouts_ = 0;
    root_evals_ = NodeEvals{};
    const bool fresh = root_->PrepareRootNode(network_, root_state_, root_evals_, caller_rng_);
    if (!reused && fresh) root_->Update(&root_evals_);

    // the raw (temperature 1) root policy, from the cache when it is there
    const auto net = network_.GetOutput(root_state_, Network::kRandom, Network::Query{}, caller_rng_);
    const int n = root_state_.GetNumIntersections();
    root_raw_probabilities_.assign(net.probabilities.begin(), net.probabilities.begin() + n);
    root_raw_probabilities_.push_back(net.pass_probability);

    UpdateComputationResult(result);
    prev_kld_visits_ = result.visits;
    prev_kld_policy_.assign(result.target_policy_dist.begin(), result.target_policy_dist.begin() + (n + 1));
    PrepareParam();
}

int Search::GetPlayoutsLeft(int cap, int tag) const {
    int done = playouts_;
    if (tag & kUnreused) done = root_->GetVisits() - 1; // visit cap: the root's own visit does not count
    return std::max(cap - done, 0);
}

bool Search::HaveAlternateMoves() const {
    size_t valid = 0;
    for (const auto& c : root_->GetChildren())
        if (c.Get()->IsActive()) ++valid;
    return valid != 1; // a single candidate needs no search
}

bool Search::StoppedByKldGain(ComputationResult& result, int tag) {
    const int visits_diff = root_->GetVisits() - prev_kld_visits_;
    if (active_->kldgain_interval <= 0 || visits_diff < active_->kldgain_interval) return false;
    UpdateComputationResult(result);
    const int n = root_state_.GetNumIntersections();
    std::vector<double> now(result.target_policy_dist.begin(), result.target_policy_dist.begin() + (n + 1));
    const double gain = KlDivergence(now, prev_kld_policy_);
    bool stop = gain / visits_diff < active_->kldgain_per_node;
    prev_kld_visits_ = result.visits;
    prev_kld_policy_ = now;
    if (active_->fastsearch_playouts > 0 && active_->fastsearch_playouts_prob > 0.0 && !AchieveCap(active_->fastsearch_playouts, tag)) stop = false;
    return stop;
}

ComputationResult Search::Computation(int playouts, int tag) {
    ComputationResult result;
    playouts = std::min(playouts, kMaxPlayouts);
    int removed_passes = 0;
    if (tag & kForced) {
        while (root_state_.GetPasses() >= 2) {
            root_state_.UndoMove();
            root_state_.UndoMove();
            removed_passes += 2;
        }
    }
    if (tag & kNoExploring) std::swap(active_, passive_);

    const int color = root_state_.GetToMove();
    result.to_move = color;
    result.board_size = root_state_.GetBoardSize();
    result.komi = root_state_.GetKomi();
    result.movenum = root_state_.GetMoveNumber();
    result.visits = root_ ? root_->GetVisits() : 0;
    result.playouts = 0;

    if (root_state_.IsGameOver()) {
        result.high_priority_move = kPassMove;
        if (tag & kNoExploring) std::swap(active_, passive_);
        return result;
    }

    // fold the move log into its shared prefix: every fork of the root below (one per playout, one per root
    // child in the superko filter) then copies a pointer instead of the frames
    root_state_.Freeze();
    PrepareRootNode(result, tag);

    bool running = !AchieveCap(playouts, tag);
    last_single_candidate_ = false;
    if (running && !HaveAlternateMoves()) {
        running = false;
        single_candidate_searches_ += 1;
        last_single_candidate_ = true;
    }
    while (running) {
        GameState fork = root_state_;
        PlayoutResult pr;
        PlaySimulation(fork, root_.get(), 0, pr);
        if (pr.valid) {
            playouts_ += 1;
            total_playouts_.fetch_add(1, std::memory_order_relaxed);
        }
        if (AchieveCap(playouts, tag)) running = false;
        else if (abort_ && abort_->load(std::memory_order_relaxed)) running = false;
        else if (active_->kldgain_interval > 0 && StoppedByKldGain(result, tag)) running = false;
    }
    UpdateComputationResult(result);
    last_state_ = root_state_;

    if (tag & kForced)
        for (int i = 0; i < removed_passes; ++i) root_state_.PlayMove(kPassMove);
    if (tag & kNoExploring) std::swap(active_, passive_);
    return result;
}

void Search::UpdateComputationResult(ComputationResult& result) {
    const int color = root_state_.GetToMove();
    const int n = root_state_.GetNumIntersections();
    const Position& board = root_state_.board_;
    result.visits = root_->GetVisits();
    result.playouts = playouts_;
    PrepareParam();

    result.best_move = root_->GetBestMove(true);
    result.best_no_pass_move = root_->GetBestMove(false);
    if (active_->gumbel || active_->no_exploring_phase) {
        result.random_move = root_->GetRandomMoveWithLogitsQ(root_state_, 1.f, caller_rng_);
    } else {
        result.random_move = root_->GetRandomMoveProportionally(active_->random_moves_temp, active_->random_min_ratio,
                                                                 active_->random_min_visits, caller_rng_);
    }
    result.gumbel_move = root_->GetGumbelMove(true, caller_rng_);
    result.gumbel_no_pass_move = root_->GetGumbelMove(false, caller_rng_);
    result.root_score_lead = root_->GetFinalScore(color);
    result.root_eval = root_->GetWL(color, false);
    result.root_score_stddev = root_->GetScoreStddev();
    result.root_eval_stddev = root_->GetWLStddev();
    {
        Node* best = root_->GetChild(result.best_move);
        result.best_eval = best->GetVisits() >= 1 ? best->GetWL(color, false) : result.root_eval;
    }
    result.side_resign = result.root_eval < active_->resign_threshold || result.root_eval > (1.f - active_->resign_threshold);

    // ---- training targets
    result.root_ownership.assign(static_cast<size_t>(n), 0);
    result.root_searched_visits.assign(static_cast<size_t>(n + 1), 0);
    result.root_estimated_q.assign(static_cast<size_t>(n + 1), 0);
    result.root_visits_dist.assign(static_cast<size_t>(n + 1), 0);
    result.target_policy_dist.assign(static_cast<size_t>(n + 1), 0);
    const auto ownership = root_->GetOwnership(color);
    std::copy(ownership.begin(), ownership.begin() + n, result.root_ownership.begin());

    int children_visits = 0;
    float visited_policy = 0.0f;
    auto& children = root_->GetChildren();
    for (const auto& c : children) {
        const int v = c.Get()->GetVisits();
        children_visits += v;
        if (v > 0) visited_policy += c.Get()->GetPolicy();
    }
    for (const auto& c : children) {
        Node* node = c.Get();
        const int visits = node->GetVisits();
        const size_t idx = static_cast<size_t>(board.VertexToIndexOrPass(node->GetVertex()));
        result.root_searched_visits[idx] = visits;
        // NB the child's own FPU, evaluated with the root's visited-policy mass (search.cc:548-549)
        result.root_estimated_q[idx] = visits == 0 ? node->GetFpu(color, visited_policy, true) : node->GetWL(color, false) + node->GetScoreEval(color);
    }
    if (children_visits == 0) {
        for (auto& v : result.root_visits_dist) v = 1.f / (n + 1);
    } else {
        for (int i = 0; i < n + 1; ++i) result.root_visits_dist[static_cast<size_t>(i)] = static_cast<float>(result.root_searched_visits[static_cast<size_t>(i)]) / children_visits;
    }

    const auto completed_q_policy = root_->GetProbLogitsCompletedQ(root_state_);
    if (children_visits == 0) {
        result.target_policy_dist = result.root_visits_dist;
    } else if (root_->ShouldApplyGumbel() || active_->always_completed_q_policy) {
        result.target_policy_dist = completed_q_policy;
    } else {
        // blend visits with the completed-Q policy (all visits from 800 on), then prune visits PUCT would not
        // have spent on a move given how much worse its value is than the best move's ("policy target pruning")
        const float damping = 800.f;
        auto target = result.root_visits_dist;
        for (int i = 0; i < n + 1; ++i) {
            const float factor = std::min(std::min(children_visits, static_cast<int>(damping)) / damping, 1.0f);
            target[static_cast<size_t>(i)] = factor * target[static_cast<size_t>(i)] + (1.0f - factor) * completed_q_policy[static_cast<size_t>(i)];
        }
        int best = 0;
        for (int i = 0; i < n + 1; ++i)
            if (target[static_cast<size_t>(i)] > target[static_cast<size_t>(best)]) best = i;
        const int virtual_visits = std::max(3200, children_visits);
        const float cpuct = root_->GetCpuct(virtual_visits);
        float accum = 0.0f;
        for (int i = 0; i < n + 1; ++i) {
            if (i != best) {
                const float scaling = cpuct * root_raw_probabilities_[static_cast<size_t>(i)] * virtual_visits;
                const float diff = result.root_estimated_q[static_cast<size_t>(best)] - result.root_estimated_q[static_cast<size_t>(i)];
                if (diff > 0) {
                    const int wanted = std::max(0, static_cast<int>(std::round(scaling / diff)) - 1);
                    const float wanted_prob = static_cast<float>(wanted) / virtual_visits;
                    target[static_cast<size_t>(i)] = std::min(wanted_prob, target[static_cast<size_t>(i)]);
                }
            }
            accum += target[static_cast<size_t>(i)];
        }
        if (accum < 1e-4f) {
            result.target_policy_dist = result.root_visits_dist;
        } else {
            for (auto& v : target) v /= accum;
            result.target_policy_dist = target;
        }
    }
    result.policy_kld = KlDivergence(result.target_policy_dist, root_raw_probabilities_);

    // ---- dead / alive chains by searched ownership (provably settled points count as certain)
    constexpr float kOwnerThreshold = 0.75f;
    const auto safe_owner = root_state_.GetOwnership();
    const auto safe = root_state_.GetStrictSafeArea();
    std::vector<std::vector<int>> alive, dead;
    auto chain_of = [&](int vtx) {
        int buf[kMaxPoints];
        const int m = board.ChainMembers(vtx, buf);
        return std::vector<int>(buf, buf + m);
    };
    auto owner_at = [&](int i) -> float {
        return safe[static_cast<size_t>(i)] ? 2 * static_cast<float>(safe_owner[static_cast<size_t>(i)] == color) - 1 : result.root_ownership[static_cast<size_t>(i)];
    };
    for (int i = 0; i < n; ++i) {
        const int vtx = root_state_.IndexToVertex(i);
        const float owner = owner_at(i);
        const int s = root_state_.GetState(vtx);
        if (owner > kOwnerThreshold) {
            if (s == color) alive.push_back(chain_of(vtx));
            else if (s == (color ^ 1)) dead.push_back(chain_of(vtx));
        } else if (owner < -kOwnerThreshold) {
            if (s == (color ^ 1)) alive.push_back(chain_of(vtx));
            else if (s == color) dead.push_back(chain_of(vtx));
        }
    }
    std::sort(alive.begin(), alive.end());
    alive.erase(std::unique(alive.begin(), alive.end()), alive.end());
    std::sort(dead.begin(), dead.end());
    dead.erase(std::unique(dead.begin(), dead.end()), dead.end());
    result.alive_strings = alive;
    result.dead_strings = dead;

    if (active_->capture_all_dead) {
        std::vector<int> fill;
        const auto raw = root_state_.GetRawOwnership();
        for (int i = 0; i < n; ++i) {
            const int vtx = root_state_.IndexToVertex(i);
            if (owner_at(i) > kOwnerThreshold && root_state_.IsLegalMove(vtx, color)) {
                if (raw[static_cast<size_t>(i)] == kEmpty && root_state_.IsNeighborColor(vtx, color)) fill.push_back(vtx);
                if (raw[static_cast<size_t>(i)] == (color ^ 1)) fill.push_back(vtx);
            }
        }
        if (!fill.empty()) {
            std::shuffle(fill.begin(), fill.end(), caller_rng_);
            std::sort(fill.begin(), fill.end(), [&](const int& a, const int& b) {
                return static_cast<int>(board.IsCaptureMove(a, color)) > static_cast<int>(board.IsCaptureMove(b, color));
            });
            for (int move : fill) {
                GameState fork = root_state_;
                fork.PlayMove(move, color);
                if (!fork.IsSuperko()) {
                    result.capture_all_dead_move = move;
                    break;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
bool ShouldResign(GameState& state, ComputationResult& result, const SearchParams* param) {
    const int movenum = state.GetMoveNumber(), n = state.GetNumIntersections(), bs = state.GetBoardSize();
    float threshold = param->resign_threshold;
    if (threshold <= 0.0f || movenum <= n / 4 || state.IsGameOver()) return false;
    // a komi handicap lowers the bar early in the game
    const float komi_diff = state.GetKomi() - 7.0f;
    const int to_move = state.GetToMove();
    if ((komi_diff > 0.f && to_move == kBlack) || (komi_diff < 0.f && to_move == kWhite)) {
        const float blend = std::min(1.0f, movenum / (0.6f * n));
        const float blended = blend * threshold + (1.0f - blend) * threshold / std::max(1.f, std::abs(5.f * komi_diff / bs));
        threshold = std::min(blended, threshold);
    }
    const int handicap = state.GetHandicap();
    if (handicap > 0 && to_move == kWhite) {
        const float handicap_threshold = (threshold - 1.f) * handicap / 20.f;
        const float blend = std::min(1.0f, movenum / (0.6f * n));
        const float blended = blend * threshold + (1.0f - blend) * handicap_threshold;
        threshold = std::min(blended, threshold);
    }
    return result.best_eval < threshold;
}

bool ShouldPass(GameState& state, ComputationResult& result, const SearchParams* param) {
    if (!param->friendly_pass || state.GetLastMove() != kPassMove || state.GetScoringRule() != kAreaScoring) return false;
    const int n = state.GetNumIntersections();
    if (state.GetMoveNumber() <= n / 3) return false;
    std::vector<int> dead;
    GameState fork = state;
    for (const auto& chain : result.dead_strings)
        for (int v : chain) dead.push_back(v);
    fork.RemoveDeadStrings(dead);
    for (int i = 0; i < n; ++i) {
        const int vtx = fork.IndexToVertex(i);
        if (fork.GetState(vtx) != kEmpty && fork.GetLiberties(vtx) == 1) return false; // a live chain in atari
    }
    return fork.GetFinalScore(state.GetToMove()) > 0.1f;
}

bool ShouldForbidPass(GameState& state, ComputationResult& result, NodeEvals& root_evals) {
    // self-play never resigns: final positions feed the targets, so passing waits until the board is settled
    const int n = state.GetNumIntersections();
    if (state.GetMoveNumber() <= n / 6) return true;
    if (state.GetScoringRule() == kTerritoryScoring) return false;
    const int to_move = result.to_move;
    const auto safe_owner = state.GetOwnership();
    for (const auto& chain : result.dead_strings) {
        const int vtx = chain[0];
        if (state.GetState(vtx) == (to_move ^ 1) && safe_owner[static_cast<size_t>(state.VertexToIndex(vtx))] != to_move) return true;
    }
    constexpr float kRawThreshold = 0.8f;
    for (int i = 0; i < n; ++i) {
        float owner = root_evals.black_ownership[static_cast<size_t>(i)];
        if (to_move == kWhite) owner = 0.f - owner;
        if (owner >= kRawThreshold && safe_owner[static_cast<size_t>(i)] != to_move) return true;
    }
    constexpr int kMaxEmptyGroup = 8;
    bool seen[sayuri_go::kMaxVertices] = {false};
    for (int i = 0; i < n; ++i) {
        const int vtx = state.IndexToVertex(i);
        if (safe_owner[static_cast<size_t>(i)] == kEmpty && !seen[vtx]) {
            if (state.board_.ReachGroup(vtx, kEmpty, seen) >= kMaxEmptyGroup) return true; // unsettled open area
        }
    }
    return false;
}

int Search::GetBestMove(int playouts, int tag) {
    auto result = Computation(playouts, tag);
    if (ShouldResign(root_state_, result, active_)) return kResignMove;
    if (result.high_priority_move != kNoVertex) return result.high_priority_move;
    int move = result.best_move;
    const int random_moves = static_cast<int>(active_->random_moves_factor * root_state_.GetNumIntersections());
    if (root_state_.GetMoveNumber() < random_moves) move = result.random_move;
    if (ShouldPass(root_state_, result, active_)) move = kPassMove;
    if (active_->capture_all_dead && move == kPassMove && root_state_.GetScoringRule() == kAreaScoring &&
        result.capture_all_dead_move != kNoVertex)
        move = result.capture_all_dead_move;
    return move;
}

int Search::ThinkBestMove() {
    const int tag = active_->reuse_tree ? kThinking : (kThinking | kUnreused);
    return GetBestMove(active_->playouts, tag);
}

int Search::GetSelfPlayMove(int tag) {
    // without tree reuse the cap is on root visits (playout-cap oscillation as in KataGo), not on playouts
    if (!active_->reuse_tree) tag |= kUnreused;
    const bool already_lost = training_buffer_.empty() ? false : training_buffer_.back().accum_resign_cnt > 0;
    const int random_moves = static_cast<int>(active_->random_moves_factor * root_state_.GetNumIntersections());
    const bool opening_random = root_state_.GetMoveNumber() < random_moves;

    int playouts = active_->playouts;
    float fast_prob = active_->fastsearch_playouts_prob;
    if (already_lost) {
        // decided games: record only a fraction of the positions
        const float record_prob = (1.0f - fast_prob) * (1.0f - active_->resign_discard_prob);
        fast_prob = 1.0f - record_prob;
    }
    if (active_->fastsearch_playouts > 0 && active_->fastsearch_playouts < active_->playouts && caller_rng_.Chance(fast_prob)) {
        playouts = std::min(playouts, active_->fastsearch_playouts);
        if (already_lost) playouts = std::min(playouts, active_->resign_playouts);
        tag |= kNoExploring;
    }
    if (!network_.Valid()) playouts /= 10; // dummy backend: random playouts, keep them cheap
    playouts = std::max(1, playouts);

    auto result = Computation(playouts, tag);
    const bool is_gumbel = root_->ShouldApplyGumbel() && !(tag & kNoExploring);
    int move = is_gumbel ? result.gumbel_move : result.best_move;
    const bool forbid_pass = ShouldForbidPass(root_state_, result, root_evals_);
    if (forbid_pass) move = is_gumbel ? result.gumbel_no_pass_move : result.best_no_pass_move;

    float root_eval = result.root_eval, root_score = result.root_score_lead;
    if ((opening_random && !is_gumbel) || (!already_lost && (tag & kNoExploring) && caller_rng_.Chance(active_->random_fastsearch_prob))) {
        if (!(forbid_pass && result.random_move == kPassMove)) move = result.random_move;
    }
    const bool discard = (tag & kNoExploring) != 0; // fast searches are not recorded
    if (result.to_move == kWhite) {
        root_eval = 1.0f - root_eval;
        root_score = 0.f - root_score;
    }
    char buf[128];
    std::snprintf(buf, sizeof(buf), "%d, %d, %.2f, %.2f, %.2f, %c", result.playouts, result.visits, root_eval, root_score,
                  result.policy_kld, discard ? 'F' : 'T');
    last_comment_ = buf;
    if (!(tag & kNoBuffer)) {
        if (last_single_candidate_) single_candidate_records_.push_back(static_cast<int>(training_buffer_.size()));
        GatherData(root_state_, result, discard);
    }
    return move;
}

void Search::UpdateTerritoryHelper() {
    GameState saved = root_state_;
    if (root_state_.GetScoringRule() == kTerritoryScoring) {
        // play the position out under area scoring until all dead stones are gone
        while (root_state_.GetLastMove() == kPassMove) root_state_.UndoMove();
        const float komi = root_state_.GetKomi();
        const float offset = root_state_.GetPenaltyOffset(kAreaScoring, kTerritoryScoring);
        root_state_.SetRule(kAreaScoring);
        root_state_.SetKomi(komi + offset);
        while (!root_state_.IsGameOver()) root_state_.PlayMove(GetSelfPlayMove(kNoExploring | kNoBuffer));
    }
    GameState end_state = root_state_;
    root_state_ = saved;
    root_state_.SetTerritoryHelper(end_state.GetOwnership());
}

// ---------------------------------------------------------------------------------------------
void Search::GatherData(const GameState& state, ComputationResult& result, bool discard) {
    if (training_buffer_.size() > 9999) return;
    TrainingData d;
    d.version = 2; // encoder version
    d.mode = 0;
    d.discard = discard;
    d.board_size = result.board_size;
    d.komi = state.GetKomiWithPenalty();
    d.side_to_move = result.to_move;
    d.q_value = 2 * result.root_eval - 1.f;
    d.score_lead = result.root_score_lead;
    d.score_stddev = result.root_score_stddev;
    d.q_stddev = result.root_eval_stddev;
    {
        sayuri_host::PackedPlanes pp;
        Encoder::Packed(state, 0, -1, &pp);  // the same planes Encoder::Planes(state, 0, -1) fills, bit-packed
        d.plane_bits.assign(&pp.bits[0][0], &pp.bits[0][0] + static_cast<size_t>(pp.binary_planes) * sayuri_host::PackedPlanes::kWords);
    }
    d.probabilities = result.target_policy_dist;
    d.wave = state.GetWave();
    d.rule = state.GetScoringRule() == kAreaScoring ? 0.f : 1.f;
    d.kld = result.policy_kld;
    d.accum_resign_cnt = training_buffer_.empty() || !result.side_resign ? 0 : training_buffer_.back().accum_resign_cnt + 1;
    training_buffer_.push_back(std::move(d));
}

void Search::GatherTrainingBuffer(std::vector<TrainingData>& chunk) {
    const auto ownership = root_state_.GetOwnership();
    const int n = root_state_.GetNumIntersections();
    const float black_score = root_state_.GetFinalScore(kBlack);
    int winner = sayuri_go::kUndecided;
    if (std::abs(black_score) < 1e-4f) winner = sayuri_go::kDrawGame;
    else if (black_score > 0) winner = sayuri_go::kBlackWon;
    else if (black_score < 0) winner = sayuri_go::kWhiteWon;

    const int size = static_cast<int>(training_buffer_.size());
    for (int i = 0; i < size; ++i) {
        auto& buf = training_buffer_[static_cast<size_t>(i)];
        if (winner == sayuri_go::kDrawGame) {
            buf.final_score = 0;
            buf.result = 0;
        } else {
            buf.result = winner == buf.side_to_move ? 1 : -1;
            buf.final_score = buf.side_to_move == kBlack ? black_score : -black_score;
        }
        buf.ownership.assign(static_cast<size_t>(n), 0);
        for (int k = 0; k < n; ++k) {
            const int owner = ownership[static_cast<size_t>(k)];
            buf.ownership[static_cast<size_t>(k)] = owner == buf.side_to_move ? 1 : owner == (buf.side_to_move ^ 1) ? -1 : 0;
        }
        // windowed means of the searched value / score around this move
        const int half = std::max(3, buf.board_size / 2);
        float q_sum = 0.f, s_sum = 0.f;
        int count = 0;
        for (int w = -half; w <= half; ++w) {
            const int j = i + w;
            if (j < 0 || j >= size) continue;
            const auto& o = training_buffer_[static_cast<size_t>(j)];
            const float sign = o.side_to_move == buf.side_to_move ? 1.f : -1.f;
            if (sign > 0) {
                q_sum += o.q_value;
                s_sum += o.score_lead;
            } else {
                q_sum -= o.q_value;
                s_sum -= o.score_lead;
            }
            count += 1;
        }
        buf.avg_q_value = q_sum / count;
        buf.avg_score_lead = s_sum / count;
    }
    // exponentially discounted look-ahead means with three horizons (KataGo's short-term value targets)
    for (int i = 0; i < size; ++i) {
        auto& buf = training_buffer_[static_cast<size_t>(i)];
        const double lambda[3] = {1.0 / (1.0 + n * 0.18), 1.0 / (1.0 + n * 0.06), 1.0 / (1.0 + n * 0.02)};
        double q[3] = {0, 0, 0}, s[3] = {0, 0, 0}, gamma[3] = {1., 1., 1.};
        for (int h = 0; h < size; ++h) {
            const auto& cur = training_buffer_[static_cast<size_t>(std::min(i + h, size - 1))];
            const double sign = cur.side_to_move == buf.side_to_move ? 1 : -1;
            const double avg_q = cur.avg_q_value, avg_s = cur.avg_score_lead;
            for (int t = 0; t < 3; ++t) {
                q[t] += (1. - lambda[t]) * sign * (gamma[t] * avg_q);
                s[t] += (1. - lambda[t]) * sign * (gamma[t] * avg_s);
                gamma[t] *= lambda[t];
            }
        }
        buf.short_avg_q = static_cast<float>(q[0]);
        buf.middle_avg_q = static_cast<float>(q[1]);
        buf.long_avg_q = static_cast<float>(q[2]);
        buf.short_avg_score = static_cast<float>(s[0]);
        buf.middle_avg_score = static_cast<float>(s[1]);
        buf.long_avg_score = static_cast<float>(s[2]);
    }
    // auxiliary policy = the next position's policy target; the last one is a certain pass
    std::vector<float> aux(static_cast<size_t>(n + 1), 0);
    aux[static_cast<size_t>(n)] = 1.f;
    for (int i = size - 1; i >= 0; --i) {
        auto& buf = training_buffer_[static_cast<size_t>(i)];
        buf.auxiliary_probabilities = aux;
        aux = buf.probabilities;
    }
    for (auto& buf : training_buffer_) chunk.push_back(buf);
    training_buffer_.clear();
    single_candidate_records_.clear();
}

// ---------------------------------------------------------------------------------------------
void TrainingData::StreamOut(std::ostream& out) const {
    if (discard) return;
    auto floats = [&](const std::vector<float>& a) {
        for (size_t i = 0; i < a.size(); ++i) {
            out << a[i];
            if (i + 1 != a.size()) out << ' ';
        }
        out << std::endl;
    };
    out << version << std::endl << mode << std::endl;
    out << board_size << std::endl << komi << std::endl << rule << std::endl << wave << std::endl;
    // binary planes, four cells per hex digit (lowest bit first), a trailing 0/1 when the area is not a multiple of 4
    constexpr size_t kW = sayuri_host::PackedPlanes::kWords;
    const size_t saved = plane_bits.size() / kW;  // 43 - 6 planes
    const size_t spatial = static_cast<size_t>(board_size) * static_cast<size_t>(board_size);
    auto bit = [&](size_t p, size_t i) -> int { return static_cast<int>((plane_bits[p * kW + (i >> 5)] >> (i & 31)) & 1u); };
    for (size_t p = 0; p < saved; ++p) {
        for (size_t i = 0; i + 4 <= spatial; i += 4) {
            int hex = 0;
            for (size_t b = 0; b < 4; ++b) hex += bit(p, i + b) << b;
            out << std::hex << hex;
        }
        if (spatial % 4 != 0) out << static_cast<bool>(bit(p, spatial - 1));
        out << std::dec << std::endl;
    }
    out << (side_to_move == kBlack ? 1 : 0) << std::endl;
    floats(probabilities);
    floats(auxiliary_probabilities);
    for (int v : ownership) out << (v == 0 ? 0 : v == 1 ? 1 : 3);
    out << std::endl;
    out << result << std::endl;
    out << avg_q_value << ' ' << short_avg_q << ' ' << middle_avg_q << ' ' << long_avg_q << std::endl;
    out << final_score << std::endl;
    out << avg_score_lead << ' ' << short_avg_score << ' ' << middle_avg_score << ' ' << long_avg_score << std::endl;
    out << q_stddev << ' ' << score_stddev << std::endl;
    out << kld << std::endl;
}

} // namespace sayuri_engine
