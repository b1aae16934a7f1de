#include "engine_options.h"

#include <sstream>
#include <stdexcept>

namespace sayuri_engine {

namespace {
bool ToBool(const std::string& v) { return v == "1" || v == "true" || v == "on" || v == "yes"; }
} // namespace

void EngineOptions::Parse(const std::string& text) {
    std::istringstream in(text);
    std::string tok;
    while (in >> tok) {
        const size_t eq = tok.find('=');
        std::string k = tok.substr(0, eq), v = eq == std::string::npos ? "1" : tok.substr(eq + 1);
        for (auto& c : k)
            if (c == '-') c = '_';
        SearchParams& s = search;
#define OPT_INT(name, field) else if (k == name) field = std::stoi(v)
#define OPT_FLT(name, field) else if (k == name) field = std::stof(v)
#define OPT_DBL(name, field) else if (k == name) field = std::stod(v)
#define OPT_BOOL(name, field) else if (k == name) field = ToBool(v)
        if (k.empty()) continue;
        OPT_INT("threads", s.threads);
        OPT_INT("batch_size", s.batch_size);
        OPT_INT("playouts", s.playouts);
        OPT_INT("virtual_loss_count", s.virtual_loss_count);
        OPT_INT("random_min_visits", s.random_min_visits);
        OPT_FLT("random_min_ratio", s.random_min_ratio);
        OPT_FLT("random_moves_factor", s.random_moves_factor);
        OPT_FLT("random_moves_temp", s.random_moves_temp);
        OPT_FLT("resign_threshold", s.resign_threshold);
        OPT_FLT("lcb_reduction", s.lcb_reduction);
        OPT_FLT("fpu_reduction", s.fpu_reduction);
        OPT_FLT("root_fpu_reduction", s.root_fpu_reduction);
        OPT_FLT("cpuct_init", s.cpuct_init);
        OPT_FLT("cpuct_base_factor", s.cpuct_base_factor);
        OPT_FLT("cpuct_base", s.cpuct_base);
        OPT_BOOL("cpuct_dynamic", s.cpuct_dynamic);
        OPT_FLT("cpuct_dynamic_k_factor", s.cpuct_dynamic_k_factor);
        OPT_FLT("cpuct_dynamic_k_base", s.cpuct_dynamic_k_base);
        OPT_FLT("forced_playouts_k", s.forced_playouts_k);
        OPT_FLT("suppress_pass_factor", s.suppress_pass_factor);
        OPT_FLT("gumbel_c_visit", s.gumbel_c_visit);
        OPT_FLT("gumbel_c_scale", s.gumbel_c_scale);
        OPT_INT("gumbel_prom_visits", s.gumbel_prom_visits);
        OPT_INT("gumbel_considered_moves", s.gumbel_considered_moves);
        OPT_INT("gumbel_playouts_threshold", s.gumbel_playouts_threshold);
        OPT_BOOL("gumbel", s.gumbel);
        OPT_BOOL("always_completed_q_policy", s.always_completed_q_policy);
        OPT_BOOL("dirichlet_noise", s.dirichlet_noise);
        OPT_FLT("dirichlet_epsilon", s.dirichlet_epsilon);
        OPT_FLT("dirichlet_factor", s.dirichlet_factor);
        OPT_FLT("dirichlet_init", s.dirichlet_init);
        OPT_DBL("kldgain_per_node", s.kldgain_per_node);
        OPT_INT("kldgain_interval", s.kldgain_interval);
        OPT_FLT("score_utility_factor", s.score_utility_factor);
        OPT_FLT("score_utility_div", s.score_utility_div);
        OPT_FLT("root_policy_temp", s.root_policy_temp);
        OPT_FLT("policy_temp", s.policy_temp);
        OPT_INT("resign_playouts", s.resign_playouts);
        OPT_INT("fastsearch_playouts", s.fastsearch_playouts);
        OPT_FLT("fastsearch_playouts_prob", s.fastsearch_playouts_prob);
        OPT_FLT("random_fastsearch_prob", s.random_fastsearch_prob);
        OPT_FLT("resign_discard_prob", s.resign_discard_prob);
        OPT_BOOL("reuse_tree", s.reuse_tree);
        OPT_BOOL("friendly_pass", s.friendly_pass);
        OPT_BOOL("first_pass_bonus", s.first_pass_bonus);
        OPT_BOOL("symm_pruning", s.symm_pruning);
        OPT_BOOL("use_stm_winrate", s.use_stm_winrate);
        OPT_BOOL("capture_all_dead", s.capture_all_dead);
        OPT_FLT("ci_alpha", s.ci_alpha);
        OPT_BOOL("no_cache", network.no_cache);
        OPT_BOOL("early_symm_cache", network.early_symm_cache);
        OPT_BOOL("packed_inputs", network.packed_inputs);
        else if (k == "cache_memory_mib") network.cache_memory_mib = static_cast<size_t>(std::stol(v));
        else if (k == "policy_buffer_offset") network.default_policy_offset = static_cast<PolicyBufferOffset>(std::stoi(v));
        OPT_INT("num_games", selfplay.num_games);
        OPT_INT("parallel_games", selfplay.parallel_games);
        OPT_FLT("komi_stddev", selfplay.komi_stddev);
        OPT_FLT("komi_big_stddev", selfplay.komi_big_stddev);
        OPT_FLT("komi_big_stddev_prob", selfplay.komi_big_stddev_prob);
        OPT_FLT("handicap_fair_komi_prob", selfplay.handicap_fair_komi_prob);
        OPT_FLT("random_opening_prob", selfplay.random_opening_prob);
        OPT_FLT("random_opening_temp", selfplay.random_opening_temp);
        OPT_INT("defualt_boardsize", selfplay.default_boardsize); // sic: the reference's spelling (config.cc:40)
        OPT_INT("default_boardsize", selfplay.default_boardsize);
        OPT_FLT("defualt_komi", selfplay.default_komi);
        OPT_FLT("default_komi", selfplay.default_komi);
        OPT_INT("scoring_rule", selfplay.scoring_rule);
        else if (k == "selfplay_query") selfplay.selfplay_queries.push_back(v);
        else if (k == "target_directory") selfplay.target_directory = v;
        else if (k == "weights_dir") selfplay.weights_dir = v;
        else if (k == "weights_file") selfplay.weights_file = v;
        OPT_INT("stagger_moves", selfplay.stagger_moves);
        OPT_INT("game_threads", selfplay.game_threads);
        OPT_INT("chunk_pool_games", selfplay.chunk_pool_games);
        else if (k == "seed") selfplay.seed = std::stoull(v);
        else throw std::invalid_argument("unknown engine option: " + k);
#undef OPT_INT
#undef OPT_FLT
#undef OPT_DBL
#undef OPT_BOOL
    }
}

} // namespace sayuri_engine
