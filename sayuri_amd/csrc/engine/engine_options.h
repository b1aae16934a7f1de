// Engine options as "key=value" text, with the names of the reference's option map (src/config.cc:21-133)
// so a settings file written for the reference reads the same here.
#pragma once

#include <string>
#include <vector>

#include "network.h"
#include "search_params.h"

namespace sayuri_engine {

struct SelfplayOptions {
    int num_games{0};
    int parallel_games{1};
    float komi_stddev{0.f};
    float komi_big_stddev{0.f};
    float komi_big_stddev_prob{0.f};
    float handicap_fair_komi_prob{0.f};
    float random_opening_prob{0.f};
    float random_opening_temp{1.f};
    int default_boardsize{19};
    float default_komi{7.5f};
    int scoring_rule{0};
    std::vector<std::string> selfplay_queries; // "bkp:19:7.5:0.2", "bhp:9:2:0.1", "srs:area:territory"
    std::string target_directory;
    std::uint64_t seed{0}; // 0 = from the clock
    // weights roll-over (reference Engine::SelectWeights / ShouldHalt, engine.cc:63-90): the loop winds down once the
    // newest file in weights_dir is no longer weights_file
    std::string weights_dir, weights_file;
    // extension (measurement only): the first game of worker g starts after g * stagger_moves / parallel_games
    // policy-sampled moves, so that a short window sees games in every phase, as a long-running self-play does
    int stagger_moves{0};
    // extension: how the concurrent games are scheduled (selfplay.cc): 0 = fibers on a few threads per usable core from 256
    // games on, one OS thread per game below; N > 0 = fibers on N threads; -1 = always one thread per game
    int game_threads{0};
    // extension (measurement only): how many finished games the data writer holds back while the workers run.  0 = the
    // reference's rule, `parallel_games` (pipe.cc:206-208: chunks leave in shuffled order out of a pool that large, so the
    // first `parallel_games` finished games are written only when as many more have finished or the run ends).  A short
    // measuring window sets it small, so that the writer's steady state -- one chunk out per finished game, what hours of
    // self-play see -- falls inside the window instead of behind it.
    int chunk_pool_games{0};
};

struct EngineOptions {
    SearchParams search;
    NetworkOptions network;
    SelfplayOptions selfplay;
    // Parses whitespace-separated key=value pairs; unknown keys throw std::invalid_argument.
    void Parse(const std::string& text);
};

} // namespace sayuri_engine
