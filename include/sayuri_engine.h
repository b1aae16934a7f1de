/* sayuri_engine.h -- C entry points of the host library (libsayuri_host.so): the weight loader, the
 * NetworkForwardPipe built on the HIP library, the Go engine (board / game state / input encoder), the
 * evaluation facade, the tree search and the self-play loop.
 *
 * Everything here sits ABOVE the device boundary of include/sayuri_hip.h and is what a non-C++ front end binds
 * (the Python face under sayuri_amd/ does, with ctypes).  A C++ program embeds the classes directly
 * (sayuri_amd/csrc/host/hip_forward_pipe.h, sayuri_amd/csrc/engine/{game_state,encoder,network,search,selfplay}.h).
 *
 * Conventions: handles are opaque pointers; functions returning int give 0 on success and -1 on failure with the
 * message in sayuri_host_last_error() / sayuri_engine_last_error(); moves are intersection indices (0..N-1 row
 * major, N = pass, -1 = resign); colours are 0 black, 1 white, 2 empty; options are "key=value key=value" text
 * with the names of the reference's option map (src/config.cc:21-133).
 */
#ifndef SAYURI_ENGINE_H
#define SAYURI_ENGINE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- weights (reference DNNLoader, src/neural/loader.cc:26-121,628-831) ------------------------------------ */
const char* sayuri_host_last_error(void);
void* sayuri_weights_load(const char* path);
void sayuri_weights_free(void* weights);
int sayuri_weights_info(void* weights, int* info12);
int sayuri_weights_block_info(void* weights, int block, int* binfo5);
long sayuri_weights_tensor(void* weights, const char* name, float* dst, long cap);

/* ---- forward pipe (reference NetworkForwardPipe / BatchForwardPipe, src/neural/network_basic.h:132-161,
 *      src/neural/batch_forward_pipe.cc:7-193) ---------------------------------------------------------------- */
void* sayuri_pipe_create(const char* weights_path, int board, int batch, int fp16, int device, int waittime_ms);
void sayuri_pipe_destroy(void* pipe);
int sayuri_pipe_num_workers(void* pipe);
void* sayuri_pipe_ctx(void* pipe, int gpu);                 /* the sayuri_hip_ctx of one GPU */
int sayuri_pipe_reconstruct(void* pipe, int board, int batch);
void* sayuri_pipe_raw(void* pipe);                          /* as NetworkForwardPipe*, for the engine entry points */
int sayuri_pipe_weights_version(void* pipe);
int sayuri_pipe_eval(void* pipe, int mode, int gpu, int n, const float* planes, const int* board_sizes,
                     const float* komi, const int* offsets, float* out);
int sayuri_pipe_netbench(void* pipe, int threads, double seconds, int board, double* evals_per_sec, long* total);
void sayuri_pipe_pump_times(void* pipe, double* out8, long* batches, long* evals);

/* ---- Go engine (reference GameState / Board / Encoder, src/game/game_state.h, src/game/board.h,
 *      src/neural/encoder.h:24-61) -------------------------------------------------------------------------- */
void* sayuri_go_new(int board, float komi, int scoring);
void* sayuri_go_clone(void* game);
void sayuri_go_free(void* game);
int sayuri_go_play(void* game, int move, int color);        /* color < 0: side to move; returns 1 if legal */
int sayuri_go_append(void* game, int move, int color);      /* set-up stone */
int sayuri_go_undo(void* game);
int sayuri_go_fixed_handicap(void* game, int stones);
void sayuri_go_set_komi(void* game, float komi);
void sayuri_go_set_rule(void* game, int scoring);
void sayuri_go_set_to_move(void* game, int color);
void sayuri_go_set_territory_helper_from_ownership(void* game);
void sayuri_go_freeze(void* game);
void sayuri_go_info(void* game, uint64_t* info16);
void sayuri_go_scalars(void* game, float* out6);
void sayuri_go_maps(void* game, uint8_t* out /* [9][N+1] */);
int sayuri_go_planes(void* game, int symmetry, int weights_version, float* planes /* [43|38][N] */);
/* the same planes in compact form (csrc/host/packed_planes.h): record = uint32 bits[binary][12] (bit y*bs+x of a 0/1 plane)
 * followed by 8 floats (one per broadcast plane); returns the number of binary planes (37, or 34 for v1/v2 nets) */
int sayuri_go_planes_packed(void* game, int symmetry, int weights_version, unsigned* record /* [binary*12 + 8] */);
/* measurement: seconds for `iters` encodings of the position; packed = 0: the 43 fp32 planes, 1: the compact record */
double sayuri_go_encode_seconds(void* game, int iters, int packed, int symmetry, int weights_version);
void sayuri_go_rng_stream(uint64_t seed, int n, uint32_t range, double prob, uint64_t* out /* [3n] */);

/* ---- evaluation facade and tree search (reference Network, src/neural/network.h:17-98; Search,
 *      src/mcts/search.h:155-296) --------------------------------------------------------------------------- */
const char* sayuri_engine_last_error(void);
void* sayuri_engine_net_new_pipe(void* raw_pipe, int weights_version, const char* options);
/* tests: a C forward function instead of a pipe (kind 0: fn(bs, komi, stm, offset, planes, out); kind 1:
 * fn(user, bs, komi, offset, planes, out)); fn == NULL selects the dummy random-output backend */
void* sayuri_engine_net_new_callback(void* forward_fn, int kind, const void* user, int weights_version, const char* options);
void sayuri_engine_net_free(void* net);
unsigned long sayuri_engine_net_queries(void* net);
void sayuri_engine_net_output(void* net, void* game, int ensemble, int symmetry, float temperature, int use_cache,
                              uint64_t seed, float* out /* [2N+9] */);
void* sayuri_engine_search_new(void* game, void* net, const char* options);
void sayuri_engine_search_free(void* search);
void sayuri_engine_search_seed(void* search, uint64_t caller_seed, uint64_t playout_seed);
void sayuri_engine_search_computation(void* search, void* game, int playouts, int tag, int* ints16, float* floats8,
                                      int* visits, float* estimated_q, float* target_policy, float* ownership);
int sayuri_engine_search_selfplay_move(void* search, void* game, int tag);
int sayuri_engine_search_think(void* search, void* game);
void sayuri_engine_search_update_territory_helper(void* search);
int sayuri_engine_search_single_candidate(void* search, int* record_indices, int cap);
long sayuri_engine_search_gather(void* search, char* text, long cap); /* the game's 53-line training records */

/* ---- self-play (reference SelfPlayPipe + Engine, src/selfplay/pipe.cc, src/selfplay/engine.cc) -------------- */
/* raw_pipe == NULL: dummy backend.  seconds > 0: time window; otherwise plays num_games complete games.
 * stats[10]: games_started, games_done, moves, playouts, nn_queries, cache_lookups, cache_hits, records, chunks, 0 */
int sayuri_selfplay_run(void* raw_pipe, int weights_version, const char* options, const char* name_suffix, double seconds,
                        int move_cap, uint64_t* stats, double* elapsed);
/* The same with the periodic exchange hook of the games-parallel multi-GPU path: every interval_seconds the thread
 * that called this function hands on_stats a snapshot (the same 10 counters, seconds since start) and this process's
 * own halt wish -- 1 once the newest file in the option weights_dir is no longer weights_file (reference
 * Engine::ShouldHalt, src/selfplay/engine.cc:88-90).  A non-zero return makes the loop wind down as the reference
 * does (src/selfplay/pipe.cc:246-258: max games = games in flight + 25, rounded up to 25).  stats (and the hook's
 * stats10) have TWENTY slots here: the ten of sayuri_selfplay_run, [9] = the final max-games value, [10] = sum of the
 * move numbers of the finished games, [11] = policy-sampled moves played by the stagger_moves option; the data writer
 * (reference SaveChunk + gzip, pipe.cc:116-159,181-233): [12] = chunks on disk when the window ended ([8] counts the
 * flush of the writer's pool behind it too), [13] = CPU nanoseconds of the writer thread, [14] = the same when the
 * window ended, [15] = bytes written to disk, [16] = bytes of record text before gzip, [17] = wall nanoseconds of the
 * final flush, [18..19] = 0.  The driver all-gathers the records inside the hook (sayuri_amd/shard.py). */
typedef int (*sayuri_selfplay_stats_fn)(const uint64_t* stats10, double elapsed, int local_halt, void* user);
int sayuri_selfplay_run_ex(void* raw_pipe, int weights_version, const char* options, const char* name_suffix, double seconds,
                           int move_cap, sayuri_selfplay_stats_fn on_stats, void* user, double interval_seconds,
                           uint64_t* stats, double* elapsed);

/* ---- search benchmark (reference --mode benchmark, src/benchmark/benchmark.cc:78-161) -------------------------------
 * `positions` policy-sampled openings, each searched once (options: playouts, default_boardsize, ...) with a fresh tree
 * and the evaluation cache off, `concurrent` searches at a time (this engine batches across searches, not inside a tree).
 * out8: playouts per move, playouts/s per search (the reference's figure), playouts/s of all searches together,
 * NN evals/s, wall seconds, KataGo's Elo estimate (benchmark.cc:14-28), NN queries, positions. */
int sayuri_engine_benchmark(void* raw_pipe, int weights_version, const char* options, int positions, int concurrent,
                            double* out8);

/* ---- M:N game scheduling self-test (csrc/host/fiber.h): `fibers` coroutines on `threads` OS threads, each waits
 * `rounds` times for its own word to change; returns fibers * rounds, or -1. */
long sayuri_fiber_selftest(int fibers, int threads, int rounds);

#ifdef __cplusplus
}
#endif
#endif /* SAYURI_ENGINE_H */
