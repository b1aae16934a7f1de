// capi.cc -- flat C wrapper of the host library for ctypes (tests, bench.py, smoke()).
// Not part of the drop-in boundary (that is include/sayuri_hip.h + HipForwardPipe).
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "hip_forward_pipe.h"

using namespace sayuri_host;

namespace {
struct PipeHandle {
    std::shared_ptr<DNNWeights> weights;
    std::unique_ptr<HipForwardPipe> pipe;
};
thread_local std::string g_err;

std::vector<float>* FindTensor(DNNWeights& w, const std::string& name) {
    const auto dot = name.rfind('.');
    if (dot == std::string::npos) return nullptr;
    const std::string kind = name.substr(dot + 1);
    std::string path = name.substr(0, dot);
    ConvLayer* conv = nullptr;
    LinearLayer* fc = nullptr;
    if (path.rfind("tower.", 0) == 0) {
        const auto d2 = path.find('.', 6);
        if (d2 == std::string::npos) return nullptr;
        const int idx = std::stoi(path.substr(6, d2 - 6));
        if (idx < 0 || idx >= w.residual_blocks) return nullptr;
        BlockBasic& b = *w.tower[idx];
        const std::string l = path.substr(d2 + 1);
        if (l == "conv1") conv = &b.conv1;
        else if (l == "conv2") conv = &b.conv2;
        else if (l == "conv3") conv = &b.conv3;
        else if (l == "conv4") conv = &b.conv4;
        else if (l == "pre_btl_conv") conv = &b.pre_btl_conv;
        else if (l == "post_btl_conv") conv = &b.post_btl_conv;
        else if (l == "dw_conv") conv = &b.dw_conv;
        else if (l == "squeeze") fc = &b.squeeze;
        else if (l == "excite") fc = &b.excite;
    } else {
        if (path == "input_conv") conv = &w.input_conv;
        else if (path == "p_hd_conv") conv = &w.p_hd_conv;
        else if (path == "p_dw_conv") conv = &w.p_dw_conv;
        else if (path == "p_pt_conv") conv = &w.p_pt_conv;
        else if (path == "prob_conv") conv = &w.prob_conv;
        else if (path == "v_hd_conv") conv = &w.v_hd_conv;
        else if (path == "v_ownership") conv = &w.v_ownership;
        else if (path == "p_inter_fc") fc = &w.p_inter_fc;
        else if (path == "pass_fc") fc = &w.pass_fc;
        else if (path == "v_inter_fc") fc = &w.v_inter_fc;
        else if (path == "v_misc") fc = &w.v_misc;
    }
    if (conv) return kind == "w" ? &conv->GetWeights() : kind == "b" ? &conv->GetBiases() : nullptr;
    if (fc) return kind == "w" ? &fc->GetWeights() : kind == "b" ? &fc->GetBiases() : nullptr;
    return nullptr;
}
}  // namespace

extern "C" {

const char* sayuri_host_last_error() { return g_err.c_str(); }

// ---- weights only (no GPU needed)
void* sayuri_weights_load(const char* path) {
    auto w = std::make_unique<DNNWeights>();
    std::string err;
    if (!LoadWeightsFile(path, w.get(), &err)) {
        g_err = err;
        return nullptr;
    }
    return w.release();
}
void sayuri_weights_free(void* h) { delete static_cast<DNNWeights*>(h); }
int sayuri_weights_info(void* h, int* info) {
    auto* w = static_cast<DNNWeights*>(h);
    if (!w) return -1;
    info[0] = w->version; info[1] = w->input_channels; info[2] = w->residual_blocks; info[3] = w->residual_channels;
    info[4] = w->policy_head_channels; info[5] = w->value_head_channels; info[6] = w->probabilities_channels;
    info[7] = w->pass_probability_outputs; info[8] = w->ownership_channels; info[9] = w->value_misc_outputs;
    info[10] = static_cast<int>(w->default_act); info[11] = w->policy_head_type == PolicyHeadType::kRepLK ? 1 : 0;
    return 0;
}
int sayuri_weights_block_info(void* h, int idx, int* binfo) {
    auto* w = static_cast<DNNWeights*>(h);
    if (!w || idx < 0 || idx >= w->residual_blocks) return -1;
    const BlockBasic& b = *w->tower[idx];
    binfo[0] = static_cast<int>(b.type); binfo[1] = b.apply_se; binfo[2] = b.se_size;
    binfo[3] = b.bottleneck_channels; binfo[4] = b.feedforward_channels;
    return 0;
}
long sayuri_weights_tensor(void* h, const char* name, float* dst, long cap) {
    auto* w = static_cast<DNNWeights*>(h);
    if (!w) return -1;
    auto* v = FindTensor(*w, name);
    if (!v) return -1;
    const long n = static_cast<long>(v->size());
    if (dst) std::memcpy(dst, v->data(), sizeof(float) * static_cast<size_t>(n < cap ? n : cap));
    return n;
}

// ---- the pipe
void* sayuri_pipe_create(const char* weights_path, int board, int batch, int fp16, int device, int waittime_ms) {
    try {
        auto h = std::make_unique<PipeHandle>();
        h->weights = std::make_shared<DNNWeights>();
        std::string err;
        if (!LoadWeightsFile(weights_path, h->weights.get(), &err)) {
            g_err = err;
            return nullptr;
        }
        HipPipeConfig cfg;
        cfg.batch_size = batch;
        cfg.fp16 = fp16 != 0;
        cfg.default_boardsize = board;
        cfg.gpu_waittime_ms = waittime_ms;
        if (device >= 0) cfg.gpus = {device};
        h->pipe = std::make_unique<HipForwardPipe>(cfg);
        h->pipe->Initialize(h->weights);
        return h.release();
    } catch (const std::exception& e) {
        g_err = e.what();
        return nullptr;
    }
}

void sayuri_pipe_destroy(void* hp) {
    auto* h = static_cast<PipeHandle*>(hp);
    if (!h) return;
    try {
        h->pipe->Destroy();
    } catch (...) {
    }
    delete h;
}

void sayuri_pipe_pump_times(void* hp, double* out6, long* batches, long* evals) {
    auto* h = static_cast<PipeHandle*>(hp);
    h->pipe->pump_times(out6);
    *batches = static_cast<long>(h->pipe->num_batches());
    *evals = static_cast<long>(h->pipe->num_evals());
}
int sayuri_pipe_num_workers(void* hp) { return static_cast<PipeHandle*>(hp)->pipe->GetNumWorkers(); }
void* sayuri_pipe_ctx(void* hp, int gpu) { return static_cast<PipeHandle*>(hp)->pipe->ctx(gpu); }
int sayuri_pipe_reconstruct(void* hp, int board, int batch) {
    try {
        static_cast<PipeHandle*>(hp)->pipe->Construct(ForwardPipeOption::Get().SetBoardSize(board).SetBatchSize(batch), nullptr);
        return 0;
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    }
}

// netbench: the reference's own definition of "NN evals/s" (GTP `netbench`, src/game/gtp.cc:1468-1568):
// `threads` host threads hammer NetworkForwardPipe::Forward with cache off for `seconds`; the count
// includes queueing, staging, H2D / D2H.  (The reference also runs the encoder per call; the
// encoder is not part of this backend yet, so each thread re-submits a fixed random position.)
int sayuri_pipe_netbench(void* hp, int threads, double seconds, int board, double* evals_per_sec, long* total);

// n evaluations.  planes [n][43*361] in InputData layout (each sample packed with its OWN board
// stride), out [n][2*361 + 9] = prob[bs*bs], own[bs*bs] (both packed with the sample's stride,
// zero tail), then at offset 722: pass, wdl[3], stm, score, q_err, score_err, offset.
// mode 0: HipForwardPipe::BatchForward(gpu) after re-padding the inputs here the way
//         SendQueryAndWait would, outputs un-padded here; mode 1: n concurrent Forward() calls
//         through the queue (n threads); mode 2: the same through ForwardPacked() (packed_planes.h), mode 3: odd
//         requests packed, even ones fp32 (a mixed batch).
int sayuri_pipe_eval(void* hp, int mode, int gpu, int n, const float* planes, const int* board_sizes,
                     const float* komi, const int* offsets, float* out);
}

#include <thread>

// The pipe behind a handle as the abstract plugin interface (for the engine, which only knows
// NetworkForwardPipe), and the version of the weights it was built from.
extern "C" void* sayuri_pipe_raw(void* hp) {
    auto* h = static_cast<PipeHandle*>(hp);
    return h ? static_cast<NetworkForwardPipe*>(h->pipe.get()) : nullptr;
}
extern "C" int sayuri_pipe_weights_version(void* hp) {
    auto* h = static_cast<PipeHandle*>(hp);
    return h && h->weights ? h->weights->version : -1;
}

extern "C" int sayuri_pipe_eval(void* hp, int mode, int gpu, int n, const float* planes, const int* board_sizes,
                                const float* komi, const int* offsets, float* out) {
    auto* h = static_cast<PipeHandle*>(hp);
    if (!h || n <= 0) return -1;
    const int PL = kInputChannels * kNumIntersections, OL = 2 * kNumIntersections + 9;
    const int C = h->weights->input_channels;
    try {
        std::vector<InputData> inputs(n);
        for (int i = 0; i < n; ++i) {
            inputs[i].board_size = board_sizes[i];
            inputs[i].komi = komi ? komi[i] : 7.5f;
            inputs[i].offset = static_cast<PolicyBufferOffset>(offsets ? offsets[i] : 0);
            std::memcpy(inputs[i].planes.data(), planes + static_cast<size_t>(i) * PL, sizeof(float) * PL);
        }
        std::vector<OutputResult> outs(n);
        if (mode == 0) {
            const int B = h->pipe->board_size();
            std::vector<InputData> padded = inputs;
            for (int i = 0; i < n; ++i) {
                const int bs = inputs[i].board_size;
                if (bs == B) continue;
                padded[i].planes.fill(0.f);
                for (int c = 0; c < C; ++c)
                    for (int y = 0; y < bs; ++y)
                        for (int x = 0; x < bs; ++x)
                            padded[i].planes[(c * B + y) * B + x] = inputs[i].planes[(c * bs + y) * bs + x];
            }
            outs = h->pipe->BatchForward(gpu, padded);
            for (int i = 0; i < n; ++i) {
                const int bs = inputs[i].board_size;
                if (bs == B) continue;
                OutputResult r = outs[i];
                r.probabilities.fill(0.f);
                r.ownership.fill(0.f);
                for (int y = 0; y < bs; ++y)
                    for (int x = 0; x < bs; ++x) {
                        r.probabilities[y * bs + x] = outs[i].probabilities[y * B + x];
                        r.ownership[y * bs + x] = outs[i].ownership[y * B + x];
                    }
                outs[i] = r;
            }
        } else {
            // modes 2 / 3: every / every other request goes in as packed planes (packed_planes.h); the planes handed in
            // must be packable (0/1 binary planes, constant scalar planes), as every encoder output is
            const int nbin = PackedPlanes::BinaryPlanes(C);
            std::vector<PackedPlanes> packed(mode >= 2 ? n : 0);
            for (int i = 0; i < static_cast<int>(packed.size()); ++i) {
                PackedPlanes& pk = packed[i];
                const int bs = inputs[i].board_size, cells = bs * bs;
                pk.Clear(nbin);
                pk.board_size = bs;
                pk.komi = inputs[i].komi;
                pk.offset = static_cast<int>(inputs[i].offset);
                for (int c = 0; c < C; ++c) {
                    const float* pl = inputs[i].planes.data() + static_cast<size_t>(c) * cells;
                    if (c >= nbin) { pk.scalars[c - nbin] = pl[0]; continue; }
                    for (int k = 0; k < cells; ++k)
                        if (pl[k] != 0.f) pk.Set(c, k);
                }
            }
            std::vector<std::thread> th;
            std::vector<std::string> errs(n);
            for (int i = 0; i < n; ++i)
                th.emplace_back([&, i] {
                    try {
                        if (mode == 2 || (mode == 3 && (i & 1))) outs[i] = h->pipe->ForwardPacked(packed[i]);
                        else outs[i] = h->pipe->Forward(inputs[i]);
                    } catch (const std::exception& e) {
                        errs[i] = e.what();
                    }
                });
            for (auto& t : th) t.join();
            for (auto& e : errs)
                if (!e.empty()) throw std::runtime_error(e);
        }
        for (int i = 0; i < n; ++i) {
            float* o = out + static_cast<size_t>(i) * OL;
            const OutputResult& r = outs[i];
            std::memcpy(o, r.probabilities.data(), sizeof(float) * kNumIntersections);
            std::memcpy(o + kNumIntersections, r.ownership.data(), sizeof(float) * kNumIntersections);
            float* t = o + 2 * kNumIntersections;
            t[0] = r.pass_probability; t[1] = r.wdl[0]; t[2] = r.wdl[1]; t[3] = r.wdl[2];
            t[4] = r.stm_winrate; t[5] = r.final_score; t[6] = r.q_error; t[7] = r.score_error;
            t[8] = static_cast<float>(static_cast<int>(r.offset));
        }
        return 0;
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    }
}

#include <atomic>
#include <chrono>
#include <random>

extern "C" int sayuri_pipe_netbench(void* hp, int threads, double seconds, int board, double* evals_per_sec,
                                    long* total) {
    auto* h = static_cast<PipeHandle*>(hp);
    if (!h || threads <= 0 || seconds <= 0) return -1;
    try {
        std::atomic<long> count{0};
        std::atomic<bool> stop{false}, failed{false};
        std::string err;
        std::mutex err_mu;
        std::vector<std::thread> th;
        const int C = h->weights->input_channels;
        const auto t0 = std::chrono::steady_clock::now();
        for (int t = 0; t < threads; ++t)
            th.emplace_back([&, t] {
                auto in = std::make_unique<InputData>();
                std::mt19937 rng(1234 + t);
                in->board_size = board;
                in->komi = 7.5f;
                in->offset = PolicyBufferOffset::kNormal;
                const int s = board * board;
                for (int c = 0; c < C; ++c)
                    for (int i = 0; i < s; ++i) in->planes[c * s + i] = c < 37 ? float((rng() % 5) == 0) : (c == 42 ? 1.f : 0.3f);
                try {
                    while (!stop.load(std::memory_order_relaxed)) {
                        OutputResult r = h->pipe->Forward(*in);
                        (void)r;
                        count.fetch_add(1, std::memory_order_relaxed);
                    }
                } catch (const std::exception& e) {
                    std::lock_guard<std::mutex> lk(err_mu);
                    err = e.what();
                    failed.store(true);
                }
            });
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds && !failed.load())
            std::this_thread::sleep_for(std::chrono::milliseconds(5));
        const long n = count.load();
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        stop.store(true);
        for (auto& t : th) t.join();
        if (failed.load()) throw std::runtime_error(err);
        *evals_per_sec = n / dt;
        *total = n;
        return 0;
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    }
}

// Self-test of the fiber runtime (fiber.h), for tests/test_host_cpu.py: `fibers` coroutines on `threads` OS threads, each
// waits `rounds` times for its own word to change; a driver thread changes the words in a scrambled order and calls
// NotifyAll.  Returns the number of completed waits (fibers * rounds), or -1.
#include "fiber.h"
extern "C" long sayuri_fiber_selftest(int fibers, int threads, int rounds) {
    if (fibers <= 0 || threads <= 0 || rounds <= 0) return -1;
    std::vector<std::atomic<int>> words(static_cast<size_t>(fibers)), acks(static_cast<size_t>(fibers));
    for (auto& w : words) w.store(0);
    for (auto& a : acks) a.store(0);
    std::atomic<long> waits{0};
    std::atomic<int> finished{0};
    sayuri_fiber::FiberPool pool(64 << 10);
    for (int f = 0; f < fibers; ++f)
        pool.Add([&, f] {
            // some stack use and a value carried across every switch (a fiber may resume on another thread: FiberPool::Run)
            volatile char pad[2048];
            pad[0] = static_cast<char>(f);
            long mine = 0;
            for (int r = 0; r < rounds; ++r) {
                sayuri_fiber::WaitWhileEqual(&words[static_cast<size_t>(f)], r);  // until the driver has put r + 1 there
                if (words[static_cast<size_t>(f)].load(std::memory_order_acquire) != r + 1) return;  // resumed too early: counted as a failure
                ++mine;
                acks[static_cast<size_t>(f)].store(r + 1, std::memory_order_release);
            }
            if (pad[0] == static_cast<char>(f)) waits.fetch_add(mine);
            finished.fetch_add(1);
        });
    std::atomic<bool> give_up{false};
    std::thread driver([&] {
        std::mt19937 rng(7);
        const auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < rounds && !give_up.load(); ++r) {
            std::vector<int> order(static_cast<size_t>(fibers));
            for (int i = 0; i < fibers; ++i) order[static_cast<size_t>(i)] = i;
            std::shuffle(order.begin(), order.end(), rng);
            for (int i : order) {
                // the word of fiber i moves on only when the fiber has been through round r - 1: every round is a real suspension,
                // and the new value often lands while the fiber is still on its way out (the case a thief must not act on early)
                while (acks[static_cast<size_t>(i)].load(std::memory_order_acquire) != r) {
                    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) { give_up.store(true); break; }
                    std::this_thread::yield();
                }
                words[static_cast<size_t>(i)].store(r + 1, std::memory_order_release);
                if ((i & 7) == 0) sayuri_fiber::NotifyAll();
            }
            sayuri_fiber::NotifyAll();
        }
        if (give_up.load())  // let every fiber run out so that Run returns: the result reports the failure
            for (int i = 0; i < fibers; ++i) words[static_cast<size_t>(i)].store(-1, std::memory_order_release);
        sayuri_fiber::NotifyAll();
    });
    pool.Run(threads);
    driver.join();
    return finished.load() == fibers ? waits.load() : -1;
}
