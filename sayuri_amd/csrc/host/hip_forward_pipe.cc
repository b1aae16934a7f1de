// hip_forward_pipe.cc -- see hip_forward_pipe.h.
#include "hip_forward_pipe.h"
#include "fiber.h"

#include <pthread.h>
#include <algorithm>
#include <chrono>
#include <cstring>
#include <stdexcept>

#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <climits>

#include "../../../include/sayuri_hip.h"

SAYURI_HOST_BEGIN

#ifndef SAYURI_IN_TREE
std::string NetworkForwardPipe::GetName() const { return Valid() ? weights_->name : "random"; }
int NetworkForwardPipe::GetVersion() const { return Valid() ? weights_->version : -1; }
#endif

namespace {

// Completion is a per-request 32-bit flag; blocked callers sleep in the kernel on that word
// (futex) instead of polling, so hundreds of waiting search threads leave the host cores to
// the pump thread and the encoder.  (The reference parks each caller on its own heap-allocated
// mutex + condition variable, batch_forward_pipe.cc:35-46.)
static_assert(sizeof(std::atomic<int>) == sizeof(int), "futex needs a plain 32-bit word");
inline void FutexWait(std::atomic<int>* w, int expected) {
    syscall(SYS_futex, reinterpret_cast<int*>(w), FUTEX_WAIT_PRIVATE, expected, nullptr, nullptr, 0);
}
inline void FutexWakeAll(std::atomic<int>* w) {
    syscall(SYS_futex, reinterpret_cast<int*>(w), FUTEX_WAKE_PRIVATE, INT_MAX, nullptr, nullptr, 0);
}

[[noreturn]] void ThrowHip(const char* what) {
    throw std::runtime_error(std::string(what) + ": " + sayuri_hip_last_error());
}

void LoadConv(sayuri_hip_ctx* ctx, int id, ConvLayer& c) {
    if (sayuri_hip_load_tensor(ctx, id, SAYURI_T_WEIGHTS, c.GetWeights().data(), c.GetWeights().size()) ||
        sayuri_hip_load_tensor(ctx, id, SAYURI_T_BIASES, c.GetBiases().data(), c.GetBiases().size()))
        ThrowHip("sayuri_hip_load_tensor");
}
void LoadFc(sayuri_hip_ctx* ctx, int id, LinearLayer& f) {
    if (sayuri_hip_load_tensor(ctx, id, SAYURI_T_WEIGHTS, f.GetWeights().data(), f.GetWeights().size()) ||
        sayuri_hip_load_tensor(ctx, id, SAYURI_T_BIASES, f.GetBiases().data(), f.GetBiases().size()))
        ThrowHip("sayuri_hip_load_tensor");
}

int BlockTypeCode(BlockBasic& b) {  // the reference's Is*Block() are non-const
    if (b.IsResidualBlock()) return SAYURI_BLOCK_RESIDUAL;
    if (b.IsBottleneckBlock()) return SAYURI_BLOCK_BOTTLENECK;
    if (b.IsNestedBottleneckBlock()) return SAYURI_BLOCK_NESTED_BOTTLENECK;
    if (b.IsMixerBlock()) return SAYURI_BLOCK_MIXER;
    throw std::runtime_error("unknown block type in DNNWeights");
}

// Network description + tensors -> one device graph (NNGraph::ConstructGraph,
// cuda_forward_pipe.cc:133-613).
sayuri_hip_ctx* BuildCtx(int device, DNNWeights& w, int max_batch, int board, bool fp16) {
    std::vector<sayuri_hip_blockdesc> blocks(w.residual_blocks);
    for (int i = 0; i < w.residual_blocks; ++i) {
        BlockBasic& b = *w.tower[i];
        blocks[i].type = BlockTypeCode(b);
        blocks[i].apply_se = b.apply_se;
        blocks[i].se_size = b.se_size;
        blocks[i].bottleneck_channels = b.bottleneck_channels;
        blocks[i].feedforward_channels = b.feedforward_channels;
        blocks[i].dw_filter = b.IsMixerBlock() ? b.dw_conv.GetFilter() : 0;
    }
    sayuri_hip_netdesc d{};
    d.version = w.version;
    d.input_channels = w.input_channels;
    d.residual_channels = w.residual_channels;
    d.residual_blocks = w.residual_blocks;
    d.policy_head_channels = w.policy_head_channels;
    d.value_head_channels = w.value_head_channels;
    d.probabilities_channels = w.probabilities_channels;
    d.pass_probability_outputs = w.pass_probability_outputs;
    d.ownership_channels = w.ownership_channels;
    d.value_misc_outputs = w.value_misc_outputs;
    d.default_act = static_cast<int>(w.default_act);
    d.policy_head_type = w.policy_head_type == PolicyHeadType::kRepLK ? 1 : 0;
    d.policy_dw_filter = d.policy_head_type ? w.p_dw_conv.GetFilter() : 0;
    d.blocks = blocks.data();

    sayuri_hip_ctx* ctx = sayuri_hip_create(device, &d, max_batch, board, fp16 ? 1 : 0);
    if (!ctx) ThrowHip("sayuri_hip_create");
    try {
        LoadConv(ctx, SAYURI_L_INPUT_CONV, w.input_conv);
        for (int i = 0; i < w.residual_blocks; ++i) {
            BlockBasic& b = *w.tower[i];
            if (b.IsResidualBlock()) {
                LoadConv(ctx, SAYURI_L_BLOCK(i, SAYURI_S_CONV1), b.conv1);
                LoadConv(ctx, SAYURI_L_BLOCK(i, SAYURI_S_CONV2), b.conv2);
            } else if (b.IsBottleneckBlock() || b.IsNestedBottleneckBlock()) {
                LoadConv(ctx, SAYURI_L_BLOCK(i, SAYURI_S_PRE_BTL), b.pre_btl_conv);
                LoadConv(ctx, SAYURI_L_BLOCK(i, SAYURI_S_CONV1), b.conv1);
                LoadConv(ctx, SAYURI_L_BLOCK(i, SAYURI_S_CONV2), b.conv2);
                if (b.IsNestedBottleneckBlock()) {
                    LoadConv(ctx, SAYURI_L_BLOCK(i, SAYURI_S_CONV3), b.conv3);
                    LoadConv(ctx, SAYURI_L_BLOCK(i, SAYURI_S_CONV4), b.conv4);
                }
                LoadConv(ctx, SAYURI_L_BLOCK(i, SAYURI_S_POST_BTL), b.post_btl_conv);
            } else {
                LoadConv(ctx, SAYURI_L_BLOCK(i, SAYURI_S_DW_CONV), b.dw_conv);
                LoadConv(ctx, SAYURI_L_BLOCK(i, SAYURI_S_CONV1), b.conv1);
                LoadConv(ctx, SAYURI_L_BLOCK(i, SAYURI_S_CONV2), b.conv2);
            }
            if (b.apply_se) {
                LoadFc(ctx, SAYURI_L_BLOCK(i, SAYURI_S_SQUEEZE), b.squeeze);
                LoadFc(ctx, SAYURI_L_BLOCK(i, SAYURI_S_EXCITE), b.excite);
            }
        }
        LoadConv(ctx, SAYURI_L_P_HD_CONV, w.p_hd_conv);
        if (d.policy_head_type) {
            LoadConv(ctx, SAYURI_L_P_DW_CONV, w.p_dw_conv);
            LoadConv(ctx, SAYURI_L_P_PT_CONV, w.p_pt_conv);
        }
        LoadFc(ctx, SAYURI_L_P_INTER_FC, w.p_inter_fc);
        LoadConv(ctx, SAYURI_L_PROB_CONV, w.prob_conv);
        LoadFc(ctx, SAYURI_L_PASS_FC, w.pass_fc);
        LoadConv(ctx, SAYURI_L_V_HD_CONV, w.v_hd_conv);
        LoadFc(ctx, SAYURI_L_V_INTER_FC, w.v_inter_fc);
        LoadConv(ctx, SAYURI_L_V_OWNERSHIP, w.v_ownership);
        LoadFc(ctx, SAYURI_L_V_MISC, w.v_misc);
    } catch (...) {
        sayuri_hip_destroy(ctx);
        throw;
    }
    return ctx;
}

}  // namespace

HipForwardPipe::HipForwardPipe(HipPipeConfig cfg) : cfg_(std::move(cfg)) {}

HipForwardPipe::~HipForwardPipe() {
    try {
        Destroy();
    } catch (...) {
    }
}

bool HipForwardPipe::Valid() const { return weights_ != nullptr; }

sayuri_hip_ctx* HipForwardPipe::ctx(int gpu) const {
    return gpu >= 0 && gpu < static_cast<int>(graphs_.size()) ? graphs_[gpu]->ctx : nullptr;
}

void HipForwardPipe::Initialize(std::shared_ptr<DNNWeights> weights) {
    // cuda_forward_pipe.cc:14-25
    Construct(ForwardPipeOption::Get().SetBoardSize(cfg_.default_boardsize).SetBatchSize(cfg_.batch_size), weights);
}

void HipForwardPipe::Construct(ForwardPipeOption option, std::shared_ptr<DNNWeights> weights) {
    // cuda_forward_pipe.cc:44-119: rebuild only when the board changes or the batch grows.
    if (weights) weights_ = weights;
    if (weights_ == nullptr) return;  // Network falls back to its dummy backend
    int board = option.IsValidBoardSize() ? option.board_size : board_size_;
    int batch = option.IsValidBatchSize() ? option.batch_size : max_batch_;
    board = std::max(board, cfg_.fixed_nn_boardsize);
    if (board <= 0 || batch <= 0) return;
    if (board > kBoardSize) throw std::runtime_error("NN board size exceeds MAX_BOARD_SIZE");
    cfg_.batch_size = batch;  // forwarding size of the collector (SetForwardingSize)
    forward_size_.store(batch, std::memory_order_release);  // the pump and Reserve() read this one (the pump may be running)
    if (board_size_ == board && batch <= max_batch_ && !graphs_.empty() && !weights) return;
    Release();
    board_size_ = board;
    max_batch_ = batch;
    BuildGraphs();
}

void HipForwardPipe::BuildGraphs() {
    const int ndev = sayuri_hip_device_count();
    std::vector<int> devices;
    for (int g : cfg_.gpus)
        if (g >= 0 && g < ndev) devices.push_back(g);
    if (devices.empty())
        for (int i = 0; i < ndev; ++i) devices.push_back(i);
    if (devices.empty()) throw std::runtime_error("No executable GPU device!");

    const size_t B2 = static_cast<size_t>(board_size_) * board_size_;
    DNNWeights& w = *weights_;
    for (int dev : devices) {
        auto g = std::make_unique<Graph>();
        g->device = dev;
        g->ctx = BuildCtx(dev, w, max_batch_, board_size_, cfg_.fp16);
        auto pinned = [&](size_t count) {
            float* p = static_cast<float*>(sayuri_hip_host_alloc(sizeof(float) * count));
            if (!p) ThrowHip("sayuri_hip_host_alloc");
            return p;
        };
        for (Staging& s : g->st) {
            s.planes = pinned(static_cast<size_t>(max_batch_) * w.input_channels * B2);
            s.packed = reinterpret_cast<std::uint32_t*>(pinned(static_cast<size_t>(max_batch_) * PackedPlanes::RecordWords(BinaryPlanes())));
            s.is_packed.assign(max_batch_, 0);
            s.prob = pinned(static_cast<size_t>(max_batch_) * w.probabilities_channels * B2);
            s.pass = pinned(static_cast<size_t>(max_batch_) * w.pass_probability_outputs);
            s.misc = pinned(static_cast<size_t>(max_batch_) * w.value_misc_outputs);
            s.own = pinned(static_cast<size_t>(max_batch_) * B2);
            s.bsz.assign(max_batch_, board_size_);
            s.reqs.resize(max_batch_);
            s.fin_reqs.resize(max_batch_);
            s.fin_list.resize(max_batch_);
            s.fin_pos.resize(max_batch_);
        }
        graphs_.push_back(std::move(g));
    }
    trace_ = std::getenv("SAYURI_PIPE_TRACE") != nullptr;
    if (const char* e = std::getenv("SAYURI_PIPE_TAIL")) tail_frac_ = std::atof(e);
    if (const char* e = std::getenv("SAYURI_AB_ROTATE_NOTIFY")) rotate_notify_ = std::atoi(e) != 0;
    running_.store(true);
    for (auto& g : graphs_) g->pump = std::thread([this, gp = g.get()] { PumpLoop(gp); });
}

void HipForwardPipe::TraceDump() const {
    std::fprintf(stderr, "[pipe trace] arrivals after the last finished batch, 250 us bins:");
    for (const auto& a : trace_arrival_) std::fprintf(stderr, " %ld", a.load());
    std::fprintf(stderr, "\n[pipe trace] fibers, last finished batch -> the game runs again:");
    for (const auto& a : trace_resume_) std::fprintf(stderr, " %ld", a.load());
    std::fprintf(stderr, "\n[pipe trace] fibers, the game runs again -> its next request:");
    for (const auto& a : trace_think_) std::fprintf(stderr, " %ld", a.load());
    std::fprintf(stderr, "\n[pipe trace] tail rule (SAYURI_PIPE_TAIL), age of the running batch at the close:");
    for (const auto& a : trace_tail_) std::fprintf(stderr, " %ld", a.load());
    std::fprintf(stderr, "\n[pipe trace] batch sizes /16:");
    for (const auto& a : trace_size_) std::fprintf(stderr, " %ld", a.load());
    std::fprintf(stderr, "\n[pipe trace] closed: full %ld, idle-wait %ld, 85%%-rule %ld, stray %ld\n", trace_reason_[0].load(),
                 trace_reason_[1].load(), trace_reason_[2].load(), trace_reason_[3].load());
}

void HipForwardPipe::DestroyGraphs() {
    if (trace_ && !graphs_.empty()) TraceDump();
    running_.store(false);
    for (auto& g : graphs_) {
        {
            std::lock_guard<std::mutex> lk(g->mu);
        }
        g->cv.notify_all();
        g->epoch.fetch_add(1, std::memory_order_release);
        FutexWakeAll(&g->epoch);  // callers parked for a free staging set see running_ == false
    }
    for (auto& g : graphs_)
        if (g->pump.joinable()) g->pump.join();
    for (auto& g : graphs_) {
        for (Staging& s : g->st) {
            sayuri_hip_host_free(s.planes);
            sayuri_hip_host_free(s.packed);
            sayuri_hip_host_free(s.prob);
            sayuri_hip_host_free(s.pass);
            sayuri_hip_host_free(s.misc);
            sayuri_hip_host_free(s.own);
        }
        if (g->ctx) sayuri_hip_destroy(g->ctx);
    }
    graphs_.clear();
}

void HipForwardPipe::Release() { DestroyGraphs(); }

void HipForwardPipe::Destroy() { DestroyGraphs(); }

// Copy one request into staging slot `slot`, re-padding a smaller board top-left into the NN
// grid (what SendQueryAndWait does with a temporary InputData, batch_forward_pipe.cc:15-33).
void HipForwardPipe::StageInput(Staging* st, int slot, const InputData& in, bool already_padded) {
    const int B = board_size_, bs = in.board_size, C = weights_->input_channels;
    if (bs < 2 || bs > B) throw std::runtime_error("InputData board size does not fit the NN board");
    float* dst = st->planes + static_cast<size_t>(slot) * C * B * B;
    st->bsz[slot] = bs;
    if (bs == B || already_padded) {
        std::memcpy(dst, in.planes.data(), sizeof(float) * C * B * B);
        return;
    }
    std::memset(dst, 0, sizeof(float) * C * B * B);
    for (int c = 0; c < C; ++c)
        for (int y = 0; y < bs; ++y)
            std::memcpy(dst + (static_cast<size_t>(c) * B + y) * B, in.planes.data() + (static_cast<size_t>(c) * bs + y) * bs,
                        sizeof(float) * bs);
}

int HipForwardPipe::BinaryPlanes() const { return PackedPlanes::BinaryPlanes(weights_->input_channels); }

// The packed flavour: the record goes into the pinned packed buffer as it is -- the bits are in the sample's own cell
// order, the device kernel knows the sample's board size, nothing is re-padded.
void HipForwardPipe::StagePacked(Staging* st, int slot, const PackedPlanes& in) {
    if (in.board_size < 2 || in.board_size > board_size_) throw std::runtime_error("PackedPlanes board size does not fit the NN board");
    if (in.binary_planes != BinaryPlanes()) throw std::runtime_error("PackedPlanes: binary plane count does not match the network");
    st->bsz[slot] = in.board_size;
    in.Store(st->packed + static_cast<size_t>(slot) * PackedPlanes::RecordWords(in.binary_planes));
}

// A packed slot of a batch that also holds fp32 requests: expand it into the fp32 staging (NN grid), pump thread.
void HipForwardPipe::ExpandPacked(Staging* st, int slot) {
    const int B = board_size_, bs = st->bsz[slot], C = weights_->input_channels, nbin = BinaryPlanes();
    const std::uint32_t* rec = st->packed + static_cast<size_t>(slot) * PackedPlanes::RecordWords(nbin);
    float* dst = st->planes + static_cast<size_t>(slot) * C * B * B;
    std::memset(dst, 0, sizeof(float) * C * B * B);
    for (int c = 0; c < C; ++c) {
        float scalar = 0.f;
        if (c >= nbin) std::memcpy(&scalar, rec + nbin * PackedPlanes::kWords + (c - nbin), sizeof scalar);
        for (int y = 0; y < bs; ++y)
            for (int x = 0; x < bs; ++x) {
                const int cell = y * bs + x;
                dst[(static_cast<size_t>(c) * B + y) * B + x] =
                    c < nbin ? static_cast<float>((rec[c * PackedPlanes::kWords + (cell >> 5)] >> (cell & 31)) & 1u) : scalar;
            }
    }
}

// FillOutputs of the CPU pipe (blas_forward_pipe.cc:565-619 -- the oracle; the CUDA pipe's
// pass[0] for every offset, cuda_forward_pipe.cc:1074, is a known discrepancy) fused with the
// un-padding of SendQueryAndWait (batch_forward_pipe.cc:48-68).
void HipForwardPipe::FillOutput(const Staging* g, int slot, const Echo& in, bool unpad, OutputResult* out) const {
    DNNWeights& w = *weights_;
    const int B = board_size_, B2 = B * B, bs = in.board_size;
    const bool v1 = w.version <= 2;  // Encoder::GetEncoderVersion, encoder.h:64-77
    int offset = v1 ? 0 : static_cast<int>(in.offset);
    if (offset < 0 || offset >= w.probabilities_channels) offset = 0;
    const float* prob = g->prob + (static_cast<size_t>(slot) * w.probabilities_channels + offset) * B2;
    const float* own = g->own + static_cast<size_t>(slot) * B2;
    const float* misc = g->misc + static_cast<size_t>(slot) * w.value_misc_outputs;
    const float* pass = g->pass + static_cast<size_t>(slot) * w.pass_probability_outputs;
    if (unpad && bs != B) {
        for (int y = 0; y < bs; ++y)
            for (int x = 0; x < bs; ++x) {
                out->probabilities[y * bs + x] = prob[y * B + x];
                out->ownership[y * bs + x] = own[y * B + x];
            }
    } else {
        std::copy(prob, prob + B2, out->probabilities.begin());
        std::copy(own, own + B2, out->ownership.begin());
    }
    out->pass_probability = pass[offset];
    out->wdl[0] = misc[0];
    out->wdl[1] = misc[1];
    out->wdl[2] = misc[2];
    out->stm_winrate = misc[3];
    if (v1) {
        out->final_score = misc[4];
        out->q_error = 0.f;
        out->score_error = 0.f;
        out->offset = PolicyBufferOffset::kNormal;
    } else {
        out->final_score = misc[8];
        out->q_error = misc[13];
        out->score_error = misc[14];
        out->offset = in.offset;
    }
    out->board_size = bs;
    out->komi = in.komi;
    out->fp16 = cfg_.fp16;
}

void HipForwardPipe::SubmitBatch(Graph* g, Staging* s, int n) {
    const auto t0 = std::chrono::steady_clock::now();
    int npacked = 0;
    for (int i = 0; i < n; ++i) npacked += s->is_packed[i];
    if (npacked > 0 && npacked < n)  // a mixed batch travels as fp32 planes
        for (int i = 0; i < n; ++i)
            if (s->is_packed[i]) ExpandPacked(s, i);
    std::lock_guard<std::mutex> dev(g->dev_mu);
    if (npacked == n) {
        if (sayuri_hip_submit_packed(g->ctx, n, s->packed, BinaryPlanes(), s->bsz.data(), s->prob, s->pass, s->misc, s->own, &s->ticket))
            ThrowHip("sayuri_hip_submit_packed");
    } else if (sayuri_hip_submit(g->ctx, n, s->planes, s->bsz.data(), s->prob, s->pass, s->misc, s->own, &s->ticket)) {
        ThrowHip("sayuri_hip_submit");
    }
    s->n_inflight = n;
    pump_ns_[0] += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
}

void HipForwardPipe::FinishBatch(Graph* g, Staging* s, int n) {
    const auto t0 = std::chrono::steady_clock::now();
    int rc;
    {
        std::lock_guard<std::mutex> dev(g->dev_mu);
        rc = sayuri_hip_wait(g->ctx, s->ticket);
    }
    const auto t1 = std::chrono::steady_clock::now();
    // the previous batch of this set was handed out at least one batch time ago
    while (s->wakes_done.load(std::memory_order_acquire) < s->fin_count) std::this_thread::yield();
    int count = 0, fibers = 0;
    for (int i = 0; i < n; ++i) {
        const Request& r = s->reqs[i];
        s->fin_reqs[i] = r;
        if (r.fiber) {
            ++fibers;  // flagged below, once fin_status is in place
        } else if (r.self_serve) {
            s->fin_pos[i] = count;
            s->fin_list[count++] = i;
        } else {  // asynchronous Submit(): filled and signalled here
            if (rc == 0) FillOutput(s, i, r.echo, true, r.output);
            r.done->store(rc == 0 ? 1 : -1, std::memory_order_release);
            FutexWakeAll(r.done);
        }
    }
    batches_.fetch_add(1, std::memory_order_relaxed);
    evals_.fetch_add(static_cast<size_t>(n), std::memory_order_relaxed);
    s->n_inflight = 0;
    s->fin_count = count;
    s->fin_fibers = fibers;
    s->wakes_done.store(0, std::memory_order_relaxed);
    s->consumed.store(0, std::memory_order_relaxed);
    s->fin_status.store(rc == 0 ? 1 : -1, std::memory_order_relaxed);
    if (fibers > 0) {  // games scheduled as fibers: one flag store each, one wake word for their scheduler threads
        for (int i = 0; i < n; ++i)
            if (s->fin_reqs[i].fiber) s->fin_reqs[i].done->store(rc == 0 ? 1 : -1, std::memory_order_release);
        sayuri_fiber::NotifyAll();
    }
    if (count > 0) {  // root of the wake tree
        std::atomic<int>* root = s->fin_reqs[s->fin_list[0]].done;
        root->store(1, std::memory_order_release);
        FutexWakeAll(root);
    }
    // The set takes new requests at once: its inputs are on the GPU already, the snapshot above keeps the hand-out
    // independent of new reservations, and the pinned OUTPUT buffers are not written again before the set's next
    // batch is submitted -- SubmitBatch waits for `consumed` there (callers need microseconds, that is a batch away).
    Reopen(g, s);
    const auto t2 = std::chrono::steady_clock::now();
    pump_ns_[3] += std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count();
    pump_ns_[1] += std::chrono::duration_cast<std::chrono::nanoseconds>(t2 - t1).count();
}

void HipForwardPipe::Reopen(Graph* g, Staging* s) {
    s->ready.store(0, std::memory_order_relaxed);
    s->reserved.store(0, std::memory_order_release);  // re-open for callers
    g->epoch.fetch_add(1, std::memory_order_seq_cst);
    FutexWakeAll(&g->epoch);
    if (epoch_parked_.load(std::memory_order_seq_cst) > 0) sayuri_fiber::NotifyAll();  // fibers parked for a free staging set (wake_callers)
}

// One persistent pump per GPU over a ring of staging sets.  Callers fill the set `fill` points at; when it holds
// batch_size requests (or its wait expired) the pump closes it and points callers at the next set of the ring.
// Closed sets are ENQUEUED on the GPU in order, at most two at a time (H2D, graph, D2H are stream-ordered, so
// the second runs back to back with the first); the pump hands out a batch's results when its event fires and
// re-opens the set.  With more leaves in flight than two batches the remaining sets hold full batches that are
// ready the moment the GPU frees a slot.  A set that is not full is sent after gpu_waittime_ms, and once that
// happened the pump stops waiting until the fill set runs dry again (the adaptive 0 <-> base wait of
// batch_forward_pipe.cc:99-193).
void HipForwardPipe::PumpLoop(Graph* g) {
    pthread_setname_np(pthread_self(), "sayuri-pump");
    using clock = std::chrono::steady_clock;
    constexpr int K = Graph::kSets;
    auto count = [](const Staging& s) { return s.reserved.load(std::memory_order_acquire) & ~Staging::kClosed; };
    auto closed = [](const Staging& s) { return (s.reserved.load(std::memory_order_acquire) & Staging::kClosed) != 0; };
    // forwarding size, read live: Construct() may lower it while the pump runs (no rebuild)
    auto want_now = [this] { return static_cast<unsigned>(std::min(forward_size_.load(std::memory_order_acquire), max_batch_)); };
    int cur = 0;              // set currently filling
    int pending[K], n_pending = 0, pending_n[K];  // closed sets (and their sizes) waiting for a GPU slot, FIFO
    int inflight[2], n_in = 0;                    // FIFO of sets on the GPU
    bool timing = false;
    clock::time_point first_seen{};
    clock::time_point gpu_busy_since{};  // when the batch at the head of the GPU queue started executing (estimate)
    double gpu_batch_us = 1000.0;        // running mean of a batch's time on the GPU
    clock::time_point gpu_idle_since{};
    bool ever_busy = false;

    auto wake_callers = [&] {
        const auto t0 = clock::now();
        g->epoch.fetch_add(1, std::memory_order_seq_cst);
        FutexWakeAll(&g->epoch);
        // fibers parked in Reserve() on the epoch word are woken by their scheduler threads, which may be asleep themselves
        // -- only when a fiber IS parked there (with four staging sets that is rare): a NotifyAll wakes every scheduler thread,
        // and one per rotation was half of all wake-ups of a self-play rank (futex calls: 11 % of its host time)
        if (rotate_notify_ && epoch_parked_.load(std::memory_order_seq_cst) > 0) sayuri_fiber::NotifyAll();
        pump_ns_[4] += std::chrono::duration_cast<std::chrono::nanoseconds>(clock::now() - t0).count();
    };
    // close the fill set and point callers at the next one of the ring
    auto close_and_rotate = [&](int reason) {
        Staging& s = g->st[cur];
        const unsigned prev = s.reserved.fetch_or(Staging::kClosed, std::memory_order_acq_rel);
        const int n = static_cast<int>(std::min<unsigned>(prev & ~Staging::kClosed, static_cast<unsigned>(max_batch_)));
        if (n == 0) {
            s.reserved.store(0, std::memory_order_release);
            return;
        }
        if (static_cast<unsigned>(n) < want_now()) pump_ns_[5] += 1000;  // counts partial batches (reported /1000)
        if (trace_) {
            trace_size_[std::min(n / 16, 16)].fetch_add(1, std::memory_order_relaxed);
            trace_reason_[reason].fetch_add(1, std::memory_order_relaxed);
        }
        pending[n_pending] = cur;
        pending_n[n_pending++] = n;
        cur = (cur + 1) % K;
        g->fill.store(cur, std::memory_order_release);
        wake_callers();
    };
    // A caller that read `fill`, was preempted, and resumed after that set had been closed, evaluated and re-opened
    // finds `reserved` == 0 there and takes a slot in a set that is no longer the fill set.  Such a set is open, not
    // `cur`, and holds requests: close it where it stands and queue it like any other batch (the fill index does not
    // move).  Without this the request would wait until the ring wraps around to its set -- forever once traffic stops.
    auto close_strays = [&] {
        bool any = false;
        for (int k = 0; k < K && n_pending < K; ++k) {
            if (k == cur) continue;
            Staging& s = g->st[k];
            const unsigned seen = s.reserved.load(std::memory_order_acquire);
            if ((seen & Staging::kClosed) || seen == 0) continue;
            const unsigned prev = s.reserved.fetch_or(Staging::kClosed, std::memory_order_acq_rel);
            const int n = static_cast<int>(std::min<unsigned>(prev & ~Staging::kClosed, static_cast<unsigned>(max_batch_)));
            if (n == 0) { s.reserved.store(0, std::memory_order_release); continue; }
            pump_ns_[5] += 1000;
            if (trace_) trace_reason_[3].fetch_add(1, std::memory_order_relaxed);
            pending[n_pending] = k;
            pending_n[n_pending++] = n;
            any = true;
        }
        return any;
    };
    // wait for the callers' plane copies of the oldest pending set, then enqueue it
    auto submit_pending = [&] {
        const int i = pending[0], n = pending_n[0];
        for (int k = 1; k < n_pending; ++k) {
            pending[k - 1] = pending[k];
            pending_n[k - 1] = pending_n[k];
        }
        --n_pending;
        Staging& s = g->st[i];
        const auto tc0 = clock::now();
        while (s.ready.load(std::memory_order_acquire) < static_cast<unsigned>(n)) std::this_thread::yield();
        // every blocking caller of this set's previous batch has copied its result out of the pinned outputs
        while (s.consumed.load(std::memory_order_acquire) < s.fin_count + s.fin_fibers) std::this_thread::yield();
        pump_ns_[7] += std::chrono::duration_cast<std::chrono::nanoseconds>(clock::now() - tc0).count();
        try {
            SubmitBatch(g, &s, n);
            if (n_in == 0) {
                gpu_busy_since = clock::now();
                if (ever_busy) pump_ns_[6] += std::chrono::duration_cast<std::chrono::nanoseconds>(gpu_busy_since - gpu_idle_since).count();
                ever_busy = true;
            }
            inflight[n_in++] = i;
        } catch (const std::exception&) {
            // nothing was evaluated: every caller gets the failure directly (no tree: nothing to copy out)
            for (int k = 0; k < n; ++k) {
                s.reqs[k].done->store(-1, std::memory_order_release);
                FutexWakeAll(s.reqs[k].done);
            }
            sayuri_fiber::NotifyAll();
            Reopen(g, &s);
        }
    };
    auto finish_oldest = [&] {
        Staging& s = g->st[inflight[0]];
        if (trace_) trace_last_done_ns_.store(std::chrono::duration_cast<std::chrono::nanoseconds>(clock::now().time_since_epoch()).count(), std::memory_order_relaxed);
        FinishBatch(g, &s, s.n_inflight);
        const auto now = clock::now();
        const double us = std::chrono::duration<double, std::micro>(now - gpu_busy_since).count();
        gpu_batch_us = 0.8 * gpu_batch_us + 0.2 * us;
        gpu_busy_since = now;  // the next queued batch (if any) has the GPU from here
        inflight[0] = inflight[1];
        --n_in;
        if (n_in == 0) gpu_idle_since = now;
    };

    while (true) {
        if (close_strays()) continue;
        if (!running_.load() && n_in == 0 && n_pending == 0) {
            bool idle = true;
            for (const Staging& s : g->st) idle &= count(s) == 0;
            if (idle) return;
        }
        // keep the GPU fed: closed sets go out while it has a free slot
        if (n_pending > 0 && n_in < 2) {
            submit_pending();
            continue;
        }
        Staging& s = g->st[cur];
        if (closed(s)) {  // the ring is full: wait for the oldest batch, its set is the one callers are parked on
            if (n_in > 0) finish_oldest();
            timing = false;
            continue;
        }
        const unsigned c = count(s);
        if (c >= want_now() || (c > 0 && (cfg_.gpu_waittime_ms <= 0 || !running_.load()))) {
            close_and_rotate(0);
            timing = false;
            continue;
        }
        if (c > 0 && n_in == 0) {
            // partial batch and an idle GPU: give stragglers gpu_waittime_ms, then send what is there
            if (!timing) { timing = true; first_seen = clock::now(); }
            if (clock::now() - first_seen >= std::chrono::milliseconds(cfg_.gpu_waittime_ms)) {
                close_and_rotate(1);
                timing = false;
                continue;
            }
        } else if (c > 0 && n_in == 1) {
            // one batch is running and nothing is queued behind it: let the partial set keep filling, but enqueue it
            // shortly before that batch is expected to finish, so its upload hides under the running batch's tail
            timing = false;
            const double run_us = std::chrono::duration<double, std::micro>(clock::now() - gpu_busy_since).count();
            if (run_us >= tail_frac_ * gpu_batch_us) {
                if (trace_) trace_tail_[std::min(static_cast<int>(run_us / 250.0), 31)].fetch_add(1, std::memory_order_relaxed);
                close_and_rotate(2);
                continue;
            }
        } else {
            timing = false;
        }
        // nothing to send yet: retire a finished batch if there is one, else nap
        if (n_in > 0) {
            int q;
            {
                std::lock_guard<std::mutex> dev(g->dev_mu);
                q = sayuri_hip_query(g->ctx, g->st[inflight[0]].ticket);
            }
            if (q != 0) {
                finish_oldest();
                continue;
            }
        }
        const auto tw0 = clock::now();
        {
            std::unique_lock<std::mutex> lk(g->mu);
            g->cv.wait_for(lk, std::chrono::microseconds(n_in > 0 ? 50 : 200));
        }
        pump_ns_[2] += std::chrono::duration_cast<std::chrono::nanoseconds>(clock::now() - tw0).count();
    }
}

HipForwardPipe::Ticket HipForwardPipe::Reserve(const InputData* input, const PackedPlanes* packed, OutputResult* out,
                                               std::atomic<int>* done, bool self_serve, bool fiber) {
    if (graphs_.empty()) throw std::runtime_error("HipForwardPipe is not constructed");
    const Echo echo = input ? Echo{input->board_size, input->offset, input->komi}
                            : Echo{packed->board_size, static_cast<PolicyBufferOffset>(packed->offset), packed->komi};
    if (echo.board_size < 2 || echo.board_size > board_size_)
        throw std::runtime_error("InputData board size does not fit the NN board");
    done->store(0, std::memory_order_relaxed);
    if (trace_) {
        const long long now = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
        const long long d = now - trace_last_done_ns_.load(std::memory_order_relaxed);
        trace_arrival_[std::min<long long>(std::max<long long>(d, 0) / 250000, 31)].fetch_add(1, std::memory_order_relaxed);
        if (long long* st = sayuri_fiber::FiberStamp()) {
            if (*st) trace_think_[std::min<long long>(std::max<long long>(now - *st, 0) / 250000, 31)].fetch_add(1, std::memory_order_relaxed);
        }
    }
    Graph* g = graphs_[next_graph_.fetch_add(1, std::memory_order_relaxed) % graphs_.size()].get();
    const unsigned cap = static_cast<unsigned>(max_batch_);
    const unsigned want = static_cast<unsigned>(std::min(forward_size_.load(std::memory_order_acquire), max_batch_));
    for (;;) {
        const int epoch = g->epoch.load(std::memory_order_acquire);
        Staging& s = g->st[g->fill.load(std::memory_order_acquire)];
        const unsigned seen = s.reserved.load(std::memory_order_acquire);
        if (!(seen & Staging::kClosed) && seen < cap) {
            const unsigned r = s.reserved.fetch_add(1, std::memory_order_acq_rel);
            if (!(r & Staging::kClosed) && r < cap) {
                const int slot = static_cast<int>(r);
                // the one copy of the planes, by the calling thread
                if (input) StageInput(&s, slot, *input, false);
                else StagePacked(&s, slot, *packed);
                s.is_packed[slot] = input ? 0 : 1;
                s.reqs[slot] = Request{echo, out, done, self_serve, fiber};
                s.ready.fetch_add(1, std::memory_order_release);
                if (r == 0 || r + 1 >= want) g->cv.notify_one();
                return Ticket{g, &s, slot};
            }
        }
        // both sets are taken (one on the GPU, one full or being rotated): sleep until the pump re-opens one
        if (!running_.load()) throw std::runtime_error("HipForwardPipe is shutting down");
        if (fiber) {
            // let the thread's other games run meanwhile.  The count is raised BEFORE the word is looked at again (inside
            // WaitWhileEqual) and the pump looks at the count AFTER it changed the word: one of the two sees the other.
            epoch_parked_.fetch_add(1, std::memory_order_seq_cst);
            sayuri_fiber::WaitWhileEqual(&g->epoch, epoch);
            epoch_parked_.fetch_sub(1, std::memory_order_seq_cst);
        } else {
            FutexWait(&g->epoch, epoch);
        }
    }
}

void HipForwardPipe::Submit(const InputData& input, OutputResult* out, std::atomic<int>* done) {
    Reserve(&input, nullptr, out, done, false);
}

OutputResult HipForwardPipe::Forward(const InputData& input) { return ForwardAny(&input, nullptr); }

OutputResult HipForwardPipe::ForwardPacked(const PackedPlanes& input) { return ForwardAny(nullptr, &input); }

OutputResult HipForwardPipe::ForwardAny(const InputData* in, const PackedPlanes* pk) {
    OutputResult out;
    std::atomic<int> done{0};
    const Echo input = in ? Echo{in->board_size, in->offset, in->komi}
                          : Echo{pk->board_size, static_cast<PolicyBufferOffset>(pk->offset), pk->komi};
    if (sayuri_fiber::InFiber()) {
        fibers_seen_.store(true, std::memory_order_relaxed);
        // M:N game scheduling (fiber.h): hand the request over and run this thread's other games until the batch is back
        const Ticket t = Reserve(in, pk, nullptr, &done, false, true);
        sayuri_fiber::WaitWhileEqual(&done, 0);
        if (trace_) {
            const long long now = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
            const long long d = now - trace_last_done_ns_.load(std::memory_order_relaxed);
            trace_resume_[std::min<long long>(std::max<long long>(d, 0) / 250000, 31)].fetch_add(1, std::memory_order_relaxed);
            if (long long* st = sayuri_fiber::FiberStamp()) *st = now;
        }
        Staging& s = *t.s;
        const int status = done.load(std::memory_order_acquire);
        if (status > 0) FillOutput(&s, t.slot, input, true, &out);
        s.consumed.fetch_add(1, std::memory_order_release);
        if (status < 0) throw std::runtime_error("HIP forward pipe failed while evaluating a batch");
        return out;
    }
    const Ticket t = Reserve(in, pk, nullptr, &done, true);
    int st;
    while ((st = done.load(std::memory_order_acquire)) == 0) FutexWait(&done, 0);
    if (st < 0) throw std::runtime_error("HIP forward pipe failed while evaluating a batch");
    // woken through the batch's tree: pass the wake-up on to this node's children first, then take the result
    Staging& s = *t.s;
    const int count = s.fin_count, pos = s.fin_pos[t.slot];
    for (int c = Staging::kFanout * pos + 1; c <= Staging::kFanout * pos + Staging::kFanout && c < count; ++c) {
        std::atomic<int>* child = s.fin_reqs[s.fin_list[c]].done;
        child->store(1, std::memory_order_release);
        FutexWakeAll(child);
    }
    s.wakes_done.fetch_add(1, std::memory_order_release);
    const int status = s.fin_status.load(std::memory_order_relaxed);
    if (status > 0) FillOutput(&s, t.slot, input, true, &out);
    s.consumed.fetch_add(1, std::memory_order_release);
    if (status < 0) throw std::runtime_error("HIP forward pipe failed while evaluating a batch");
    return out;
}

std::vector<OutputResult> HipForwardPipe::BatchForward(int gpu, const std::vector<InputData>& inputs) {
    if (gpu < 0 || gpu >= static_cast<int>(graphs_.size())) throw std::runtime_error("BatchForward: bad gpu index");
    const int n = static_cast<int>(inputs.size());
    if (n > max_batch_) throw std::runtime_error("BatchForward: batch exceeds the constructed maximum");
    std::vector<OutputResult> outs(inputs.size());
    if (n == 0) return outs;
    Graph* g = graphs_[gpu].get();
    // pageable scratch staging of its own: the pinned sets belong to the queue path
    DNNWeights& w = *weights_;
    const size_t B2 = static_cast<size_t>(board_size_) * board_size_;
    Staging s;
    std::vector<float> planes(static_cast<size_t>(n) * w.input_channels * B2), prob(static_cast<size_t>(n) * w.probabilities_channels * B2),
        pass(static_cast<size_t>(n) * w.pass_probability_outputs), misc(static_cast<size_t>(n) * w.value_misc_outputs), own(static_cast<size_t>(n) * B2);
    s.planes = planes.data(); s.prob = prob.data(); s.pass = pass.data(); s.misc = misc.data(); s.own = own.data();
    s.bsz.assign(n, board_size_);
    for (int i = 0; i < n; ++i) StageInput(&s, i, inputs[i], true);
    {
        std::lock_guard<std::mutex> dev(g->dev_mu);  // keep the pump's batches out while we use the ctx
        if (sayuri_hip_forward(g->ctx, n, s.planes, s.bsz.data(), s.prob, s.pass, s.misc, s.own))
            ThrowHip("sayuri_hip_forward");
    }
    for (int i = 0; i < n; ++i) FillOutput(&s, i, Echo{inputs[i].board_size, inputs[i].offset, inputs[i].komi}, false, &outs[i]);
    batches_.fetch_add(1, std::memory_order_relaxed);
    evals_.fetch_add(static_cast<size_t>(n), std::memory_order_relaxed);
    return outs;
}

SAYURI_HOST_END
