// engine_graph.h -- EngineBase / Engine<T>: the per-GPU forward graph (weights, batch i/o on two tickets, the launches of one
// forward, the persistent tower run, chains) behind the C-ABI of engine.hip.  Counterpart of the reference's
// CudaForwardPipe::NNGraph (src/neural/cuda/cuda_forward_pipe.cc:133-1090).
#pragma once
#include "engine_plan.h"

namespace sayuri {

class EngineBase {
public:
    virtual ~EngineBase() {}
    virtual int load_tensor(int layer, int kind, const float* host, size_t n) = 0;
    // `packed` != null: the inputs are packed records (packed_planes.h) with `binary` bit planes, `planes` is ignored
    virtual int upload(int n, const float* planes, const int* board_sizes, const unsigned* packed = nullptr, int binary = 0) = 0;
    virtual int run() = 0;
    virtual int sync() = 0;
    virtual int download(float* prob, float* pass, float* misc, float* own) = 0;
    virtual int time_runs(int iters, float* ms) = 0;
    virtual int profile_run(sayuri_hip_kernel_stat* rows, int cap) = 0;
    virtual int submit(int n, const float* planes, const int* bsz, float* prob, float* pass, float* misc, float* own,
                       int* ticket, const unsigned* packed = nullptr, int binary = 0) = 0;
    virtual int wait(int ticket) = 0;
    virtual int query(int ticket) = 0;
    virtual int mark_kernel(const char* name) = 0;
    virtual int timed_stat(sayuri_hip_kernel_stat* row) = 0;
    virtual size_t device_bytes() const = 0;
    virtual int last_chains() const = 0;
    virtual int tower_state() const = 0;  // 1: the persistent tower kernel is loaded, 0: one launch per layer (fallback)
    virtual int debug_read(int buf, void* host, size_t bytes) = 0;  // debugging tap: activation buffer `buf` of ticket 0
};

template <typename T> class Engine : public EngineBase {
public:
    struct TileTabs { int* src = nullptr; int2* pix = nullptr; bool fresh = false; };
    struct BoardTabs { int* src = nullptr; int2* pix = nullptr; int* cols = nullptr; int npos_built = 0; int ntiles_built = 0; bool fresh = false; };
    Engine(int device, const sayuri_hip_netdesc& d, int max_batch, int board, const EngineFlags& flags)
        : flags_(flags), device_(device), desc_(d), max_batch_(max_batch), board_(board) {
        blocks_.assign(d.blocks, d.blocks + d.residual_blocks);
        desc_.blocks = blocks_.data();
    }
    ~Engine() override { release(); }

    int init() {
        HIP_OK(hipSetDevice(device_));
        HIP_OK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
        // The tickets' compute streams: one shared stream to begin with; the rule below gives ticket 1 a stream of its own.
        compute_[0] = stream_;
        compute_[1] = stream_;
        HIP_OK(hipStreamCreateWithFlags(&h2d_stream_, hipStreamNonBlocking));
        HIP_OK(hipStreamCreateWithFlags(&d2h_stream_, hipStreamNonBlocking));
        for (int t = 0; t < 2; ++t) {
            HIP_OK(hipEventCreateWithFlags(&h2d_done_[t], hipEventDisableTiming));
            HIP_OK(hipEventCreateWithFlags(&fwd_done_[t], hipEventDisableTiming));
        }
        HIP_OK(hipEventCreate(&ev0_));
        HIP_OK(hipEventCreate(&ev1_));
        enable_big_lds<T>();
        if (sizeof(T) == 2) enable_big_lds_glds();
        if (sizeof(T) == 2 && flags_.tower && tower_load()) {
            // the persistent launch is an optimisation: without its code object the per-layer launches (SAYURI_TOWER=0) run
            // the same kernels.  Say so once, loudly, and go on.
            std::fprintf(stderr, "[sayuri_hip] persistent tower kernel not loaded (%s): falling back to one launch per layer\n",
                         sayuri_hip_last_error());
            if (tower_mod_) (void)hipModuleUnload(tower_mod_);
            tower_mod_ = nullptr;
            tower_fn_[0] = tower_fn_[1] = nullptr;
        }
        // One stream per ticket (everything of a ticket in order on its own stream, no event between streams) whenever the
        // persistent kernel is loaded -- for every network, also one the launch does not cover (384 / 192 channels: the code
        // object is loaded and the layers are still launched one by one).  History of the rule: round 1 measured two tickets'
        // per-layer launches side by side as slower (48.0 k vs 54.0 k evals/s, the 482-workgroup kernel on the 256-channel
        // network evicting each other's L2 lines) and kept one compute stream; round 5 re-measured on the kernels of today --
        // 40b x 384 through submit / wait, two tickets in flight: a stream per ticket 26.3 k evals/s, one compute stream 23.7 k
        // (a layer is 450 workgroups, two rounds of the CUs, and the other ticket's launches fill the second; 200 batches
        // bit-identical to the solo result, tools/gpu/c5_pump.py, concurrent_ctx_dbg.py).  Without the code object
        // (SAYURI_TOWER=0, or a build whose seam was rejected) the older three-stream arrangement stays.
        if (describe_layers()) return -1;
        inorder_ = flags_.io_inorder && tower_fn_[0] != nullptr;
        if (flags_.compute_streams == 2 || inorder_) HIP_OK(hipStreamCreateWithFlags(&compute_[1], hipStreamNonBlocking));
        return 0;
    }
    // every 3x3 convolution of the residual tower has 128 or 256 (padded) output channels and there is at least one
    bool tower_covers_net() const {
        int n3 = 0;
        for (const auto& kv : convs_) {
            const ConvLayerDev& L = kv.second;
            if (L.k != 3 || L.depthwise || kv.first == SAYURI_L_INPUT_CONV) continue;
            const int cs = round_up(L.cout, 32);
            if (cs != 256 && cs != 128) return false;
            ++n3;
        }
        const int c0 = round_up(desc_.residual_channels, 32);
        return n3 > 0 && (c0 == 256 || c0 == 128);
    }

    // -------------------------------------------------------------- chains (see forward())
    static constexpr int kMaxChains = 4;
    hipStream_t chain_stream_[kMaxChains] = {};
    hipEvent_t chain_fork_ = nullptr, chain_join_[kMaxChains] = {};
    int rg_tile0_ = 0, rg_ntiles_ = -1, rg_n0_ = 0, rg_ns_ = -1;  // the tiles / samples the launches of forward_graph() cover (-1: all)
    int last_chains_ = 1;
    int range_ntiles() const { return rg_ntiles_ >= 0 ? rg_ntiles_ : board_plan_.ntiles; }
    int range_ns() const { return rg_ns_ >= 0 ? rg_ns_ : geom_.n; }
    double range_px() const { return rg_ns_ >= 0 ? (double)(geom_.off[rg_n0_ + rg_ns_] - geom_.off[rg_n0_]) : (double)geom_.total; }
    int chain_setup(int G) {
        if (!chain_fork_) HIP_OK(hipEventCreateWithFlags(&chain_fork_, hipEventDisableTiming));
        for (int g = 0; g < G; ++g) {
            if (!chain_stream_[g]) HIP_OK(hipStreamCreateWithFlags(&chain_stream_[g], hipStreamNonBlocking));
            if (!chain_join_[g]) HIP_OK(hipEventCreateWithFlags(&chain_join_[g], hipEventDisableTiming));
        }
        return 0;
    }
    // How many chains the current batch is run as.  A layer of a network the persistent launch does not cover is one launch of
    // (tiles x channel tiles) workgroups of equal cost; at 450 of them (configs[4]: 150 tiles x three 128-channel tiles) the 256 CUs
    // run two rounds, the second 76 % full, whatever the item size.  The boards of a batch are independent: cut into G groups of
    // tiles, each a chain of per-layer launches on a stream of its own, a group's next layer starts on the CUs another group's
    // round leaves free -- the cross-layer pipelining of a persistent (layer, tile, channel tile) run, done by the dispatcher
    // instead of by dependency counters.  Measured from outside the engine (tools/gpu/c5_streams.py, G contexts): 24.4 k evals/s
    // as one chain, 25.8 / 25.9 / 25.6 k as 2 / 3 / 4, 21.4 k as 6.
    int chains_for_batch() {
        // The chains share the forward's activation buffers: a tile is whole samples, every layer of a network that qualifies has
        // ONE channel stride, and the packed input keeps a buffer of its own (forward_graph), so the chains' rows are disjoint
        // bytes in every buffer.  (The first version recycled the packed input's buffer, whose rows have another stride: a chain
        // that ran ahead overwrote a later chain's input -- a few dozen samples per batch off in two runs of three.)
        if (sizeof(T) != 2 || flags_.chains == 1 || profiling_ || light_ || !head_img_ || !heads_fused_enabled()) return 1;
        if (desc_.policy_head_type != 0 || tower_covers_net()) return 1;
        for (const auto& b : blocks_)
            if (b.type != SAYURI_BLOCK_RESIDUAL) return 1;  // every layer of the graph must be a board convolution or a per-sample kernel
        const ConvLayerDev& L = cv(SAYURI_L_BLOCK(0, SAYURI_S_CONV1));
        // (a network whose SE units could run inside the convolution keeps one chain: conv_se launches over the whole batch)
        for (const auto& e : kBoardEntries)
            if (e.fn_se && e.kot == L.ko_pad)
                for (const auto& b : blocks_)
                    if (b.apply_se) return 1;
        int kts = 0;
        if (!choose_board(L, &kts) || !board_plan_.ok) return 1;
        const int wgs = board_plan_.ntiles * kts;
        if (wgs <= kNumCU) return 1;  // one round already
        int G = flags_.chains > 1 ? flags_.chains : std::min(kMaxChains, std::max(2, (wgs + 159) / 160));
        G = std::min(G, board_plan_.ntiles / 8);
        return std::max(G, 1);
    }

    // -------------------------------------------------------------- weights
    int load_tensor(int layer, int kind, const float* host, size_t n) override {
        if (finalized_) return fail("load_tensor after the first forward");
        auto ci = convs_.find(layer);
        if (ci != convs_.end()) {
            ConvLayerDev& L = ci->second;
            const size_t expect = kind == SAYURI_T_WEIGHTS
                                      ? (size_t)(L.depthwise ? 1 : L.cin) * L.cout * L.k * L.k
                                      : (size_t)L.cout;
            if (n != expect) return fail("conv tensor size mismatch for layer " + std::to_string(layer));
            (kind == SAYURI_T_WEIGHTS ? L.hw : L.hb).assign(host, host + n);
            return 0;
        }
        auto fi = fcs_.find(layer);
        if (fi != fcs_.end()) {
            FcLayerDev& L = fi->second;
            const size_t expect = kind == SAYURI_T_WEIGHTS ? (size_t)L.in * L.out : (size_t)L.out;
            if (n != expect) return fail("fc tensor size mismatch for layer " + std::to_string(layer));
            (kind == SAYURI_T_WEIGHTS ? L.hw : L.hb).assign(host, host + n);
            return 0;
        }
        return fail("unknown layer id " + std::to_string(layer));
    }

    // -------------------------------------------------------------- batch i/o
    // asynchronous: H2D, graph, D2H enqueued on the stream, an event marks the end
    int submit(int n, const float* planes, const int* board_sizes, float* prob, float* pass, float* misc, float* own,
               int* ticket, const unsigned* packed = nullptr, int binary = 0) override {
        // The planes of batch k+1 cross PCIe while batch k computes, and the results of batch k while batch k+1 computes; each
        // of the two tickets owns its own device input / geometry / output buffers (and, by default, its own stream: below).
        const int t = next_ticket_;
        next_ticket_ ^= 1;
        HIP_OK(hipSetDevice(device_));
        if (finalize()) return -1;
        select_slot(t);
        // Default (SAYURI_IO_INORDER=0 is the older arrangement below): everything of a ticket on the ticket's own stream, in
        // order -- no event between streams at all.  Each record / wait is a marker the runtime's signal thread handles; the
        // seven per batch of the three-stream arrangement kept that thread at a full host core during self-play (one of
        // eight busy; profiles/r04_host_profile.txt), for the same evals/s.  The other ticket's stream overlaps its copies
        // with this one's kernels as before; a ticket's slot is free when its stream gets to the next use.
        const bool inorder = inorder_;
        // One exception: fp32 planes (62 KB per sample, 16 MB per batch).  On the ticket's own stream, behind kernels, the runtime
        // moves them with a copy KERNEL, which cannot run beside the other ticket's persistent launch: the upload waited for that
        // launch to end (72.1 -> 65.0 k evals/s through submit / wait; packed planes are 0.5 MB and do not show it).  They keep the
        // upload stream and one event; the slot is free for them when the ticket's last completion event has fired, which the
        // caller has normally seen already.
        const bool big_upload = inorder && packed == nullptr;
        hipStream_t up = inorder && !big_upload ? stream_ : h2d_stream_, down = inorder ? stream_ : d2h_stream_;
        if (!inorder) HIP_OK(hipStreamWaitEvent(h2d_stream_, fwd_done_[t], 0));  // the forward that last read this slot's inputs
        else if (big_upload && tick_ev_[t]) HIP_OK(hipStreamWaitEvent(h2d_stream_, tick_ev_[t], 0));
        if (enqueue_inputs(n, planes, board_sizes, up, packed, binary, /*in_place=*/true)) return -1;
        if (!inorder || big_upload) {
            HIP_OK(hipEventRecord(h2d_done_[t], h2d_stream_));
            HIP_OK(hipStreamWaitEvent(stream_, h2d_done_[t], 0));
            if (!inorder && tick_ev_[t]) HIP_OK(hipStreamWaitEvent(stream_, tick_ev_[t], 0));  // the download that last read this slot's outputs
        }
        have_batch_ = true;
        if (fwdstat_) {  // SAYURI_HIP_FWDSTAT (measuring aid): device time of every submitted forward
            for (int k = 0; k < 2; ++k)
                if (!fs_ev_[t][k]) HIP_OK(hipEventCreate(&fs_ev_[t][k]));

            HIP_OK(hipEventRecord(fs_ev_[t][0], stream_));
        }
        const int uploads_before = table_uploads_;
        // The persistent tower launch holds every CU (all registers, all LDS) for the whole forward.  A copy the runtime
        // does with a blit KERNEL (everything below 16 KB: pass, misc, the three geometry arrays, the tower table) can
        // therefore not run beside the OTHER ticket's tower launch: with two batches in flight a finished batch's small
        // downloads -- and with them its completion event -- waited for the next batch's 3.5 ms launch to end, the next
        // batch's geometry uploads likewise (driver line of round 3: 64.4 k evals/s in self-play against 70.5 k resident).
        // So nothing small is copied any more: the heads kernel stores pass / misc straight into the caller's pinned
        // buffers (20 KB of posted PCIe writes), a uniform batch uses geometry arrays that are resident (enqueue_inputs),
        // and the tower table does not depend on the batch size (tower_append).  The two large outputs keep their DMA copies.
        zc_pass_ = flags_.io_zc ? zc_device_pointer(pass) : nullptr;
        zc_misc_ = zc_pass_ ? zc_device_pointer(misc) : nullptr;
        if (!zc_misc_) zc_pass_ = nullptr;  // both or neither: the heads kernel takes one path
        const int frc = forward();
        const bool small_direct = zc_pass_ != nullptr;
        zc_pass_ = zc_misc_ = nullptr;
        if (frc) return -1;
        if (fwdstat_) {
            HIP_OK(hipEventRecord(fs_ev_[t][1], stream_));
            fs_pending_[t] = true;
            fs_n_[t] = n;
            fs_uploads_ += table_uploads_ - uploads_before;
        }
        if (!inorder) {
            HIP_OK(hipEventRecord(fwd_done_[t], stream_));
            HIP_OK(hipStreamWaitEvent(d2h_stream_, fwd_done_[t], 0));
        }
        const size_t B2 = (size_t)board_ * board_;
        HIP_OK(hipMemcpyAsync(prob, d_prob_, sizeof(float) * n * desc_.probabilities_channels * B2, hipMemcpyDeviceToHost, down));
        if (!small_direct) {
            HIP_OK(hipMemcpyAsync(pass, d_pass_, sizeof(float) * n * desc_.pass_probability_outputs, hipMemcpyDeviceToHost, down));
            HIP_OK(hipMemcpyAsync(misc, d_misc_, sizeof(float) * n * desc_.value_misc_outputs, hipMemcpyDeviceToHost, down));
        }
        HIP_OK(hipMemcpyAsync(own, d_own_, sizeof(float) * n * B2, hipMemcpyDeviceToHost, down));
        if (!tick_ev_[t]) HIP_OK(hipEventCreateWithFlags(&tick_ev_[t], hipEventDisableTiming));
        HIP_OK(hipEventRecord(tick_ev_[t], down));
        if (fwdstat_) {
            if (!fs_ev_[t][2]) HIP_OK(hipEventCreate(&fs_ev_[t][2]));
            HIP_OK(hipEventRecord(fs_ev_[t][2], down));
        }
        *ticket = t;
        return 0;
    }
    int wait(int ticket) override {
        if (ticket < 0 || ticket > 1 || !tick_ev_[ticket]) return fail("wait: bad ticket");
        HIP_OK(hipSetDevice(device_));
        HIP_OK(hipEventSynchronize(tick_ev_[ticket]));
        if (sx_check()) return -1;
        if (fwdstat_ && fs_pending_[ticket]) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, fs_ev_[ticket][0], fs_ev_[ticket][1]) == hipSuccess) {
                const int b = fs_n_[ticket] >= max_batch_ ? 2 : (fs_n_[ticket] * 2 > max_batch_ ? 1 : 0);
                fs_ms_[b] += ms;
                fs_cnt_[b] += 1;
            }
            if (fs_ev_[ticket][2] && hipEventElapsedTime(&ms, fs_ev_[ticket][1], fs_ev_[ticket][2]) == hipSuccess) {
                fs_d2h_ms_ += ms;
                fs_d2h_max_ = std::max(fs_d2h_max_, (double)ms);
                fs_d2h_slow_ += ms > 1.0f;
            }

            fs_pending_[ticket] = false;
        }
        return 0;
    }
    int query(int ticket) override {
        if (ticket < 0 || ticket > 1 || !tick_ev_[ticket]) return fail("query: bad ticket");
        const hipError_t e = hipEventQuery(tick_ev_[ticket]);
        if (e == hipSuccess) return 1;
        if (e == hipErrorNotReady) return 0;
        return fail(std::string("hipEventQuery: ") + hipGetErrorString(e));
    }

    int upload(int n, const float* planes, const int* board_sizes, const unsigned* packed = nullptr, int binary = 0) override {
        HIP_OK(hipSetDevice(device_));
        if (finalize()) return -1;
        HIP_OK(hipStreamSynchronize(h2d_stream_));
        HIP_OK(hipStreamSynchronize(d2h_stream_));
        for (hipStream_t cs : compute_)
            if (cs) HIP_OK(hipStreamSynchronize(cs));
        select_slot(0);
        if (enqueue_inputs(n, planes, board_sizes, stream_, packed, binary)) return -1;
        HIP_OK(hipStreamSynchronize(stream_));
        have_batch_ = true;
        return 0;
    }

    // The heads kernel stores pass / misc straight into the caller's buffers when the device can address them: page-locked
    // memory that is MAPPED (sayuri_hip_host_alloc, hipHostMalloc, hipHostRegister with the mapped flag).  Anything else --
    // pinned but unmapped, pageable, another allocator's -- would fault the GPU inside the kernel with no error returned, so a
    // pointer is asked about once (hipHostGetDevicePointer) and the answer kept; a buffer that is not device-addressable gets
    // its results through the device-side copies and hipMemcpyAsync as before.
    float* zc_device_pointer(float* host) {
        if (!host) return nullptr;
        // (an address may be handed out again after sayuri_hip_host_free, to memory of another kind: answers do not outlive a free)
        const unsigned gen = g_host_free_gen.load(std::memory_order_acquire);
        if (gen != zc_gen_) {
            zc_known_.clear();
            zc_gen_ = gen;
        }
        auto it = zc_known_.find(host);
        if (it != zc_known_.end()) return it->second;
        void* dp = nullptr;
        float* ans = nullptr;
        if (hipHostGetDevicePointer(&dp, host, 0) == hipSuccess && dp) ans = (float*)dp;
        else (void)hipGetLastError();  // not an error of this engine: the fallback path is taken
        if (zc_known_.size() >= 64) zc_known_.clear();
        zc_known_[host] = ans;
        return ans;
    }
    std::map<float*, float*> zc_known_;
    unsigned zc_gen_ = 0;

    // resident geometry arrays of a uniform batch of `bs` x `bs` boards (valid for every n <= max_batch)
    struct IdentGeom { int *off = nullptr, *bsz = nullptr, *perm = nullptr; };
    const IdentGeom* ident_geom(int bs) {
        auto it = ident_.find(bs);
        if (it != ident_.end()) return &it->second;
        IdentGeom g;
        std::vector<int> off(max_batch_ + 1), bz(max_batch_, bs), pm(max_batch_);
        for (int i = 0; i <= max_batch_; ++i) off[i] = i * bs * bs;
        for (int i = 0; i < max_batch_; ++i) pm[i] = i;
        if (dev_upload(&g.off, off) || dev_upload(&g.bsz, bz) || dev_upload(&g.perm, pm)) return nullptr;
        return &ident_.emplace(bs, g).first->second;
    }
    std::map<int, IdentGeom> ident_;

    // geometry + planes H2D on the stream (no sync).  The geometry arrays are staged in a
    // 2-deep pinned ring so a second batch can be enqueued while the first is still copying.
    int enqueue_inputs(int n, const float* planes, const int* board_sizes, hipStream_t copy_stream, const unsigned* packed = nullptr,
                       int binary = 0, bool in_place = false) {
        HIP_OK(hipSetDevice(device_));
        if (n <= 0 || n > max_batch_) return fail("batch size out of range");
        if (packed && (binary <= 0 || binary > desc_.input_channels || desc_.input_channels - binary > 8 || board_ * board_ > 12 * 32))
            return fail("packed planes: bad binary plane count for this network");
        if (finalize()) return -1;
        prev_bsz_.swap(geom_.bsz);
        geom_.n = n;
        geom_.bsz.resize(n);
        geom_.off.resize(n + 1);
        geom_.off[0] = 0;
        // Device order = the batch's samples sorted by board size (largest first, stable): samples of one size become
        // neighbours, so the one-workgroup-per-board convolution packs them into full tiles (two 13x13 or four 9x9 boards
        // per tile) whatever order the queue collected them in.  perm[i] = the caller's slot of device sample i; only
        // pack_input (reads the planes) and head_tail (writes the outputs) see the caller's order.
        perm_.resize(n);
        bool mixed = false;
        for (int i = 0; i < n; ++i) {
            const int bs = board_sizes ? board_sizes[i] : board_;
            if (bs < 2 || bs > board_) return fail("sample board size out of range");
            perm_[i] = i;
            mixed |= board_sizes && bs != board_sizes[0];
        }
        if (mixed) std::stable_sort(perm_.begin(), perm_.end(), [&](int a, int b) { return board_sizes[a] > board_sizes[b]; });
        for (int i = 0; i < n; ++i) {
            geom_.bsz[i] = board_sizes ? board_sizes[perm_[i]] : board_;
            geom_.off[i + 1] = geom_.off[i] + geom_.bsz[i] * geom_.bsz[i];
        }
        geom_.total = geom_.off[n];
        if (geom_.bsz != prev_bsz_) {  // tile choices and index tables depend on the geometry only
            tile_cache_.clear();
            glds_cache_.clear();
            board_plan_valid_ = false;
        }
        IoSlot& slot = io_[cur_slot_];
        const bool uniform = !mixed;
        // One sample per tile and one board size: the tables of a LONGER batch of the same size serve a shorter one (tile i
        // depends on sample i alone), so a queue that alternates between 256 and 250 positions keeps its tables.
        const bool one_per_tile = uniform && 2 * geom_.bsz[0] * geom_.bsz[0] > kBoardPT;
        const bool prefix = flags_.io_prefix && one_per_tile && slot.tabs_single && slot.tabs_bsz.size() >= (size_t)n &&
                            slot.tabs_bsz[0] == geom_.bsz[0];
        if (!prefix && slot.tabs_bsz != geom_.bsz) {  // this slot's tables were built for another geometry
            for (auto& kv : slot.tabs) kv.second.fresh = false;
            slot.board.fresh = false;
            slot.tabs_bsz = geom_.bsz;
            slot.tabs_single = one_per_tile;
        }
        // the tables of the across-sample tiles (conv_glds.h) know the pixel total: they serve exactly the batch size they were
        // built for (the board tables above serve every prefix)
        if (slot.tabs_n != n) {
            for (auto& kv : slot.tabs) kv.second.fresh = false;
            slot.tabs_n = n;
        }
        // Geometry arrays on the device.  A uniform batch (the self-play queue: every position on the NN board) uses arrays
        // that are resident -- off[i] = i * bs^2, bsz[i] = bs, perm[i] = i hold for every n -- so nothing is copied.  A mixed
        // batch stages its arrays in a pinned ring and a one-workgroup kernel ON THE FORWARD'S OWN STREAM moves them: a
        // copy of a few KB is a blit kernel to the runtime, and on the copy stream it would wait for the other ticket's
        // persistent tower launch to give up a CU (see submit()).
        if (flags_.io_geom) {
            if (uniform) {
                const IdentGeom* id = ident_geom(geom_.bsz[0]);
                if (!id) return -1;
                d_off_ = id->off; d_bsz_ = id->bsz; d_perm_ = id->perm;
            } else {
                d_off_ = slot.off; d_bsz_ = slot.bsz; d_perm_ = slot.perm;
                int* hg = h_geom_ + (size_t)geom_slot_ * (3 * max_batch_ + 1);
                geom_slot_ ^= 1;
                std::memcpy(hg, geom_.off.data(), sizeof(int) * (n + 1));
                std::memcpy(hg + max_batch_ + 1, geom_.bsz.data(), sizeof(int) * n);
                std::memcpy(hg + 2 * max_batch_ + 1, perm_.data(), sizeof(int) * n);
                hipLaunchKernelGGL(geom_stage_kernel, dim3(1), dim3(256), 0, stream_, (const int*)hg, max_batch_, n, d_off_, d_bsz_, d_perm_);
                HIP_OK(hipGetLastError());
            }
        } else {
            int* hg = h_geom_ + (size_t)geom_slot_ * (3 * max_batch_ + 1);
            geom_slot_ ^= 1;
            std::memcpy(hg, geom_.off.data(), sizeof(int) * (n + 1));
            std::memcpy(hg + max_batch_ + 1, geom_.bsz.data(), sizeof(int) * n);
            std::memcpy(hg + 2 * max_batch_ + 1, perm_.data(), sizeof(int) * n);
            HIP_OK(hipMemcpyAsync(d_off_, hg, sizeof(int) * (n + 1), hipMemcpyHostToDevice, copy_stream));
            HIP_OK(hipMemcpyAsync(d_bsz_, hg + max_batch_ + 1, sizeof(int) * n, hipMemcpyHostToDevice, copy_stream));
            HIP_OK(hipMemcpyAsync(d_perm_, hg + 2 * max_batch_ + 1, sizeof(int) * n, hipMemcpyHostToDevice, copy_stream));
        }
        IoSlot& io = io_[cur_slot_];
        io.packed_binary = packed ? binary : 0;
        if (packed) {
            const size_t words = (size_t)binary * 12 + 8;
            // Packed records in device-addressable host memory (sayuri_hip_host_alloc: the pump's buffers) are not copied at all:
            // pack_bits_kernel reads the 1.8 KB per sample across PCIe itself.  The copy was 14 us of DMA -- but the copy engine
            // takes its packets in the order they were submitted, and the OTHER ticket's two downloads, submitted earlier and
            // waiting for that ticket's heads kernel, were ahead of it: batch k+1's upload, and with it pack_bits and the tower
            // launch, started only after batch k's results had gone out (135 us between two tower launches against 54 us for
            // one forward after another on one stream; tools/pump_gaps.py, profiles/r05_pump_gaps.txt).  The caller keeps the
            // records untouched until wait(), as it must for the asynchronous copy.
            io.packed_src = nullptr;
            if (in_place && flags_.io_zc_in) io.packed_src = (const unsigned*)zc_device_pointer((float*)const_cast<unsigned*>(packed));
            if (io.packed_src) return 0;
            if (!io.packed && dev_alloc(&io.packed, (size_t)max_batch_ * (40 * 12 + 8))) return -1;
            HIP_OK(hipMemcpyAsync(io.packed, packed, sizeof(unsigned) * n * words, hipMemcpyHostToDevice, copy_stream));
            return 0;
        }
        HIP_OK(hipMemcpyAsync(d_planes_, planes, sizeof(float) * (size_t)n * desc_.input_channels * board_ * board_,
                              hipMemcpyHostToDevice, copy_stream));
        return 0;
    }

    int run() override {
        if (!have_batch_) return fail("run before upload");
        HIP_OK(hipSetDevice(device_));
        return forward();
    }

    int sync() override {
        HIP_OK(hipSetDevice(device_));
        HIP_OK(hipStreamSynchronize(stream_));
        return sx_check();
    }

    int download(float* prob, float* pass, float* misc, float* own) override {
        HIP_OK(hipSetDevice(device_));
        const size_t n = geom_.n, B2 = (size_t)board_ * board_;
        if (prob) HIP_OK(hipMemcpyAsync(prob, d_prob_, sizeof(float) * n * desc_.probabilities_channels * B2, hipMemcpyDeviceToHost, stream_));
        if (pass) HIP_OK(hipMemcpyAsync(pass, d_pass_, sizeof(float) * n * desc_.pass_probability_outputs, hipMemcpyDeviceToHost, stream_));
        if (misc) HIP_OK(hipMemcpyAsync(misc, d_misc_, sizeof(float) * n * desc_.value_misc_outputs, hipMemcpyDeviceToHost, stream_));
        if (own) HIP_OK(hipMemcpyAsync(own, d_own_, sizeof(float) * n * B2, hipMemcpyDeviceToHost, stream_));
        HIP_OK(hipStreamSynchronize(stream_));
        return sx_check();
    }

    int time_runs(int iters, float* ms) override {
        if (!have_batch_) return fail("time_runs before upload");
        HIP_OK(hipSetDevice(device_));
        pool_used_ = 0;
        group_counts_.clear();
        group_open_ = false;
        light_ = !light_name_.empty();
        HIP_OK(hipEventRecord(ev0_, stream_));
        for (int i = 0; i < iters; ++i)
            if (forward()) { light_ = false; return -1; }
        if (group_open_ && close_group()) { light_ = false; return -1; }
        HIP_OK(hipEventRecord(ev1_, stream_));
        light_ = false;
        HIP_OK(hipEventSynchronize(ev1_));
        HIP_OK(hipEventElapsedTime(ms, ev0_, ev1_));
        // fold the event pairs of the marked kernel class into one stat row (a pair brackets group_counts_[k] launches)
        timed_stat_ = Stat{};
        for (size_t k = 0; k < group_counts_.size() && 2 * k + 1 < pool_used_; ++k) {
            float t = 0.f;
            HIP_OK(hipEventElapsedTime(&t, pool_[2 * k], pool_[2 * k + 1]));
            timed_stat_.launches += group_counts_[k];
            timed_stat_.ms += t;
            timed_stat_.flops += light_flops_ * group_counts_[k];
            timed_stat_.bytes += light_bytes_ * group_counts_[k];
        }
        return 0;
    }

    // "class" brackets every launch of the class with its own event pair; "class/N" brackets RUNS of up to N consecutive
    // launches of the class with one pair (a launch of another class ends the run).  An event is a barrier packet on the
    // stream: the launch that follows it starts from an empty pipeline (~2-3 us), which a pair per launch both adds to
    // the step (34 tower launches: ~5 %) and counts into every measured duration; per run of five it is a fifth of that.
    int mark_kernel(const char* name) override {
        light_name_ = name ? name : "";
        light_group_ = 1;
        const size_t slash = light_name_.find('/');
        if (slash != std::string::npos) {
            light_group_ = std::max(1, atoi(light_name_.c_str() + slash + 1));
            light_name_.resize(slash);
        }
        return 0;
    }
    int timed_stat(sayuri_hip_kernel_stat* row) override {
        std::memset(row, 0, sizeof(*row));
        std::snprintf(row->name, sizeof(row->name), "%s", light_name_.c_str());
        row->launches = timed_stat_.launches;
        row->total_ms = timed_stat_.ms;
        row->flops = timed_stat_.flops;
        row->bytes = timed_stat_.bytes;
        return 0;
    }

    int profile_run(sayuri_hip_kernel_stat* rows, int cap) override {
        if (!have_batch_) return fail("profile_run before upload");
        HIP_OK(hipSetDevice(device_));
        stats_.clear();
        profiling_ = true;
        const int rc = forward();
        profiling_ = false;
        if (rc) return -1;
        if (d_hdbg_) {
            std::vector<unsigned long long> h(4 * 8);
            HIP_OK(hipMemcpy(h.data(), d_hdbg_, h.size() * 8, hipMemcpyDeviceToHost));
            for (int wg = 0; wg < 4; ++wg) {
                const unsigned long long* d = &h[wg * 8];
                fprintf(stderr, "[heads timeline wg%d] DMA + K loop %llu | act, per-pixel MFMA, pooling partials %llu | pool fold %llu | FC 1 %llu | FC 2 + row bias %llu | stores %llu | total %llu\n",
                        wg, d[1] - d[0], d[2] - d[1], d[3] - d[2], d[4] - d[3], d[5] - d[4], d[6] - d[5], d[6] - d[0]);
            }
        }
        if (d_sxdbg_) {  // SAYURI_SX_DBG: the split SE convolution from inside (100 MHz ticks; blocks 0 / 8 / 16 = the siblings of a tile, 1)
            std::vector<unsigned long long> h(4 * 64);
            HIP_OK(hipMemcpy(h.data(), d_sxdbg_, h.size() * 8, hipMemcpyDeviceToHost));
            static const char* who[4] = {"block 0 (tile 0, kt 0)", "block 8 (tile 0, kt 1)", "block 16 (tile 0, kt 2)", "block 1 (tile 1, kt 0)"};
            for (int wg = 0; wg < 4; ++wg) {
                const unsigned long long* d = &h[(size_t)wg * 64];
                fprintf(stderr, "[split SE timeline %s] start +%llu | K loop %llu | pooling %llu | squeeze + publish %llu | siblings in %llu | mid + excite %llu | gate %llu | epilogue %llu | total %llu\n",
                        who[wg], d[0] - h[0], d[1] - d[0], d[2] - d[1], d[3] - d[2], d[4] - d[3], d[5] - d[4], d[6] - d[5], d[7] - d[6], d[7] - d[0]);
            }
        }
        if (d_dbg_) {  // SAYURI_BOARD_DBG: s_memtime timeline of the last tower convolution (workgroups 0-3, all waves)
            std::vector<unsigned long long> h(4 * 8 * 8);
            HIP_OK(hipMemcpy(h.data(), d_dbg_, h.size() * 8, hipMemcpyDeviceToHost));
            for (int wg = 0; wg < 4 && dbg_is_se_; ++wg)
                for (int w = 0; w < 8; w += 4) {
                    const unsigned long long* d = &h[((size_t)wg * 8 + w) * 8];
                    fprintf(stderr, "[board+SE timeline wg%d wave%d] prologue+K loop %llu | pooling %llu | squeeze FC %llu | excite FC %llu | gate applied %llu | epilogue %llu | total %llu\n",
                            wg, w, d[1] - d[0], d[2] - d[1], d[3] - d[2], d[4] - d[3], d[5] - d[4], d[6] - d[5], d[6] - d[0]);
                }
            for (int wg = 0; wg < 4 && !dbg_is_se_; ++wg)
                for (int w = 0; w < 8; ++w) {
                    const unsigned long long* d = &h[((size_t)wg * 8 + w) * 8];
                    fprintf(stderr, "[board timeline wg%d wave%d] tables+first DMA %llu | first barrier %llu | main loop %llu (sync %llu) | epilogue %llu | total %llu\n",
                            wg, w, d[1] - d[0], d[2] - d[1], d[3] - d[2], d[5], d[4] - d[3], d[4] - d[0]);
                }
        }
        int i = 0;
        for (auto& kv : stats_) {
            if (i >= cap) break;
            std::memset(&rows[i], 0, sizeof(rows[i]));
            std::snprintf(rows[i].name, sizeof(rows[i].name), "%s", kv.first.c_str());
            rows[i].launches = kv.second.launches;
            rows[i].total_ms = kv.second.ms;
            rows[i].flops = kv.second.flops;
            rows[i].bytes = kv.second.bytes;
            ++i;
        }
        return i;
    }

    size_t device_bytes() const override { return dev_bytes_; }
    int last_chains() const override { return last_chains_; }
    int tower_state() const override { return tower_fn_[0] != nullptr ? 1 : 0; }
    int debug_read(int buf, void* host, size_t bytes) override {
        if (buf < 0 || buf >= kNumBufs || !io_[0].bufs[buf]) return fail("debug_read: no such buffer");
        HIP_OK(hipSetDevice(device_));
        HIP_OK(hipDeviceSynchronize());
        const size_t have = (size_t)max_batch_ * slot_pix_ * cs_max_ * sizeof(T);
        HIP_OK(hipMemcpy(host, io_[0].bufs[buf], std::min(bytes, have), hipMemcpyDeviceToHost));
        return 0;
    }

private:
    // -------------------------------------------------------------- construction helpers
    void add_conv(int id, int cin, int cout, int k, bool depthwise = false) {
        ConvLayerDev L;
        L.cin = cin; L.cout = cout; L.k = k; L.depthwise = depthwise;
        convs_[id] = L;
    }
    void add_fc(int id, int in, int out) {
        FcLayerDev L;
        L.in = in; L.out = out;
        fcs_[id] = L;
    }

    int describe_layers() {
        const auto& d = desc_;
        const int C = d.residual_channels;
        if (C <= 0 || d.input_channels <= 0) return fail("bad net description");
        add_conv(SAYURI_L_INPUT_CONV, d.input_channels, C, 3);
        cs_max_ = std::max(round_up(C, 32), round_up(d.input_channels, 32));
        for (int b = 0; b < d.residual_blocks; ++b) {
            const auto& bd = blocks_[b];
            const int I = bd.bottleneck_channels, F = bd.feedforward_channels;
            switch (bd.type) {
            case SAYURI_BLOCK_RESIDUAL:
                add_conv(SAYURI_L_BLOCK(b, SAYURI_S_CONV1), C, C, 3);
                add_conv(SAYURI_L_BLOCK(b, SAYURI_S_CONV2), C, C, 3);
                break;
            case SAYURI_BLOCK_BOTTLENECK:
            case SAYURI_BLOCK_NESTED_BOTTLENECK:
                if (I <= 0) return fail("bottleneck block without bottleneck_channels");
                add_conv(SAYURI_L_BLOCK(b, SAYURI_S_PRE_BTL), C, I, 1);
                add_conv(SAYURI_L_BLOCK(b, SAYURI_S_CONV1), I, I, 3);
                add_conv(SAYURI_L_BLOCK(b, SAYURI_S_CONV2), I, I, 3);
                if (bd.type == SAYURI_BLOCK_NESTED_BOTTLENECK) {
                    add_conv(SAYURI_L_BLOCK(b, SAYURI_S_CONV3), I, I, 3);
                    add_conv(SAYURI_L_BLOCK(b, SAYURI_S_CONV4), I, I, 3);
                }
                add_conv(SAYURI_L_BLOCK(b, SAYURI_S_POST_BTL), I, C, 1);
                cs_max_ = std::max(cs_max_, round_up(I, 32));
                break;
            case SAYURI_BLOCK_MIXER:
                if (F <= 0 || bd.dw_filter <= 0) return fail("mixer block without ffn channels / dw filter");
                add_conv(SAYURI_L_BLOCK(b, SAYURI_S_DW_CONV), C, C, bd.dw_filter, true);
                add_conv(SAYURI_L_BLOCK(b, SAYURI_S_CONV1), C, F, 1);
                add_conv(SAYURI_L_BLOCK(b, SAYURI_S_CONV2), F, C, 1);
                cs_max_ = std::max(cs_max_, round_up(F, 32));
                break;
            default:
                return fail("unknown block type");
            }
            if (bd.apply_se) {
                if (bd.se_size <= 0) return fail("SE block without se_size");
                add_fc(SAYURI_L_BLOCK(b, SAYURI_S_SQUEEZE), 3 * C, bd.se_size);
                add_fc(SAYURI_L_BLOCK(b, SAYURI_S_EXCITE), bd.se_size, 2 * C);
            }
        }
        const int Cp = d.policy_head_channels, Cv = d.value_head_channels;
        if (Cp <= 0 || Cv <= 0) return fail("bad head channels");
        if (d.ownership_channels != 1) return fail("ownership_channels must be 1");
        if (d.probabilities_channels > 8) return fail("too many policy planes");
        add_conv(SAYURI_L_P_HD_CONV, C, Cp, 1);
        if (d.policy_head_type == 1) {
            if (d.policy_dw_filter <= 0) return fail("RepLK head without dw filter");
            add_conv(SAYURI_L_P_DW_CONV, Cp, Cp, d.policy_dw_filter, true);
            add_conv(SAYURI_L_P_PT_CONV, Cp, Cp, 1);
        }
        add_fc(SAYURI_L_P_INTER_FC, 3 * Cp, Cp);
        add_conv(SAYURI_L_PROB_CONV, Cp, d.probabilities_channels, 1);
        add_fc(SAYURI_L_PASS_FC, Cp, d.pass_probability_outputs);
        add_conv(SAYURI_L_V_HD_CONV, C, Cv, 1);
        add_fc(SAYURI_L_V_INTER_FC, 3 * Cv, 3 * Cv);
        add_conv(SAYURI_L_V_OWNERSHIP, Cv, 1, 1);
        add_fc(SAYURI_L_V_MISC, 3 * Cv, d.value_misc_outputs);
        cs_max_ = std::max(cs_max_, std::max(round_up(Cp, 32), round_up(Cv, 32)));
        return 0;
    }

    template <typename U> int dev_alloc(U** p, size_t count) {
        void* q = nullptr;
        const size_t bytes = std::max<size_t>(count * sizeof(U), 256);
        HIP_OK(hipMalloc(&q, bytes));
        // The zero fill runs on the NULL stream and hipMemset returns before it has: this engine's streams are non-blocking, so
        // a copy or kernel they write into the new buffer right away could be overtaken by it (round 4: the first packed batch of
        // a staging slot lost the tail of its records to the fill of the buffer allocated for them one call earlier).
        HIP_OK(hipMemset(q, 0, bytes));
        HIP_OK(hipStreamSynchronize(nullptr));
        allocs_.push_back(q);
        dev_bytes_ += bytes;
        *p = (U*)q;
        return 0;
    }
    template <typename U> int dev_upload(U** p, const std::vector<U>& h) {
        if (dev_alloc(p, h.size())) return -1;
        HIP_OK(hipMemcpy(*p, h.data(), h.size() * sizeof(U), hipMemcpyHostToDevice));
        return 0;
    }

    static bool is_tiny_head_conv(int id) { return id == SAYURI_L_PROB_CONV || id == SAYURI_L_V_OWNERSHIP; }

    bool heads_fused_enabled() const { return flags_.heads_fused; }
    // Stacked [policy | value] head-convolution image for head_board_kernel (fp16 engine, normal policy head).
    int build_head_image() {
        const sayuri_hip_netdesc& d = desc_;
        if (sizeof(T) != 2 || d.policy_head_type != 0) return 0;
        const ConvLayerDev& P = convs_.at(SAYURI_L_P_HD_CONV);
        const ConvLayerDev& V = convs_.at(SAYURI_L_V_HD_CONV);
        const ConvLayerDev& PW = convs_.at(SAYURI_L_PROB_CONV);
        const ConvLayerDev& OW = convs_.at(SAYURI_L_V_OWNERSHIP);
        if (P.hw.empty() || V.hw.empty() || P.hb.empty() || V.hb.empty() || PW.hw.empty() || OW.hw.empty()) return 0;
        HeadImages hi;
        head_fn_ = make_head_images(P.cin, P.cout, V.cout, d.probabilities_channels, board_, P.hw.data(), P.hb.data(), V.hw.data(),
                                    V.hb.data(), PW.hw.data(), OW.hw.data(), &hi);
        if (!head_fn_) return 0;
        f16 *w = nullptr, *w2 = nullptr;
        if (dev_upload(&w2, hi.img2) || dev_upload(&w, hi.img) || dev_upload(&head_bias_, hi.bias)) return -1;
        head_img2_ = w2;
        head_img_ = w;
        head_pt_ = hi.PT;
        head_vt_ = hi.VT;
        return 0;
    }

    int finalize() {
        if (finalized_) return 0;
        if (build_head_image()) return -1;
        for (auto& kv : convs_) {
            ConvLayerDev& L = kv.second;
            if (L.hw.empty() || L.hb.empty()) return fail("missing tensors for conv layer " + std::to_string(kv.first));
            L.cin_s = round_up(L.cin, 32);
            L.cout_s = round_up(L.cout, 32);
            if (is_tiny_head_conv(kv.first)) {  // consumed by head_tail_kernel in fp32
                if (dev_upload(&L.w32, L.hw) || dev_upload(&L.bias, L.hb)) return -1;
            } else if (L.depthwise) {
                const int kk = L.k * L.k;
                std::vector<float> wt((size_t)kk * L.cout_s, 0.f), b(L.cout_s, 0.f);
                for (int c = 0; c < L.cout; ++c) {
                    for (int t = 0; t < kk; ++t) wt[(size_t)t * L.cout_s + c] = L.hw[(size_t)c * kk + t];
                    b[c] = L.hb[c];
                }
                float* w = nullptr;
                if (dev_upload(&w, wt) || dev_upload(&L.bias, b)) return -1;
                L.w = w;
            } else {
                L.wmt = pick_wmt(L.cout_s, sizeof(T) == 2);
                const int kot = L.wmt * 32;
                L.ko_pad = round_up(L.cout_s, kot);
                const int taps = L.k * L.k, nch = L.cin_s / 32;
                std::vector<T> img((size_t)taps * nch * 4 * L.ko_pad * 8, from_float_host(0.f));
                for (int t = 0; t < taps; ++t)
                    for (int ch = 0; ch < nch; ++ch)
                        for (int kg = 0; kg < 4; ++kg)
                            for (int ko = 0; ko < L.cout; ++ko)
                                for (int e = 0; e < 8; ++e) {
                                    const int c = ch * 32 + kg * 8 + e;
                                    if (c >= L.cin) continue;
                                    const float v = L.hw[((size_t)ko * L.cin + c) * taps + t];
                                    img[((((size_t)t * nch + ch) * 4 + kg) * L.ko_pad + ko) * 8 + e] = from_float_host(v);
                                }
                std::vector<float> b(L.ko_pad, 0.f);
                std::copy(L.hb.begin(), L.hb.end(), b.begin());
                T* w = nullptr;
                if (dev_upload(&w, img) || dev_upload(&L.bias, b)) return -1;
                L.w = w;
                if (sizeof(T) == 2 && L.k == 3 && L.ko_pad % 128 == 0) {
                    T* wb = nullptr;
                    std::vector<float> bb(L.ko_pad);
                    for (int r = 0; r < L.ko_pad; ++r) bb[r] = b[board_row_channel(r)];
                    if (dev_upload(&wb, board_row_order(img, L.ko_pad)) || dev_upload(&L.bias_board, bb)) return -1;
                    L.w_board = wb;
                }
            }
            std::vector<float>().swap(L.hw);
            std::vector<float>().swap(L.hb);
        }
        if (build_se_images()) return -1;
        for (auto& kv : fcs_) {
            FcLayerDev& L = kv.second;
            if (L.hw.empty() || L.hb.empty()) return fail("missing tensors for fc layer " + std::to_string(kv.first));
            std::vector<float> wt((size_t)L.in * L.out);
            for (int o = 0; o < L.out; ++o)
                for (int i = 0; i < L.in; ++i) wt[(size_t)i * L.out + o] = L.hw[(size_t)o * L.in + i];
            if (dev_upload(&L.wt, wt) || dev_upload(&L.b, L.hb)) return -1;
            std::vector<float>().swap(L.hw);
            std::vector<float>().swap(L.hb);
        }
        // workspaces
        slot_pix_ = board_ * board_;
        const size_t act_elems = (size_t)max_batch_ * slot_pix_ * cs_max_;
        // the board kernels address an activation buffer with 24-bit row indices (__umul24) and 32-bit byte offsets from a
        // uniform base (conv_board.h epilogue): both must cover the largest batch this ctx can be handed
        if ((size_t)max_batch_ * slot_pix_ >= (size_t(1) << 24) || act_elems * sizeof(T) >= (size_t(1) << 32))
            return fail("max_batch " + std::to_string(max_batch_) + " is too large for this network on one ctx: max_batch * board^2 must stay below 2^24 rows and an activation buffer (" +
                        std::to_string(act_elems * sizeof(T) >> 20) + " MiB) below 4 GiB");
        for (IoSlot& io : io_)
            for (int i = 0; i < kNumBufs; ++i)
                // every activation buffer starts kZeroPrefix bytes into its (zero-filled) allocation: conv_board.h reads
                // its halo cells from that prefix
                if (dev_alloc(&io.bufs[i], act_elems + kZeroPrefix / sizeof(T))) return -1;
                else io.bufs[i] += kZeroPrefix / sizeof(T);
        const size_t B2 = (size_t)board_ * board_;
        for (IoSlot& io : io_) {
            if (dev_alloc(&io.planes, (size_t)max_batch_ * desc_.input_channels * B2)) return -1;
            if (dev_alloc(&io.packed, (size_t)max_batch_ * (40 * 12 + 8))) return -1;  // the packed form of the same planes (packed_planes.h)
            if (dev_alloc(&io.off, max_batch_ + 1) || dev_alloc(&io.bsz, max_batch_) || dev_alloc(&io.perm, max_batch_)) return -1;
            if (dev_alloc(&io.prob, (size_t)max_batch_ * desc_.probabilities_channels * B2)) return -1;
            if (dev_alloc(&io.pass, (size_t)max_batch_ * desc_.pass_probability_outputs)) return -1;
            if (dev_alloc(&io.misc, (size_t)max_batch_ * desc_.value_misc_outputs)) return -1;
            if (dev_alloc(&io.own, (size_t)max_batch_ * B2)) return -1;
        }
        if (dev_alloc(&d_zeros_, 64)) return -1;
        HIP_OK(hipHostMalloc((void**)&h_geom_, sizeof(int) * 2 * (3 * max_batch_ + 1), hipHostMallocDefault));
        for (IoSlot& io : io_) {
            if (dev_alloc(&io.gate, (size_t)max_batch_ * 2 * round_up(desc_.residual_channels, 32))) return -1;
            if (dev_alloc(&io.separt, (size_t)max_batch_ * kSeSplit * 2 * round_up(desc_.residual_channels, 32))) return -1;
            // conv_board_sx.h: the granules the sibling channel tiles of a board tile exchange, [tile][kt][sample][slot]
            if (sx_kts_ && dev_alloc(&io.sx_xchg, (size_t)max_batch_ * sx_kts_ * kSxMaxSub * kSxSlots)) return -1;
            io.sx_epoch = flags_.dbg_sx_epoch0;
        }
        if (sx_kts_) {
            HIP_OK(hipHostMalloc((void**)&sx_err_host_, 64, hipHostMallocMapped));
            std::memset(sx_err_host_, 0, 64);
            HIP_OK(hipHostGetDevicePointer((void**)&sx_err_dev_, sx_err_host_, 0));
        }
        finalized_ = true;
        select_slot(0);
        return 0;
    }

    static T from_float_host(float v) { return (T)v; }

    // fp16 images of the SE units' two FCs, laid out for the LDS-DMA staging of board_se_stage (conv_board.h): the squeeze
    // weights once per board size 2..board with the scaled-mean third of the pooled vector folded into the mean third
    // (reference GlobalPooling<false>, se_unit.cc:9-40: pool = (mean, mean * (B-14)/10, max)), the excite weights with
    // both bias vectors behind them.  Units whose images do not fit the LDS keep reading fp32 weights from L2.
    int build_se_images() {
        if (sizeof(T) != 2) return 0;
        const int C = desc_.residual_channels;
        for (int b = 0; b < desc_.residual_blocks; ++b) {
            if (!blocks_[b].apply_se) continue;
            FcLayerDev& sq = fcs_.at(SAYURI_L_BLOCK(b, SAYURI_S_SQUEEZE));
            FcLayerDev& ex = fcs_.at(SAYURI_L_BLOCK(b, SAYURI_S_EXCITE));
            const int se = sq.out;
            if (sq.hw.empty() || ex.hw.empty() || sq.hb.empty() || ex.hb.empty()) continue;  // finalize() reports it
            if (sq.in != 3 * C || ex.in != se || ex.out != 2 * C) continue;
            std::vector<f16> img1;
            std::vector<unsigned char> img2;
            int w1_bytes = 0, w2_bytes = 0;
            if (!make_se_images(C, se, board_, sq.hw.data(), sq.hb.data(), ex.hw.data(), ex.hb.data(), &img1, &img2, &w1_bytes, &w2_bytes)) {
                // a layer too wide for one workgroup: the per-channel-tile images of conv_board_sx.h
                const int kts = round_up(C, 128) / 128;
                if (flags_.se_split && round_up(C, 32) == kts * 128 &&
                    make_sx_images(C, se, kts, board_, sq.hw.data(), sq.hb.data(), ex.hw.data(), ex.hb.data(), &img1, &img2, &w1_bytes, &w2_bytes)) {
                    f16* d1 = nullptr;
                    unsigned char* d2 = nullptr;
                    if (dev_upload(&d1, img1) || dev_upload(&d2, img2)) return -1;
                    sq.sx_img = d1; sq.sx_bytes = w1_bytes;
                    ex.sx_img = d2; ex.sx_bytes = w2_bytes;
                    sx_kts_ = kts;
                }
                continue;
            }
            f16* d1 = nullptr;
            unsigned char* d2 = nullptr;
            if (dev_upload(&d1, img1) || dev_upload(&d2, img2)) return -1;
            sq.img16 = d1; sq.img_bytes = w1_bytes;
            ex.img16 = d2; ex.img_bytes = w2_bytes;
        }
        return 0;
    }

    void release() {
        (void)hipSetDevice(device_);
        for (void* p : allocs_) (void)hipFree(p);
        allocs_.clear();
        ident_.clear();
        if (h_geom_) (void)hipHostFree(h_geom_);
        h_geom_ = nullptr;
        if (sx_err_host_) (void)hipHostFree(sx_err_host_);
        sx_err_host_ = nullptr;
        sx_err_dev_ = nullptr;
        for (TowerSlot& ts : tower_)
            for (int i = 0; i < 2; ++i) {
                if (ts.stage[i]) (void)hipHostFree(ts.stage[i]);
                if (ts.staged[i]) (void)hipEventDestroy(ts.staged[i]);
                ts.stage[i] = nullptr; ts.staged[i] = nullptr;
            }
        if (tower_mod_) (void)hipModuleUnload(tower_mod_);
        tower_mod_ = nullptr;
        for (hipStream_t& cs : chain_stream_) { if (cs) (void)hipStreamDestroy(cs); cs = nullptr; }
        for (hipEvent_t& e : chain_join_) { if (e) (void)hipEventDestroy(e); e = nullptr; }
        if (chain_fork_) (void)hipEventDestroy(chain_fork_);
        chain_fork_ = nullptr;
        if (fwdstat_ && fs_cnt_[0] + fs_cnt_[1] + fs_cnt_[2] > 0) {
            const long all = fs_cnt_[0] + fs_cnt_[1] + fs_cnt_[2];
            std::fprintf(stderr, "[hip fwdstat] forwards by batch size (<= half | partial | full): %ld / %ld / %ld, mean device ms %.4f / %.4f / %.4f, tower table uploads %ld; "
                         "forward end -> downloads done: mean %.3f ms, max %.3f, > 1 ms: %ld\n",
                         fs_cnt_[0], fs_cnt_[1], fs_cnt_[2], fs_cnt_[0] ? fs_ms_[0] / fs_cnt_[0] : 0.0, fs_cnt_[1] ? fs_ms_[1] / fs_cnt_[1] : 0.0,
                         fs_cnt_[2] ? fs_ms_[2] / fs_cnt_[2] : 0.0, fs_uploads_, fs_d2h_ms_ / all, fs_d2h_max_, fs_d2h_slow_);
            fs_cnt_[0] = fs_cnt_[1] = fs_cnt_[2] = 0;
        }
        for (auto& pr : fs_ev_) for (hipEvent_t& e : pr) { if (e) (void)hipEventDestroy(e); e = nullptr; }
        for (hipEvent_t& e : tick_ev_) { if (e) (void)hipEventDestroy(e); e = nullptr; }
        for (hipEvent_t e : pool_) (void)hipEventDestroy(e);
        pool_.clear();
        if (ev0_) (void)hipEventDestroy(ev0_);
        if (ev1_) (void)hipEventDestroy(ev1_);
        for (hipEvent_t& e : h2d_done_) { if (e) (void)hipEventDestroy(e); e = nullptr; }
        for (hipEvent_t& e : fwd_done_) { if (e) (void)hipEventDestroy(e); e = nullptr; }
        stream_ = compute_[0];
        if (compute_[1] && compute_[1] != compute_[0]) (void)hipStreamDestroy(compute_[1]);
        compute_[1] = nullptr;
        if (stream_) (void)hipStreamDestroy(stream_);
        if (h2d_stream_) (void)hipStreamDestroy(h2d_stream_);
        if (d2h_stream_) (void)hipStreamDestroy(d2h_stream_);
        ev0_ = ev1_ = nullptr;
        stream_ = h2d_stream_ = d2h_stream_ = nullptr;
    }

    // -------------------------------------------------------------- launch plumbing
    BatchGeom dgeom() const { return BatchGeom{d_off_, d_bsz_, geom_.n, geom_.total, slot_pix_}; }

    int close_group() {
        HIP_OK(hipEventRecord(pool_[pool_used_ + 1], stream_));
        pool_used_ += 2;
        group_open_ = false;
        return 0;
    }
    template <typename F> int timed(const char* name, double flops, double bytes, F&& launch) {
        if (!run_.empty() && tower_flush()) return -1;  // the pending run of board convolutions goes first (stream order)
        if (!profiling_) {
            // light mode: un-synchronised event pairs around runs of the dominant kernel class only
            const bool match = light_ && light_name_ == name;
            if (group_open_ && (!match || group_counts_.back() >= light_group_)) {
                if (close_group()) return -1;
            }
            if (match && !group_open_) {
                if (pool_used_ + 2 > pool_.size()) {
                    for (int i = 0; i < 64; ++i) {
                        hipEvent_t e;
                        HIP_OK(hipEventCreate(&e));
                        pool_.push_back(e);
                    }
                }
                HIP_OK(hipEventRecord(pool_[pool_used_], stream_));
                group_counts_.push_back(0);
                group_open_ = true;
            }
            launch();
            HIP_OK(hipGetLastError());
            if (rg_ntiles_ < 0) rows_reset();  // a launch on the forward's one stream: ordered against everything behind it
            if (match) {
                group_counts_.back() += 1;
                light_flops_ = flops;
                light_bytes_ = bytes;
            }
            return 0;
        }
        HIP_OK(hipEventRecord(ev0_, stream_));
        launch();
        HIP_OK(hipGetLastError());
        rows_reset();
        HIP_OK(hipEventRecord(ev1_, stream_));
        HIP_OK(hipEventSynchronize(ev1_));
        float ms = 0.f;
        HIP_OK(hipEventElapsedTime(&ms, ev0_, ev1_));
        Stat& s = stats_[name];
        s.launches += 1;
        s.ms += ms;
        s.flops += flops;
        s.bytes += bytes;
        return 0;
    }

    struct TileChoice { int wnt, npos, ntiles; const typename ConvKernelTable<T>::Entry* e; };

    int choose_tile(int wmt, int kot_tiles, TileChoice* out) {
        auto it = tile_cache_.find(wmt * 1024 + kot_tiles);
        if (it != tile_cache_.end()) { *out = it->second; return 0; }
        double best_cost = 1e30;
        TileChoice best{};
        bool found = false;
        for (const auto& e : ConvKernelTable<T>::entries()) {
            if (e.wmt != wmt) continue;
            const int PT = 64 * e.wnt;
            int npos, nsub;
            geom_.tile_bounds(PT, &npos, &nsub);
            if (npos > e.npos_cap || nsub > kMaxSub || e.lds(npos) > kMaxLds) continue;
            const int ntiles = (geom_.total + PT - 1) / PT;
            const double waves = std::ceil((double)ntiles * kot_tiles / kNumCU);
            const double cost = waves * (PT + 24);
            if (cost < best_cost) { best_cost = cost; best = TileChoice{e.wnt, npos, ntiles, &e}; found = true; }
        }
        if (!found) return fail("no conv tile configuration fits this batch geometry");
        tile_cache_[wmt * 1024 + kot_tiles] = best;
        *out = best;
        return 0;
    }

    struct GldsChoice { const GldsEntry* e; int ntiles; };
    // index tables of the current batch geometry for pixel-tile size 64*wnt (built on first use)
    int tile_tabs(const GldsEntry& e, const TileTabs** out) {
        TileTabs& t = io_[cur_slot_].tabs[e.wnt];
        if (!t.src) {
            const size_t max_tiles = ((size_t)max_batch_ * slot_pix_ + e.pt - 1) / e.pt;
            if (dev_alloc(&t.src, max_tiles * e.npos) || dev_alloc(&t.pix, max_tiles * e.pt)) return -1;
        }
        if (!t.fresh) {
            const int ntiles = (geom_.total + e.pt - 1) / e.pt;
            hipLaunchKernelGGL(e.setup, dim3(ntiles), dim3(256), 0, stream_, dgeom(), t.src, t.pix);
            HIP_OK(hipGetLastError());
            t.fresh = true;
        }
        *out = &t;
        return 0;
    }
    const GldsChoice* choose_glds(const ConvLayerDev& L) {
        if (sizeof(T) != 2 || L.k != 3 || L.ko_pad % 128 != 0) return nullptr;
        const int key = L.ko_pad % 256 == 0 ? 8 : 4;
        auto it = glds_cache_.find(key);
        if (it == glds_cache_.end()) {
            GldsChoice c{nullptr, 0};
            c.e = pick_glds(geom_, L.ko_pad, &c.ntiles, flags_.conv);
            it = glds_cache_.emplace(key, c).first;
        }
        return it->second.e ? &it->second : nullptr;
    }

    // index tables of the current batch geometry for conv_board_kernel (built on first use)
    int board_tabs(const BoardTabs** out) {
        BoardTabs& t = io_[cur_slot_].board;
        if (!t.src) {
            // a tile holds at least one sample
            if (dev_alloc(&t.src, (size_t)max_batch_ * kBoardMaxPos) || dev_alloc(&t.pix, (size_t)max_batch_ * kBoardPT) ||
                dev_alloc(&t.cols, max_batch_))
                return -1;
        }
        // (a slot's tables may be kept for a shorter batch of the same geometry -- enqueue_inputs' prefix rule trusts the
        // sizes recorded at enqueue time; what counts here is how many tiles the tables were BUILT for)
        if (!t.fresh || t.npos_built != board_plan_.npos || board_plan_.ntiles > t.ntiles_built) {
            hipLaunchKernelGGL(board_setup_kernel, dim3(board_plan_.ntiles), dim3(256), 0, stream_, dgeom(), board_plan_.npos, t.src,
                               t.pix, t.cols);
            HIP_OK(hipGetLastError());
            t.fresh = true;
            t.npos_built = board_plan_.npos;
            t.ntiles_built = board_plan_.ntiles;
        }
        *out = &t;
        return 0;
    }
    // the one-workgroup-per-board kernel applies to fp16 3x3 layers whose boards fit a tile and fill it reasonably
    const BoardEntry* choose_board(const ConvLayerDev& L, int* kot_tiles) {
        if (sizeof(T) != 2 || L.k != 3) return nullptr;
        if (!board_plan_valid_) { board_plan_ = board_plan(geom_, flags_.conv); board_plan_valid_ = true; }
        if (!board_plan_.ok || board_plan_.fill < flags_.conv.board_min_fill) return nullptr;
        return pick_board(board_plan_, L.ko_pad, kot_tiles, flags_.board_kot);
    }

    // Does this layer of the persistent launch get the generated epilogue (tower_seam.py epi_hook)?  Then its weights and bias go
    // in board_row_channel order and BoardParams::row_order says so.  What the generated text covers: Mish / ReLU / no activation, one sample per tile
    // with computed table entries (arith), the layer's channels = the channel tile, an even number of row tiles per wave.
    bool board_row_order_ok(const ConvLayerDev& L, const BoardEntry* be, const BoardParams& bp, int act) const {
        return flags_.tower_gen_epi && tower_ok(be->kot) && !bp.dbg && board_uses_row_order(be->kot) && bp.arith &&
               (act == kMish || act == kReLU || act == kIdentity) &&
               L.cout_s == be->kot && L.ko_pad == be->kot && L.w_board && L.bias_board;
    }

    // A block's last 3x3 convolution with the squeeze-and-excitation unit that follows it inside the kernel
    // (conv_board.h).  Returns 1 when the fused kernel does not apply (the caller then runs conv + se_unit), 0 / -1.
    int conv_se(const ConvLayerDev& L, const FcLayerDev& sq, const FcLayerDev& ex, const T* in, T* out, const T* res, int C, int act) {
        const bool off = !flags_.se_fused;
        int bkt = 0;
        const BoardEntry* be = nullptr;
        // the variant whose channel tile covers the whole layer, whatever the batch size: a position's result must not
        // depend on how many others share its batch (a small batch would otherwise pick two half-width workgroups and
        // the separate SE kernels, which round x to fp16 before pooling)
        if (!off && choose_board(L, &bkt))
            for (const auto& e : kBoardEntries)
                if (e.fn_se && e.kot == L.ko_pad && e.lds(board_plan_.npos) <= kMaxLds) be = &e;
        if (!be || C > be->kot) return 1;
        const bool staged = sq.img16 && ex.img16;
        if (!staged && (sq.out % 4 || sq.out > 512 || ex.out % 4 || ex.out > 2048 || 512 % (sq.out / 4) || 512 % (ex.out / 4))) return 1;
        if constexpr (sizeof(T) != 2) return 1;
        // WHICH samples take the fused form is a property of the sample alone, never of its batch mates (a position's result
        // must not depend on what else the queue collected: the fused form pools the fp32 accumulators, the separate kernels
        // pool x rounded to fp16): a board too large to share a tile with another of its size (2 bs^2 > 384 pixel slots, i.e.
        // bs >= 14) is ALWAYS alone in its tile and ALWAYS fused; a smaller board ALWAYS goes through the separate kernels, also
        // when it happens to sit alone in a tile.  The device order is largest first, so the fused samples -- and their tiles,
        // one each -- lead the batch: tiles [0, nbig) fused, tiles [nbig, ntiles) = samples [nbig, n) plain convolution + SE unit.
        int nbig = 0;
        while (nbig < geom_.n && 2 * geom_.bsz[nbig] * geom_.bsz[nbig] > kBoardPT) ++nbig;
        if (!flags_.se_by_geometry) nbig = board_plan_.single ? geom_.n : 0;  // SAYURI_SE_BY_GEOMETRY=0: round 4's rule (A/B, tests)
        if (nbig == 0) return 1;
        const bool split = nbig < geom_.n;
        const BoardTabs* tabs = nullptr;
        if (board_tabs(&tabs)) return -1;
        BoardSeParams sp;
        std::memset(&sp, 0, sizeof(sp));  // padding too: the tower table is compared bytewise with its cached copy
        BoardParams& bp = sp.b;
        bp.tab_src = tabs->src; bp.tab_pix = tabs->pix; bp.tab_cols = tabs->cols; bp.npos = board_plan_.npos; bp.dbg = nullptr;
        bp.uniform_info = board_plan_.uniform_info;
        bp.arith = (board_plan_.single && board_plan_.uniform_info >= 0 && flags_.arith) ? 1 : 0;
        ConvParams& p = bp.c;
        p.in = in; p.w = L.w; p.bias = L.bias; p.res = res; p.out = out;
        p.g = dgeom();
        p.cin_s = L.cin_s; p.cout_s = L.cout_s; p.ko_pad = L.ko_pad;
        p.taps = 9; p.act = act; p.npos = 0; p.num_pix_tiles = board_plan_.ntiles;
        sp.squeeze = sq.dev(); sp.excite = ex.dev(); sp.C = C;
        sp.w1h = staged ? sq.img16 : nullptr; sp.w2h = staged ? ex.img16 : nullptr;
        sp.w1_bytes = sq.img_bytes; sp.w2_bytes = ex.img_bytes;
#ifdef SAYURI_EXPERIMENTS
        if (flags_.board_dbg < 0) {  // negative n: timeline of the n-th SE convolution of the forward
            if (!d_dbg_ && dev_alloc(&d_dbg_, 4 * 8 * 8)) return -1;
            if (++dbg_se_call_ == -flags_.board_dbg) { bp.dbg = d_dbg_; dbg_is_se_ = true; }
        }
#endif
        const double px = geom_.total;
        const double flops = 2.0 * px * L.cin * L.cout * 9 + 2.0 * geom_.n * ((double)sq.in * sq.out + (double)ex.in * ex.out);
        const double bytes = sizeof(T) * (px * L.cin + px * L.cout * (res ? 2 : 1) + (double)L.cin * L.cout * 9);
        if (board_row_order_ok(L, be, bp, act)) { p.w = L.w_board; p.bias = L.bias_board; bp.row_order = 1; }
        const bool to_run = !split && tower_ok(be->kot) && !bp.dbg;
        if (!run_.empty() && !(to_run && run_kot_ == be->kot) && tower_flush()) return -1;
        if (rows_use(in, L.cin_s, "conv3x3_tower_se") || rows_use(out, L.cout_s, "conv3x3_tower_se") || rows_use(res, L.cout_s, "conv3x3_tower_se"))
            return -1;
        if (to_run) return tower_append(be->kot, sp, true, flops, bytes);
        const auto fn = be->fn_se;
        const size_t lds = be->lds(board_plan_.npos);
        const int grid = split ? nbig : board_plan_.ntiles;
        if (timed("conv3x3_tower_se", flops, bytes, [&] { hipLaunchKernelGGL(fn, dim3(grid), dim3(512), lds, stream_, sp); })) return -1;
        if (!split) return 0;
        // the small boards of the batch: the same convolution without epilogue extras on the tiles behind, then the unit's
        // three kernels on the samples behind
        BoardParams rest = bp;
        rest.row_order = 0;
        rest.c.w = L.w; rest.c.bias = L.bias; rest.c.res = nullptr; rest.c.act = kIdentity;
        rest.c.npos = nbig;  // first tile of the launch (conv_board_kernel)
        rest.c.num_pix_tiles = board_plan_.ntiles - nbig;
        const auto fn2 = be->fn;
        const int grid2 = rest.c.num_pix_tiles;  // be->kot covers the layer: one channel tile
        if (timed("conv3x3_tower", flops, bytes, [&] { hipLaunchKernelGGL(fn2, dim3(grid2), dim3(512), lds, stream_, rest); })) return -1;
        return se_unit(sq, ex, out, res, C, round_up(C, 32), act, nbig);
    }


    // A block's last 3x3 convolution with its SE unit when the layer's channels are split over kts = 2..4 workgroups of 128
    // (conv_board_sx.h: the siblings exchange their partial squeeze sums inside the launch).  Returns 1 when the form does not apply
    // (the caller runs conv + se_unit), 0 / -1.  WHICH samples take it is a property of the sample alone: a board of which at
    // most kSxMaxSub fit a tile (9x9 and larger) ALWAYS does, a smaller one NEVER -- the device order is largest first, so the
    // tiles [0, T) of the batch are fused and [T, ntiles) = the samples behind take the plain convolution + the unit's three
    // kernels, exactly as conv_se splits a mixed batch.  Honours the tile range of a chained forward.
    int conv_sx(const ConvLayerDev& L, const FcLayerDev& sq, const FcLayerDev& ex, const T* in, T* out, const T* res, int C, int act) {
        if constexpr (sizeof(T) != 2) return 1;
        if (!flags_.se_split || sx_disabled_ || !sq.sx_img || !ex.sx_img || !sx_kts_ || L.ko_pad != sx_kts_ * 128 || L.cout_s != L.ko_pad) return 1;
        int bkt = 0;
        if (!choose_board(L, &bkt)) return 1;
        const BoardEntry* be = nullptr;
        for (const auto& e : kBoardEntries)
            if (e.kot == 128 && e.lds(board_plan_.npos) <= kMaxLds) be = &e;
        if (!be) return 1;
        auto per_tile = [](int bs) {
            BoardPack pk;
            int k = 0;
            while (pk.fits(bs)) { pk.add(bs); ++k; }
            return k;
        };
        int nf = 0;
        while (nf < geom_.n && per_tile(geom_.bsz[nf]) <= kSxMaxSub) ++nf;
        int T0 = 0;  // first tile that is not fused
        while (T0 < board_plan_.ntiles && board_plan_.tile_first[T0] < nf) ++T0;
        const int t0 = rg_tile0_, t1 = rg_tile0_ + range_ntiles();
        const int f0 = t0, f1 = std::min(t1, T0), r0 = std::max(t0, T0), r1 = t1;
        if (!run_.empty() && tower_flush()) return -1;
        const BoardTabs* tabs = nullptr;
        if (board_tabs(&tabs)) return -1;
        if (rows_use(in, L.cin_s, "conv3x3_tower_sx") || rows_use(out, L.cout_s, "conv3x3_tower_sx") || rows_use(res, L.cout_s, "conv3x3_tower_sx"))
            return -1;
        const unsigned epoch = sx_epoch0_ + 1u + (unsigned)sx_idx_++;
        BoardSxParams sp;
        std::memset(&sp, 0, sizeof(sp));
        BoardParams& bp = sp.b;
        bp.tab_src = tabs->src; bp.tab_pix = tabs->pix; bp.tab_cols = tabs->cols; bp.npos = board_plan_.npos; bp.dbg = nullptr;
        bp.uniform_info = board_plan_.uniform_info;
        bp.arith = (board_plan_.single && board_plan_.uniform_info >= 0 && flags_.arith) ? 1 : 0;
        ConvParams& p = bp.c;
        p.in = in; p.w = L.w; p.bias = L.bias; p.res = res; p.out = out;
        p.g = dgeom();
        p.cin_s = L.cin_s; p.cout_s = L.cout_s; p.ko_pad = L.ko_pad;
        p.taps = 9; p.act = act;
        const size_t lds = be->lds(board_plan_.npos);
        const int kts = sx_kts_;
        const double px_all = range_px();
        if (f1 > f0) {
            p.npos = f0; p.num_pix_tiles = f1 - f0;
            sp.w1t = sq.sx_img; sp.w2t = ex.sx_img; sp.w1_bytes = sq.sx_bytes; sp.w2_bytes = ex.sx_bytes;
            sp.nsizes = board_ - 1; sp.se = sq.out; sp.kts = kts;
            sp.xchg = io_[cur_slot_].sx_xchg; sp.epoch = epoch; sp.err = sx_err_dev_;
            sp.dbg_stall = flags_.dbg_sx_stall ? 1 : 0;
            if (flags_.sx_dbg > 0 && profiling_ && sx_idx_ == flags_.sx_dbg) {
                if (!d_sxdbg_ && dev_alloc(&d_sxdbg_, 4 * 64)) return -1;
                sp.b.dbg = d_sxdbg_;
            }
            const int s0 = board_plan_.tile_first[f0], s1 = board_plan_.tile_first[f1];
            const double px = (double)(geom_.off[s1] - geom_.off[s0]);
            const double flops = 2.0 * px * L.cin * L.cout * 9 + 2.0 * (s1 - s0) * ((double)sq.in * sq.out + (double)ex.in * ex.out);
            const double bytes = sizeof(T) * (px * L.cin + px * L.cout * (res ? 2 : 1) + (double)L.cin * L.cout * 9);
            const int grid = (f1 - f0 + 7) / 8 * 8 * kts;
            if (timed("conv3x3_tower_sx", flops, bytes, [&] { hipLaunchKernelGGL(conv_board_sx_kernel<2>, dim3(grid), dim3(512), lds, stream_, sp); }))
                return -1;
        }
        if (r1 > r0) {
            // the small boards behind: the plain convolution (no activation, no residual) on their tiles, then the unit's kernels
            BoardParams rest = bp;
            rest.c.res = nullptr; rest.c.act = kIdentity;
            rest.c.npos = r0; rest.c.num_pix_tiles = r1 - r0;
            const int s0 = board_plan_.tile_first[r0], s1 = board_plan_.tile_first[r1];
            const double px = (double)(geom_.off[s1] - geom_.off[s0]);
            const double flops = 2.0 * px * L.cin * L.cout * 9;
            const double bytes = sizeof(T) * (px * L.cin + px * L.cout + (double)L.cin * L.cout * 9);
            const auto fn2 = be->fn;
            const int grid2 = (r1 - r0) * kts;
            if (timed("conv3x3_tower", flops, bytes, [&] { hipLaunchKernelGGL(fn2, dim3(grid2), dim3(512), lds, stream_, rest); })) return -1;
            if (se_unit(sq, ex, out, res, C, round_up(C, 32), act, s0, s1 - s0)) return -1;
        }
        (void)px_all;
        return 0;
    }

    int conv(const char* name, const ConvLayerDev& L, const T* in, T* out, const T* res, int act) {
        int bkt = 0;
        if (const BoardEntry* be = choose_board(L, &bkt)) {
            const BoardTabs* tabs = nullptr;
            if (board_tabs(&tabs)) return -1;
            BoardParams bp;
            std::memset(&bp, 0, sizeof(bp));
            bp.tab_src = tabs->src; bp.tab_pix = tabs->pix; bp.tab_cols = tabs->cols; bp.npos = board_plan_.npos;
            bp.dbg = nullptr;
            bp.uniform_info = board_plan_.uniform_info;
        bp.arith = (board_plan_.single && board_plan_.uniform_info >= 0 && flags_.arith) ? 1 : 0;
            auto fn = be->fn;
#ifdef SAYURI_EXPERIMENTS
            if (be->kot == 256 && flags_.board_dbg > 0 && !strcmp(name, "conv3x3_tower")) {
                // in-kernel timeline of the SAYURI_BOARD_DBG-th tower convolution of the forward (1 = first)
                if (!d_dbg_ && dev_alloc(&d_dbg_, 4 * 8 * 8)) return -1;
                if (++dbg_call_ == flags_.board_dbg) {
                    bp.dbg = d_dbg_;
                    fn = &conv_board_kernel<4, true>;
                }
            }
#endif
            ConvParams& p = bp.c;
            p.in = in; p.w = L.w; p.bias = L.bias; p.res = res; p.out = out;
            p.g = dgeom();
            p.cin_s = L.cin_s; p.cout_s = L.cout_s; p.ko_pad = L.ko_pad;
            p.taps = 9; p.act = act; p.npos = rg_tile0_; p.num_pix_tiles = range_ntiles();  // (npos: the launch's first tile, conv_board_kernel)
#ifdef SAYURI_EXPERIMENTS
            if (flags_.act_override >= 0) p.act = flags_.act_override;  // timing experiments only
#endif
            const double px = range_px();
            const double flops = 2.0 * px * L.cin * L.cout * 9;
            const double bytes = sizeof(T) * (px * L.cin + px * L.cout * (res ? 2 : 1) + (double)L.cin * L.cout * 9);
            const bool to_run = bkt == 1 && tower_ok(be->kot) && !bp.dbg && rg_ntiles_ < 0;
            // a pending run this layer does not join ends here (a launch = an ordered point: the table starts afresh)
            if (!run_.empty() && !(to_run && run_kot_ == be->kot) && tower_flush()) return -1;
            if (rows_use(in, L.cin_s, name) || rows_use(out, L.cout_s, name) || rows_use(res, L.cout_s, name)) return -1;
            if (to_run) {
                if (board_row_order_ok(L, be, bp, p.act)) { p.w = L.w_board; p.bias = L.bias_board; bp.row_order = 1; }
                BoardSeParams sp;
                std::memset(&sp, 0, sizeof(sp));
                sp.b = bp;
                return tower_append(be->kot, sp, false, flops, bytes);
            }
            const size_t lds = be->lds(board_plan_.npos);
            const int grid = range_ntiles() * bkt;
            return timed(name, flops, bytes, [&] { hipLaunchKernelGGL(fn, dim3(grid), dim3(512), lds, stream_, bp); });
        }
        if (rg_ntiles_ >= 0) return fail(std::string("chained forward: layer ") + name + " has no board kernel");
        if (const GldsChoice* gc = choose_glds(L)) {
            const TileTabs* tabs = nullptr;
            if (tile_tabs(*gc->e, &tabs)) return -1;
            GldsParams gp;
            gp.tab_src = tabs->src;
            gp.tab_pix = tabs->pix;
            ConvParams& p = gp.c;
            p.in = in; p.w = L.w; p.bias = L.bias; p.res = res; p.out = out;
            p.g = dgeom();
            p.cin_s = L.cin_s; p.cout_s = L.cout_s; p.ko_pad = L.ko_pad;
            p.taps = 9; p.act = act; p.npos = 0; p.num_pix_tiles = gc->ntiles;
            gp.zeros = d_zeros_;
            const double px = geom_.total;
            const double flops = 2.0 * px * L.cin * L.cout * 9;
            const double bytes = sizeof(T) * (px * L.cin + px * L.cout * (res ? 2 : 1) + (double)L.cin * L.cout * 9);
            const auto fn = gc->e->fn;
            const size_t lds = gc->e->lds;
            const int grid = gc->ntiles * (L.ko_pad / (gc->e->wmt * 32));
            return timed(name, flops, bytes, [&] { hipLaunchKernelGGL(fn, dim3(grid), dim3(512), lds, stream_, gp); });
        }
        const int kot = L.wmt * 32, kot_tiles = L.ko_pad / kot;
        TileChoice tc;
        if (choose_tile(L.wmt, kot_tiles, &tc)) return -1;
        ConvParams p;
        p.in = in; p.w = L.w; p.bias = L.bias; p.res = res; p.out = out;
        p.g = dgeom();
        p.cin_s = L.cin_s; p.cout_s = L.cout_s; p.ko_pad = L.ko_pad;
        p.taps = L.k * L.k; p.act = act; p.npos = tc.npos; p.num_pix_tiles = tc.ntiles;
        const double px = geom_.total;
        const double flops = 2.0 * px * L.cin * L.cout * p.taps;
        const double bytes = sizeof(T) * (px * L.cin + px * L.cout * (res ? 2 : 1) + (double)L.cin * L.cout * p.taps);
        const auto fn = tc.e->fn;
        const size_t lds = tc.e->lds(tc.npos);
        const int grid = tc.ntiles * kot_tiles;
        return timed(name, flops, bytes, [&] { hipLaunchKernelGGL(fn, dim3(grid), dim3(512), lds, stream_, p); });
    }

    int depthwise(const char* name, const ConvLayerDev& L, const T* in, T* out, const T* res, int act) {
        const int EPP = ElemTraits<T>::kPieceElems;
        const size_t total = (size_t)geom_.total * (L.cout_s / EPP);
        const int grid = (int)((total + 255) / 256);
        const double px = geom_.total;
        const BatchGeom g = dgeom();
        return timed(name, 2.0 * px * L.cout * L.k * L.k, sizeof(T) * px * L.cout * (res ? 3 : 2), [&] {
            hipLaunchKernelGGL(depthwise_kernel<T>, dim3(grid), dim3(256), 0, stream_, in, res, out,
                               (const float*)L.w, (const float*)L.bias, g, L.cout, L.cout_s, L.k, act);
        });
    }

    // n0: the unit runs on the samples [n0, n) of the batch (conv_se's split of a mixed batch)
    int se_unit(const FcLayerDev& sq, const FcLayerDev& ex, T* x, const T* res, int C, int cs, int act, int n0 = 0, int count = -1) {
        const BatchGeom g = dgeom();
        const int ns = count >= 0 ? count : geom_.n - n0;
        const double px = geom_.off[n0 + ns] - geom_.off[n0];
        constexpr int EPP = ElemTraits<T>::kPieceElems;
        if (cs / EPP > 256) return fail("SE unit: more than 256*8 channels is not supported");
        if (timed("se_pool", 2.0 * px * C, sizeof(T) * px * C, [&] {
                hipLaunchKernelGGL(se_pool_kernel<T>, dim3(ns * kSeSplit), dim3(256), 0, stream_, (const T*)x,
                                   d_separt_, g, cs, n0);
            }))
            return -1;
        const size_t smem = sizeof(float) * (3 * C + sq.out + kSeFcThreads);
        if (timed("se_fc", 2.0 * ns * ((double)sq.in * sq.out + (double)ex.in * ex.out),
                  4.0 * ns * ((double)sq.in * sq.out + (double)ex.in * ex.out), [&] {
                      hipLaunchKernelGGL(se_fc_kernel, dim3(ns), dim3(kSeFcThreads), smem, stream_,
                                         (const float*)d_separt_, d_gate_, g, C, cs, sq.dev(), ex.dev(), act, n0);
                  }))
            return -1;
        const int ppr = cs / EPP;
        const dim3 grid((slot_pix_ * ppr + 256 * kScaleUnroll - 1) / (256 * kScaleUnroll), ns);
        return timed("se_scale", 3.0 * px * C, sizeof(T) * px * C * 3, [&] {
            hipLaunchKernelGGL(se_scale_kernel<T>, grid, dim3(256), 0, stream_, (const T*)x, res, x,
                               (const float*)d_gate_, g, C, cs, act, n0);
        });
    }

    // small pool of activation buffers
    static constexpr int kNumBufs = 6;
    // -------------------------------------------------------------- who may write which bytes when (the buffer table)
    // Launches on one stream are ordered grid-wide; the layers INSIDE a persistent tower run are not (a workgroup walks the
    // whole run on its own clock), and neither are the chains of a chained forward (streams of their own).  In such an
    // UNORDERED SCOPE the only order is "this workgroup's / this chain's earlier layers", so a buffer may be written while
    // other workgroups still have to read it -- sound exactly when the bytes a tile touches in a buffer are the same in every
    // layer of the scope, i.e. when the buffer has ONE ROW STRIDE throughout it (a tile is whole samples, a sample's rows
    // start at sample * slot_pix * stride):
    //
    //   buffer                    written by                     row stride              may be recycled
    //   ------------------------  -----------------------------  ----------------------  ------------------------------------------
    //   `in` (packed input)       pack_bits / pack_input         input conv's cin_s (64) never inside the forward (kept to its end)
    //   pool buffers x, y, t0...  the convolution they are `out` the tower's cout_s      as soon as the layer that reads them is
    //                             of (epilogue, own tile's rows)  (256 / 384 / 128)        appended: give() -> take(), SAME stride only
    //   d_separt_, d_gate_        se_pool / se_fc, per sample     per-sample records      per sample, inside its chain
    //   se_xchg (384-ch SE)       the sibling channel tiles       per (tile, channel tile) next SE layer (tagged with the layer's epoch)
    //   d_prob_ ... d_own_        the heads kernel                per sample              by the ticket's next submit
    //
    // rounds 3-4 broke the first line (the input's buffer went back to the pool and came out again as a block's output with
    // stride 256: a late workgroup's input lay under an early workgroup's third layer).  The table below is that rule as a
    // run-time check: every use of a pool buffer inside an unordered scope names its stride, and a second stride is refused.
    int rows_stride_[kNumBufs] = {};          // 0: not used yet in the current scope
    const char* rows_first_[kNumBufs] = {};   // the layer that fixed it
    void rows_reset() {
        for (int i = 0; i < kNumBufs; ++i) rows_stride_[i] = 0;
    }
    int rows_use(const T* p, int stride, const char* layer) {
        if (!p || flags_.dbg_recycle_input >= 2) return 0;
        for (int i = 0; i < kNumBufs; ++i) {
            if (bufs_[i] != p) continue;
            if (rows_stride_[i] && rows_stride_[i] != stride)
                return fail(std::string("activation buffer ") + std::to_string(i) + " is used with row stride " + std::to_string(stride) + " by " +
                            layer + " and with " + std::to_string(rows_stride_[i]) + " by " + (rows_first_[i] ? rows_first_[i] : "?") +
                            " inside one persistent run / chained forward: the rows of different tiles would overlap");
            rows_stride_[i] = stride;
            rows_first_[i] = layer;
        }
        return 0;
    }
    int take() {
        for (int i = 0; i < kNumBufs; ++i)
            if (!busy_[i]) { busy_[i] = true; return i; }
        return -1;
    }
    void give(int i) { busy_[i] = false; }

    const ConvLayerDev& cv(int id) const { return convs_.at(id); }
    const FcLayerDev& fc(int id) const { return fcs_.at(id); }

    // -------------------------------------------------------------- the graph
    // One forward of the current batch: as ONE chain of launches on stream_, or -- chains_for_batch() -- as G chains over G
    // ranges of tiles on G streams, forked from and joined to stream_ by events (the activations, tables and outputs of the
    // ranges are disjoint: a tile is whole samples).
    int forward() {
        const int G = chains_for_batch();
        last_chains_ = G;
        rows_reset();
        if (sx_kts_) {
            IoSlot& io = io_[cur_slot_];
            int nse = 0;
            for (const auto& b : blocks_) nse += b.apply_se ? 1 : 0;
            if (io.sx_epoch > 0xfff00000u) {  // tags about to wrap: start over on a clean buffer (stream order: behind the last readers)
                HIP_OK(hipMemsetAsync(io.sx_xchg, 0, sizeof(unsigned long long) * (size_t)max_batch_ * sx_kts_ * kSxMaxSub * kSxSlots, stream_));
                io.sx_epoch = 0;
            }
            sx_epoch0_ = io.sx_epoch;
            io.sx_epoch += (unsigned)nse;
        }
        if (G <= 1) return forward_graph();
        if (chain_setup(G)) return -1;
        const BoardTabs* tabs = nullptr;
        if (board_tabs(&tabs)) return -1;  // built on stream_, in front of the fork
        hipStream_t main = stream_;
        HIP_OK(hipEventRecord(chain_fork_, main));
        int rc = 0;
        const int nt = board_plan_.ntiles;
        for (int g = 0; g < G && rc == 0; ++g) {
            rg_tile0_ = (int)((long)nt * g / G);
            rg_ntiles_ = (int)((long)nt * (g + 1) / G) - rg_tile0_;
            rg_n0_ = board_plan_.tile_first[rg_tile0_];
            rg_ns_ = board_plan_.tile_first[rg_tile0_ + rg_ntiles_] - rg_n0_;
            stream_ = chain_stream_[g];
            hipError_t e = hipStreamWaitEvent(stream_, chain_fork_, 0);
            static const bool serial = std::getenv("SAYURI_CHAINS_SERIAL") != nullptr;  // debugging aid: the chains one after another
            if (serial && g > 0 && e == hipSuccess) e = hipStreamWaitEvent(stream_, chain_join_[g - 1], 0);
            if (e == hipSuccess) rc = forward_graph();
            if (e == hipSuccess && rc == 0) e = hipEventRecord(chain_join_[g], stream_);
            if (e != hipSuccess) rc = fail(std::string("chained forward: ") + hipGetErrorString(e));
        }
        stream_ = main;
        rg_tile0_ = 0; rg_ntiles_ = -1; rg_n0_ = 0; rg_ns_ = -1;
        if (rc) return rc;
        for (int g = 0; g < G; ++g) HIP_OK(hipStreamWaitEvent(main, chain_join_[g], 0));
        return 0;
    }

    int forward_graph() {
        const auto& d = desc_;
        const int C = d.residual_channels, csC = round_up(C, 32), act = d.default_act;
        const BatchGeom g = dgeom();
        for (int i = 0; i < kNumBufs; ++i) busy_[i] = false;
        dbg_call_ = 0;
        dbg_se_call_ = 0;
        sx_idx_ = 0;
        run_.clear();
        table_used_ = 0;

        int x = take();
        {
            const ConvLayerDev& L = cv(SAYURI_L_INPUT_CONV);
            const int in = take();
            const int n0 = rg_n0_, ns = range_ns();
            const int grid = ns * kPackSplit;  // kPackSplit workgroups per sample
            const double px = range_px();
            T* dst = bufs_[in];
            const int cin = d.input_channels, cs = L.cin_s, board = board_;
            const IoSlot& io = io_[cur_slot_];
            if (io.packed_binary > 0) {
                // (records read across PCIe: one workgroup per sample, so that a record crosses once)
                const unsigned* rec = io.packed_src ? io.packed_src : io.packed;
                const int split = io.packed_src ? 1 : kPackSplit;
                const int nbin = io.packed_binary, words = nbin * 12 + 8;
                if (timed("pack_input", 0, (double)ns * words * 4 + px * cs * sizeof(T), [&] {
                        hipLaunchKernelGGL(pack_bits_kernel<T>, dim3(ns * split), dim3(256), 0, stream_, rec, words, nbin, dst, g, cin, cs,
                                           (const int*)d_perm_, n0, split);
                    }))
                    return -1;
            } else {
                const int chunk = pack_input_chunk(slot_pix_, cs, (int)sizeof(T));
                const size_t lds = (size_t)chunk * (cs * sizeof(T) + 16);
                const size_t flat_lds = pack_input_flat_lds(slot_pix_, cs, (int)sizeof(T));
                // a sample that fits 64 KiB of LDS whole and whose planes are below 2^16 floats: one workgroup per sample
                const bool flat = flat_lds <= 64 * 1024 && (size_t)cin * board * board < 65536;
                if (timed("pack_input", 0, px * cin * 4 + px * cs * sizeof(T), [&] {
                        if (flat)
                            hipLaunchKernelGGL(pack_input_flat_kernel<T>, dim3(ns), dim3(kPackFlatThreads), flat_lds, stream_,
                                               (const float*)d_planes_, dst, g, cin, cs, board, (const int*)d_perm_, n0);
                        else
                            hipLaunchKernelGGL(pack_input_kernel<T>, dim3(grid), dim3(kPackThreads), lds, stream_,
                                               (const float*)d_planes_, dst, g, cin, cs, board, (const int*)d_perm_, chunk, n0);
                    }))
                    return -1;
            }
            if (conv("conv3x3_input", L, bufs_[in], bufs_[x], nullptr, act)) return -1;
            if (flags_.dbg_recycle_input) give(in);  // SAYURI_DEBUG_RECYCLE_INPUT: rounds 3-4's hand-back, to show what catches it
            // `in` is NOT handed back: it stays the packed input's buffer for the whole forward.  Its rows have the input
            // convolution's channel stride (64), every later buffer the tower's (256 / 384): recycled as a block's output, sample
            // m's rows would lie on top of sample n's packed input.  With a kernel boundary between every two layers and one
            // stream that is harmless.  Inside the persistent launch it is not -- a workgroup's layers follow one another with no
            // grid-wide order, and a workgroup that STARTS LATE (more tiles than CUs, or the chip shared with another ticket's
            // kernels) found its packed input overwritten by an early workgroup's third layer -- and across the chains of one
            // forward neither.  Rounds 3-4 recycled it: every full-chip launch won that race by a wide margin (all 256 workgroups
            // start together, the input is read within the first ~80 us and overwritten after ~170), which is why it took round
            // 5's bit-level harness to see it (tools/gpu/concurrent_ctx_dbg.py, chains_layers_dbg.py).
        }

        for (int b = 0; b < d.residual_blocks; ++b) {
            const auto& bd = blocks_[b];
            const bool se = bd.apply_se != 0;
            const int last_act = se ? (int)kIdentity : act;
            const int y = take();
            int skip = x;  // buffer added back at the end of the block
            bool se_done = false;  // the SE unit already ran inside the block's last convolution
            if (bd.type == SAYURI_BLOCK_RESIDUAL) {
                const int t0 = take();
                if (conv("conv3x3_tower", cv(SAYURI_L_BLOCK(b, SAYURI_S_CONV1)), bufs_[x], bufs_[t0], nullptr, act)) return -1;
                int fused = 1;
                if (se) {
                    fused = conv_se(cv(SAYURI_L_BLOCK(b, SAYURI_S_CONV2)), fc(SAYURI_L_BLOCK(b, SAYURI_S_SQUEEZE)),
                                    fc(SAYURI_L_BLOCK(b, SAYURI_S_EXCITE)), bufs_[t0], bufs_[y], bufs_[x], C, act);
                    if (fused == 1)
                        fused = conv_sx(cv(SAYURI_L_BLOCK(b, SAYURI_S_CONV2)), fc(SAYURI_L_BLOCK(b, SAYURI_S_SQUEEZE)),
                                        fc(SAYURI_L_BLOCK(b, SAYURI_S_EXCITE)), bufs_[t0], bufs_[y], bufs_[x], C, act);
                    if (fused < 0) return -1;
                    se_done = fused == 0;
                }
                if (fused == 1 &&
                    conv("conv3x3_tower", cv(SAYURI_L_BLOCK(b, SAYURI_S_CONV2)), bufs_[t0], bufs_[y], se ? nullptr : bufs_[x], last_act))
                    return -1;
                give(t0);
            } else if (bd.type == SAYURI_BLOCK_BOTTLENECK) {
                const int t0 = take(), t1 = take();
                if (conv("conv1x1_tower", cv(SAYURI_L_BLOCK(b, SAYURI_S_PRE_BTL)), bufs_[x], bufs_[t0], nullptr, act)) return -1;
                if (conv("conv3x3_tower", cv(SAYURI_L_BLOCK(b, SAYURI_S_CONV1)), bufs_[t0], bufs_[t1], nullptr, act)) return -1;
                if (conv("conv3x3_tower", cv(SAYURI_L_BLOCK(b, SAYURI_S_CONV2)), bufs_[t1], bufs_[t0], nullptr, act)) return -1;
                if (conv("conv1x1_tower", cv(SAYURI_L_BLOCK(b, SAYURI_S_POST_BTL)), bufs_[t0], bufs_[y], se ? nullptr : bufs_[x], last_act)) return -1;
                give(t0); give(t1);
            } else if (bd.type == SAYURI_BLOCK_NESTED_BOTTLENECK) {
                const int r1 = take(), t0 = take(), t1 = take();
                if (conv("conv1x1_tower", cv(SAYURI_L_BLOCK(b, SAYURI_S_PRE_BTL)), bufs_[x], bufs_[r1], nullptr, act)) return -1;
                if (conv("conv3x3_tower", cv(SAYURI_L_BLOCK(b, SAYURI_S_CONV1)), bufs_[r1], bufs_[t0], nullptr, act)) return -1;
                if (conv("conv3x3_tower", cv(SAYURI_L_BLOCK(b, SAYURI_S_CONV2)), bufs_[t0], bufs_[t1], bufs_[r1], act)) return -1;
                if (conv("conv3x3_tower", cv(SAYURI_L_BLOCK(b, SAYURI_S_CONV3)), bufs_[t1], bufs_[t0], nullptr, act)) return -1;
                if (conv("conv3x3_tower", cv(SAYURI_L_BLOCK(b, SAYURI_S_CONV4)), bufs_[t0], bufs_[r1], bufs_[t1], act)) return -1;
                if (conv("conv1x1_tower", cv(SAYURI_L_BLOCK(b, SAYURI_S_POST_BTL)), bufs_[r1], bufs_[y], se ? nullptr : bufs_[x], last_act)) return -1;
                give(r1); give(t0); give(t1);
            } else {  // mixer: x' = act(dw(x)+b) + x is the new skip
                const int s2 = take(), t1 = take();
                if (depthwise("depthwise", cv(SAYURI_L_BLOCK(b, SAYURI_S_DW_CONV)), bufs_[x], bufs_[s2], bufs_[x], act)) return -1;
                if (conv("conv1x1_tower", cv(SAYURI_L_BLOCK(b, SAYURI_S_CONV1)), bufs_[s2], bufs_[t1], nullptr, act)) return -1;
                if (conv("conv1x1_tower", cv(SAYURI_L_BLOCK(b, SAYURI_S_CONV2)), bufs_[t1], bufs_[y], se ? nullptr : bufs_[s2], last_act)) return -1;
                give(t1);
                give(x);
                x = s2;
                skip = s2;
            }
            if (se && !se_done) {
                if (se_unit(fc(SAYURI_L_BLOCK(b, SAYURI_S_SQUEEZE)), fc(SAYURI_L_BLOCK(b, SAYURI_S_EXCITE)), bufs_[y],
                            bufs_[skip], C, csC, act, rg_n0_, range_ns()))
                    return -1;
            }
            give(x);
            x = y;
        }

        // heads
        const int Cp = d.policy_head_channels, Cv = d.value_head_channels;
        HeadParams h;
        h.p_inter = fc(SAYURI_L_P_INTER_FC).dev();
        h.pass_fc = fc(SAYURI_L_PASS_FC).dev();
        h.v_inter = fc(SAYURI_L_V_INTER_FC).dev();
        h.v_misc = fc(SAYURI_L_V_MISC).dev();
        h.prob_w = cv(SAYURI_L_PROB_CONV).w32;
        h.prob_b = cv(SAYURI_L_PROB_CONV).bias;
        h.own_w = cv(SAYURI_L_V_OWNERSHIP).w32;
        h.own_b = cv(SAYURI_L_V_OWNERSHIP).bias;
        h.Cp = Cp; h.cs_p = round_up(Cp, 32); h.Cv = Cv; h.cs_v = round_up(Cv, 32);
        h.prob_ch = d.probabilities_channels; h.act = act; h.board = board_;
        h.prob = d_prob_; h.pass = zc_pass_ ? zc_pass_ : d_pass_; h.misc = zc_misc_ ? zc_misc_ : d_misc_; h.own = d_own_; h.perm = d_perm_;
        if (head_img_ && heads_fused_enabled()) {
            // both heads of a sample in one workgroup: trunk -> LDS -> stacked 1x1 convolution on the matrix cores -> pooling,
            // FCs and the per-pixel planes (head_board.h)
            HeadBoardParams hp;
            hp.dbg = nullptr;
#ifdef SAYURI_EXPERIMENTS
            if (flags_.heads_dbg) {
                if (!d_hdbg_ && dev_alloc(&d_hdbg_, 4 * 8)) return -1;
                hp.dbg = d_hdbg_;
            }
#endif
            hp.trunk = bufs_[x]; hp.w = head_img_; hp.w2 = head_img2_; hp.bias = head_bias_; hp.g = g; hp.cs = csC; hp.PT = head_pt_; hp.VT = head_vt_; hp.h = h;
            hp.n0 = rg_n0_;
            const auto fn = head_fn_;
            {
                const int ns = range_ns();
                const double flops = 2.0 * range_px() * C * (Cp + Cv);
                return timed("heads_fused", flops, range_px() * csC * 2, [&] {
                    hipLaunchKernelGGL(fn, dim3(ns), dim3(512), kMaxLds, stream_, hp);
                });
            }
        }
        if (rg_ns_ >= 0) return fail("chained forward reached the separate head kernels");
        int pb = take();
        const int vb = take();
        if (conv("conv1x1_head", cv(SAYURI_L_P_HD_CONV), bufs_[x], bufs_[pb], nullptr, act)) return -1;
        if (d.policy_head_type == 1) {
            const int p2 = take();
            if (depthwise("depthwise", cv(SAYURI_L_P_DW_CONV), bufs_[pb], bufs_[p2], nullptr, act)) return -1;
            if (conv("conv1x1_head", cv(SAYURI_L_P_PT_CONV), bufs_[p2], bufs_[pb], nullptr, act)) return -1;
            give(p2);
        }
        if (conv("conv1x1_head", cv(SAYURI_L_V_HD_CONV), bufs_[x], bufs_[vb], nullptr, act)) return -1;
        const int maxc = std::max(Cp, Cv);
        const size_t smem = sizeof(float) * (7 * maxc + 512);
        const T* pc = bufs_[pb];
        const T* vc = bufs_[vb];
        return timed("head_tail", 0, 0, [&] {
            hipLaunchKernelGGL(head_tail_kernel<T>, dim3(2 * geom_.n), dim3(256), smem, stream_, pc, vc, g, h);
        });
    }

    // -------------------------------------------------------------- the persistent tower launch (conv_tower.h)
    // Consecutive board convolutions whose channel tile covers the layer are not launched one by one: conv() / conv_se()
    // append them to run_, and the first launch of anything else (timed()) -- in practice the heads -- sends the whole run
    // as ONE launch that walks a table of TowerLayer in device memory.  The table of a slot is re-uploaded only when its
    // contents change (another batch geometry; the buffers and weights of a slot never move).
    static constexpr int kTowerCap = 512;  // table elements per slot
    struct TowerSlot {
        TowerLayer* dev = nullptr;
        TowerLayer* stage[2] = {nullptr, nullptr};  // pinned staging, alternating
        hipEvent_t staged[2] = {nullptr, nullptr};   // the last copy out of stage[i]
        int next_stage = 0;
        std::vector<TowerLayer> cache;               // what dev holds
    };
    int tower_load() { return load_tower_module(&tower_mod_, tower_fn_); }
    bool tower_ok(int kot) const { return tower_mod_ && !profiling_ && (kot == 256 || kot == 128); }
    int tower_append(int kot, const BoardSeParams& sp, bool has_se, double flops, double bytes) {
        if (!run_.empty() && (run_kot_ != kot || (int)run_.size() + table_used_ >= kTowerCap) && tower_flush()) return -1;
        if (run_.empty()) { run_kot_ = kot; run_flops_ = run_bytes_ = 0; }
        TowerLayer t;
        std::memset(&t, 0, sizeof(t));
        t.sp = sp;
        t.has_se = has_se ? 1 : 0;
        if (flags_.io_v2) {
            // what the bodies never read (tile = workgroup id, the launch's grid is the batch): left out of the table, so that
            // a batch of 250 positions finds the table of a batch of 256 in place and nothing is uploaded
            ConvParams& c = t.sp.b.c;
            c.num_pix_tiles = 0;
            c.g.n_samples = 0;
            c.g.total_pix = 0;
        }
        run_.push_back(t);
        run_flops_ += flops;
        run_bytes_ += bytes;
        return 0;
    }
    int tower_flush() {
        std::vector<TowerLayer> run;
        run.swap(run_);  // timed() below must not see a pending run
        TowerSlot& ts = tower_[cur_slot_];
        if (!ts.dev) {
            if (dev_alloc(&ts.dev, kTowerCap)) return -1;
            for (int i = 0; i < 2; ++i) {
                HIP_OK(hipHostMalloc((void**)&ts.stage[i], sizeof(TowerLayer) * kTowerCap, hipHostMallocDefault));
                HIP_OK(hipEventCreateWithFlags(&ts.staged[i], hipEventDisableTiming));
            }
            ts.cache.assign(kTowerCap, TowerLayer{});
        }
        const int n = (int)run.size(), first = table_used_;
        if (first + n > kTowerCap) return fail("tower table overflow");
        if (flags_.tower_noepi_after >= 0 && tower_launches_++ >= flags_.tower_noepi_after)
            for (auto& t : run)
                if (t.sp.b.row_order == 1 && !t.has_se) t.sp.b.row_order = 3;  // MEASURING: no epilogue (tower_seam.py epi_hook)
        for (int i = 0; i < n; ++i) {
            run[i].self = ts.dev + first + i;
            run[i].last = i + 1 == n ? 1 : 0;
        }
        // weight hand-over (conv_board.h, CHAIN main loop): a plain layer without residual leaves the LDS alone after its K
        // loop, so its last K group can bring in the next layer's first weight group.  Needs the same weight geometry on
        // both sides (the piece addresses are computed with this layer's strides) and an even number of 32-channel chunks
        // (the last group then sits in ring slot 1 and slot 0 is free).
        for (int i = 0; i + 1 < n && flags_.tower_chain; ++i) {
            const ConvParams &a = run[i].sp.b.c, &b = run[i + 1].sp.b.c;
            if (run[i].has_se || a.res || a.cin_s != b.cin_s || a.ko_pad != b.ko_pad || (a.cin_s / kChunk) % 2) continue;
            run[i].sp.b.w_next = b.w;
            run[i + 1].sp.b.w_ready = 1;
        }
        if (std::memcmp(run.data(), ts.cache.data() + first, sizeof(TowerLayer) * n) != 0) {
            const int st = ts.next_stage;
            ts.next_stage ^= 1;
            HIP_OK(hipEventSynchronize(ts.staged[st]));  // the copy that last read this staging area (never recorded: returns at once)
            std::memcpy(ts.stage[st], run.data(), sizeof(TowerLayer) * n);
            HIP_OK(hipMemcpyAsync(ts.dev + first, ts.stage[st], sizeof(TowerLayer) * n, hipMemcpyHostToDevice, stream_));
            ++table_uploads_;
            HIP_OK(hipEventRecord(ts.staged[st], stream_));
            std::memcpy(ts.cache.data() + first, run.data(), sizeof(TowerLayer) * n);
        }
        table_used_ += n;
        const hipFunction_t fn = tower_fn_[run_kot_ == 256 ? 0 : 1];
        const TowerLayer* arg = ts.dev + first;
        const int grid = board_plan_.ntiles;
        hipError_t lrc = hipSuccess;
        static const bool sync_dbg = std::getenv("SAYURI_TOWER_SYNC") != nullptr;  // debugging aid: nothing overlaps the tower launch
        if (sync_dbg) HIP_OK(hipStreamSynchronize(stream_));
        const int rc = timed("tower_run", run_flops_, run_bytes_, [&] {
            void* params[] = {(void*)&arg};
            lrc = hipModuleLaunchKernel(fn, grid, 1, 1, 512, 1, 1, 0, stream_, params, nullptr);
        });
        if (sync_dbg && lrc == hipSuccess) HIP_OK(hipStreamSynchronize(stream_));
        if (lrc != hipSuccess) return fail(std::string("hipModuleLaunchKernel(conv_tower_kernel): ") + hipGetErrorString(lrc));
        return rc;
    }
    // conv_board_sx.h (SE units of layers split over channel tiles)
    int sx_kts_ = 0;                      // channel tiles per layer (0: the network has no such unit)
    unsigned* sx_err_host_ = nullptr;     // host-visible word a workgroup sets when its wait for the siblings ran out
    unsigned* sx_err_dev_ = nullptr;
    unsigned sx_epoch0_ = 0;              // tags of the current forward: sx_epoch0_ + 1 + index of the SE layer
    int sx_idx_ = 0;
    bool sx_disabled_ = false;            // set by sx_check(): the exchange timed out once on this ctx
    int sx_check() {
        if (sx_err_host_ && *(volatile unsigned*)sx_err_host_) {
            const unsigned e = *(volatile unsigned*)sx_err_host_;
            *(volatile unsigned*)sx_err_host_ = 0;
            sx_disabled_ = true;  // this ctx goes on with the separate kernels: a wait that ran out once is not tried again
            return fail("SE exchange between the channel tiles of a board tile timed out (epoch " + std::to_string(e) +
                        "): the results of this forward are invalid; from here on this context runs the unit as separate kernels (SAYURI_SE_SPLIT=0)");
        }
        return 0;
    }
    EngineFlags flags_;
    hipModule_t tower_mod_ = nullptr;
    hipFunction_t tower_fn_[2] = {nullptr, nullptr};
    TowerSlot tower_[2];
    std::vector<TowerLayer> run_;
    int run_kot_ = 0, table_used_ = 0;
    long tower_launches_ = 0;
    int table_uploads_ = 0;
    // SAYURI_HIP_FWDSTAT: device time of the forwards sent through submit(), by batch-size class
    bool fwdstat_ = std::getenv("SAYURI_HIP_FWDSTAT") != nullptr;
    hipEvent_t fs_ev_[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
    double fs_d2h_ms_ = 0, fs_d2h_max_ = 0;
    long fs_d2h_slow_ = 0;
    float *zc_pass_ = nullptr, *zc_misc_ = nullptr;  // this submit's pass / misc go straight to these (pinned host) buffers
    bool fs_pending_[2] = {false, false};
    int fs_n_[2] = {0, 0};
    double fs_ms_[3] = {0, 0, 0};
    long fs_cnt_[3] = {0, 0, 0}, fs_uploads_ = 0;
    double run_flops_ = 0, run_bytes_ = 0;

    int device_;
    sayuri_hip_netdesc desc_;
    std::vector<sayuri_hip_blockdesc> blocks_;
    int max_batch_, board_;
    int cs_max_ = 32, slot_pix_ = 0;
    std::map<int, ConvLayerDev> convs_;
    std::map<int, FcLayerDev> fcs_;
    bool finalized_ = false, have_batch_ = false, profiling_ = false;
    hipStream_t stream_ = nullptr, h2d_stream_ = nullptr, d2h_stream_ = nullptr;
    hipEvent_t ev0_ = nullptr, ev1_ = nullptr;
    hipEvent_t h2d_done_[2] = {nullptr, nullptr}, fwd_done_[2] = {nullptr, nullptr};
    // device-side batch i/o, one set per ticket; the d_* members below alias the slot the current forward uses
    struct IoSlot {
        float *planes = nullptr, *prob = nullptr, *pass = nullptr, *misc = nullptr, *own = nullptr;
        unsigned* packed = nullptr;  // packed records of the batch (allocated on first use)
        const unsigned* packed_src = nullptr;  // non-null: the batch's records are read where the caller has them (pinned host memory)
        int packed_binary = 0;       // > 0: the slot's current batch came as packed records with this many bit planes
        int *off = nullptr, *bsz = nullptr, *perm = nullptr;
        T* bufs[kNumBufs] = {};
        float *gate = nullptr, *separt = nullptr;
        unsigned long long* sx_xchg = nullptr;
        unsigned sx_epoch = 0;  // the last tag used in sx_xchg
        std::map<int, TileTabs> tabs;   // index tables of the geometry this slot last ran (keyed by tile variant)
        BoardTabs board;
        std::vector<int> tabs_bsz;
        bool tabs_single = false;  // tabs_bsz is one board size with one sample per tile
        int tabs_n = -1;           // batch size the across-sample tables (tabs) were last built for
    };
    IoSlot io_[2];
    hipStream_t compute_[2] = {nullptr, nullptr};
    bool inorder_ = false;  // a ticket's copies on the ticket's compute stream (init(), submit())
    int cur_slot_ = 0;
    void select_slot(int t) {
        IoSlot& io = io_[t];
        cur_slot_ = t;
        d_planes_ = io.planes; d_off_ = io.off; d_bsz_ = io.bsz; d_perm_ = io.perm;
        d_prob_ = io.prob; d_pass_ = io.pass; d_misc_ = io.misc; d_own_ = io.own;
        for (int i = 0; i < kNumBufs; ++i) bufs_[i] = io.bufs[i];
        d_gate_ = io.gate; d_separt_ = io.separt;
        if (compute_[t]) stream_ = compute_[t];
    }
    std::vector<void*> allocs_;
    size_t dev_bytes_ = 0;
    T* bufs_[kNumBufs] = {};
    bool busy_[kNumBufs] = {};
    float *d_planes_ = nullptr, *d_gate_ = nullptr, *d_separt_ = nullptr, *d_prob_ = nullptr, *d_pass_ = nullptr, *d_misc_ = nullptr,
          *d_own_ = nullptr;
    int *d_off_ = nullptr, *d_bsz_ = nullptr, *d_perm_ = nullptr;
    std::vector<int> perm_;  // device sample -> caller's slot (enqueue_inputs)
    float* d_zeros_ = nullptr;
    int* h_geom_ = nullptr;  // pinned 2-slot ring: [slot][off(max_batch+1) | bsz(max_batch) | perm(max_batch)]
    int geom_slot_ = 0, next_ticket_ = 0;
    hipEvent_t tick_ev_[2] = {nullptr, nullptr};
    HostGeom geom_;
    std::vector<int> prev_bsz_;
    std::map<int, GldsChoice> glds_cache_;
    BoardPlan board_plan_;
    HeadFn head_fn_ = nullptr;
    void* head_img2_ = nullptr;  // per-pixel weights (policy planes, ownership) as an MFMA image
    void* head_img_ = nullptr;   // stacked head-convolution image (head_board.h); null = separate head kernels
    float* head_bias_ = nullptr;
    int head_pt_ = 0, head_vt_ = 0;
    unsigned long long* d_hdbg_ = nullptr;  // SAYURI_HEADS_DBG timeline of head_board_kernel
    unsigned long long* d_dbg_ = nullptr;  // SAYURI_BOARD_DBG timeline of one tower convolution
    unsigned long long* d_sxdbg_ = nullptr;  // SAYURI_SX_DBG timeline of one split SE convolution
    int dbg_call_ = 0, dbg_se_call_ = 0;
    bool dbg_is_se_ = false;
    bool board_plan_valid_ = false;
    std::map<int, TileChoice> tile_cache_;
    std::map<std::string, Stat> stats_;
    // light per-launch timing of one kernel class inside time_runs()
    bool light_ = false;
    std::string light_name_;
    int light_group_ = 1;            // launches of the marked class bracketed by one event pair
    std::vector<int> group_counts_;  // launches inside each pair of the last time_runs
    bool group_open_ = false;
    std::vector<hipEvent_t> pool_;
    size_t pool_used_ = 0;
    double light_flops_ = 0, light_bytes_ = 0;
    Stat timed_stat_;

};

}  // namespace sayuri
