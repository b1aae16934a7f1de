// engine_plan.h -- what the engine decides on the host before anything is launched: the kernel registries, the switches read
// from the environment, the batch geometry and the tile plans, the layers' device images, the tower code object's loader.
// (One translation unit: engine.hip includes engine_plan.h, engine_graph.h and engine_taps.h in this order.)
#pragma once
namespace sayuri {

static thread_local std::string g_err;
static std::atomic<unsigned> g_host_free_gen{0};  // sayuri_hip_host_free calls so far (Engine::zc_device_pointer)
static thread_local int g_test_conv_kind = 0;  // kernel family the last sayuri_hip_test_conv call ran: 0 generic, 1 glds, 2 board, 3 depthwise
static int fail(const std::string& m) { g_err = m; return -1; }

#define HIP_OK(expr)                                                                      \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) {                                                           \
            g_err = std::string(#expr) + ": " + hipGetErrorString(e_);                    \
            return -1;                                                                    \
        }                                                                                 \
    } while (0)

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
constexpr size_t kMaxLds = 160 * 1024;
constexpr int kNumCU = 256;

// ------------------------------------------------------------------ conv kernel registry
template <typename T> struct ConvKernelTable {
    typedef void (*Fn)(const ConvParams);
    struct Entry { int wmt, wnt; Fn fn; size_t (*lds)(int); int npos_cap; };
    static std::vector<Entry>& entries() {
        static std::vector<Entry> e;
        return e;
    }
};

template <typename T, int WMT, int WNT> static void register_conv() {
    typedef ConvCfg<T, WMT, WNT> Cfg;
    auto fn = &conv_mfma_kernel<T, WMT, WNT>;
    ConvKernelTable<T>::entries().push_back({WMT, WNT, fn, &Cfg::lds_bytes, Cfg::NPOS_CAP});
}

template <typename T> static void register_all_convs();
template <> void register_all_convs<f16>() {
    if (!ConvKernelTable<f16>::entries().empty()) return;
    register_conv<f16, 1, 4>(); register_conv<f16, 2, 4>(); register_conv<f16, 3, 4>();
    register_conv<f16, 4, 4>(); register_conv<f16, 6, 4>(); register_conv<f16, 8, 4>();
    register_conv<f16, 1, 2>(); register_conv<f16, 2, 2>(); register_conv<f16, 3, 2>();
    register_conv<f16, 4, 2>(); register_conv<f16, 6, 2>(); register_conv<f16, 8, 2>();
    register_conv<f16, 6, 3>(); register_conv<f16, 8, 3>();
}
template <> void register_all_convs<float>() {
    if (!ConvKernelTable<float>::entries().empty()) return;
    register_conv<float, 1, 4>(); register_conv<float, 2, 4>(); register_conv<float, 3, 4>();
    register_conv<float, 4, 4>();
    register_conv<float, 1, 2>(); register_conv<float, 2, 2>(); register_conv<float, 3, 2>();
    register_conv<float, 4, 2>();
}

// tuned fp16 3x3 kernels for any batch geometry (conv_glds.h): 192 / 128 / 64-pixel tiles across samples
struct GldsEntry {
    int wmt, wnt;
    void (*fn)(const GldsParams);
    void (*setup)(BatchGeom, int*, int2*);
    size_t lds;
    int npos_cap, npos, pt;
};
static std::vector<GldsEntry>& glds_entries() {
    static std::vector<GldsEntry> e;
    return e;
}
template <int WMT, int WNT> static void register_glds() {
    typedef GldsCfg<WMT, WNT> Cfg;
    glds_entries().push_back({WMT, WNT, &conv_glds_kernel<WMT, WNT>, &tile_setup_kernel<Cfg::PT, Cfg::NPOS>, Cfg::lds_bytes(),
                              Cfg::NPOS_CAP, Cfg::NPOS, Cfg::PT});
}
static void register_all_glds() {
    if (!glds_entries().empty()) return;
    register_glds<8, 3>(); register_glds<8, 2>(); register_glds<8, 1>();
    register_glds<4, 3>(); register_glds<4, 2>(); register_glds<4, 1>();
}
// The weight image of the board kernels with an even tile count: the same planes with their rows in board_row_channel order
// (conv_board.h: a lane then holds 8 consecutive channels of a row-tile pair without any exchange).
template <typename T> static std::vector<T> board_row_order(const std::vector<T>& img, int ko_pad) {
    std::vector<T> out(img.size());
    const size_t planes = img.size() / ((size_t)ko_pad * 8);
    for (size_t pl = 0; pl < planes; ++pl)
        for (int r = 0; r < ko_pad; ++r)
            std::copy_n(img.begin() + (pl * ko_pad + board_row_channel(r)) * 8, 8, out.begin() + (pl * ko_pad + r) * 8);
    return out;
}
static bool board_uses_row_order(int kot) { return (kot / 64) % 2 == 0; }  // 256 / 128: yes; 192 (three row tiles per wave): natural order
// one-workgroup-per-board kernels (conv_board.h), by output-channel tile
typedef void (*BoardFn)(const BoardParams);
typedef void (*BoardSeFn)(const BoardSeParams);
struct BoardEntry { int kot; BoardFn fn; BoardSeFn fn_se; size_t (*lds)(int); };
static const BoardEntry kBoardEntries[] = {
    {256, &conv_board_kernel<4>, &conv_board_se_kernel<4>, &BoardCfg<4>::lds_bytes},  // SAYURI_BOARD_DBG=n swaps in <4, true> (timeline)
    {192, &conv_board_kernel<3>, nullptr, &BoardCfg<3>::lds_bytes},
    {128, &conv_board_kernel<2>, &conv_board_se_kernel<2>, &BoardCfg<2>::lds_bytes},
};
// head_board_kernel variants: {row tiles, trunk chunks in flight}; the first that fits the LDS is used (head_board_fits)
typedef void (*HeadFn)(const HeadBoardParams);
struct HeadEntry { int rt, depth; HeadFn fn; };
static const HeadEntry kHeadEntries[] = {
    {2, 5, &head_board_kernel<2, 5>}, {4, 5, &head_board_kernel<4, 5>}, {4, 3, &head_board_kernel<4, 3>},
    {6, 3, &head_board_kernel<6, 3>}, {6, 2, &head_board_kernel<6, 2>},
};
static void enable_big_lds_glds() {
    register_all_glds();
    for (const auto& e : glds_entries())
        (void)hipFuncSetAttribute((const void*)e.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds);
    (void)hipFuncSetAttribute((const void*)&conv_board_kernel<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds);
    for (const auto& e : kHeadEntries) (void)hipFuncSetAttribute((const void*)e.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds);
    (void)hipFuncSetAttribute((const void*)&conv_board_sx_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds);
    for (const auto& e : kBoardEntries) {
        (void)hipFuncSetAttribute((const void*)e.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds);
        if (e.fn_se) (void)hipFuncSetAttribute((const void*)e.fn_se, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds);
    }
}
// Switches of one engine, read from the environment ONCE, in sayuri_hip_create (and per call in the layer-level test
// taps): nothing on the launch path calls getenv.  They select between product paths that give the same results (A/B
// measurements, tests that check one path against the other).  The measuring-only switches (in-kernel timelines, forced
// activation / channel tile) exist only in builds with -DSAYURI_EXPERIMENTS.
//   SAYURI_CONV=v0 | glds[:wnt]   3x3 layers on the generic / the LDS-DMA-tiles-across-samples kernel instead of one workgroup per board
//   SAYURI_TOWER=0                one launch per convolution instead of one persistent launch per run of board convolutions
//   SAYURI_SE_FUSED=0             SE unit as se_pool / se_fc / se_scale instead of inside the convolution
//   SAYURI_HEADS_FUSED=0          conv1x1 x2 + head_tail instead of head_board_kernel
//   SAYURI_NO_ARITH=1             board kernels read their index tables instead of computing the entries
//   SAYURI_COMPUTE_STREAMS=2      the two tickets' forwards on two streams
struct ConvOverride {
    bool v0 = false, no_board = false;
    int wnt = 0;
    // The board kernel runs whenever the batch's boards fit its tiles, however empty the tiles are: which convolution kernel
    // a sample meets must not depend on its batch mates (a lone 9x9 board fills a fifth of its tile; with the across-sample
    // kernel it came out ~1e-4 away from the same position inside a larger batch).  SAYURI_BOARD_MIN_FILL=0.55 brings back the
    // rule of rounds 2-4 (tiles less than 55 % full go to the across-sample kernel: half the latency of a lone small board).
    double board_min_fill = 0.0;
};
struct EngineFlags {
    ConvOverride conv;
    bool tower = true, se_fused = true, heads_fused = true, arith = true;
    bool se_by_geometry = true;        // which samples take the fused SE form depends on their board size alone (conv_se); SAYURI_SE_BY_GEOMETRY=0: on the tiles' occupancy
    bool io_v2 = true;                 // SAYURI_IO_V2=0: geometry / small outputs by copies again (A/B; see submit())
    bool io_zc = true, io_geom = true, io_prefix = true;  // its three parts, one at a time (SAYURI_IO_ZC / _GEOM / _PREFIX = 0)
    bool io_zc_in = true;  // packed records read where the caller has them (SAYURI_IO_ZC_IN=0: copied first, rounds 2-4)
    bool tower_chain = true;           // a layer of the persistent run fetches the next layer's first weight group (SAYURI_TOWER_CHAIN=0: off)
    bool tower_gen_epi = true;         // Mish layers of the run take the generated epilogue (SAYURI_TOWER_GEN_EPI=0: the compiled one)
    bool se_split = true;              // SAYURI_SE_SPLIT=0: SE units of layers split over several channel tiles (384 channels) as
                                       // se_pool / se_fc / se_scale again instead of inside the convolution (conv_board_sx.h)
    int tower_noepi_after = -1;        // SAYURI_TOWER_NOEPI_AFTER=n (measuring): from the n-th persistent launch on, the layers with the
                                       // generated epilogue skip it (row_order = 3): timing only, the outputs are stale
    unsigned dbg_sx_epoch0 = 0;        // SAYURI_DEBUG_SX_EPOCH0=n (tests): the exchange tags of a new context start at n (the wrap of the tags)
    bool dbg_sx_stall = false;         // SAYURI_DEBUG_SX_STALL=1 (tests): one sibling of every tile never publishes under the right tag
    int sx_dbg = 0;                    // SAYURI_SX_DBG=n: s_memtime timeline of the n-th split SE convolution of a profiled forward
    int dbg_recycle_input = 0;         // SAYURI_DEBUG_RECYCLE_INPUT=1: hand the packed input's buffer back to the pool after the input
                                       // convolution, as rounds 3-4 did (the row-stride table below then REFUSES the forward); =2: and
                                       // switch the table off -- the race of rounds 3-4 is back (tests/test_gpu_fuzz.py shows that it sees it)
    int compute_streams = 1;
    int chains = 0;                    // SAYURI_CHAINS: 0 = the engine decides, 1 = never, N = N chains whenever a batch qualifies (Engine::forward)
    bool io_inorder = true;            // each ticket's upload, forward and download on the ticket's own stream (submit()); SAYURI_IO_INORDER=0: three streams and events
    int board_kot = 0;                 // experiments: only this channel tile
    int act_override = -1;             // experiments: activation of every board convolution
    int board_dbg = 0, heads_dbg = 0;  // experiments: in-kernel timelines
    static bool off(const char* name) { const char* e = getenv(name); return e && atoi(e) == 0; }
    static EngineFlags from_env() {
        EngineFlags f;
        if (const char* e = getenv("SAYURI_CONV")) {
            if (!strncmp(e, "v0", 2)) { f.conv.v0 = true; f.conv.no_board = true; }
            else if (!strncmp(e, "glds", 4)) { f.conv.no_board = true; (void)sscanf(e, "glds:%d", &f.conv.wnt); }
        }
        if (const char* e = getenv("SAYURI_BOARD_MIN_FILL")) f.conv.board_min_fill = atof(e);
        f.tower = !off("SAYURI_TOWER");
        f.tower_chain = !off("SAYURI_TOWER_CHAIN");
        f.tower_gen_epi = !off("SAYURI_TOWER_GEN_EPI");
        if (const char* e = getenv("SAYURI_DEBUG_RECYCLE_INPUT")) f.dbg_recycle_input = atoi(e);
        f.se_split = !off("SAYURI_SE_SPLIT");
        if (const char* e = getenv("SAYURI_SX_DBG")) f.sx_dbg = atoi(e);
        f.dbg_sx_stall = getenv("SAYURI_DEBUG_SX_STALL") != nullptr;
        if (const char* e = getenv("SAYURI_DEBUG_SX_EPOCH0")) f.dbg_sx_epoch0 = (unsigned)strtoul(e, nullptr, 0);
        if (const char* e = getenv("SAYURI_TOWER_NOEPI_AFTER")) f.tower_noepi_after = atoi(e);
        f.io_v2 = !off("SAYURI_IO_V2");
        f.io_zc = f.io_v2 && !off("SAYURI_IO_ZC");
        f.io_zc_in = f.io_zc && !off("SAYURI_IO_ZC_IN");
        f.io_geom = f.io_v2 && !off("SAYURI_IO_GEOM");
        f.io_prefix = f.io_v2 && !off("SAYURI_IO_PREFIX");
        f.se_fused = !off("SAYURI_SE_FUSED");
        f.se_by_geometry = !off("SAYURI_SE_BY_GEOMETRY");
        f.heads_fused = !off("SAYURI_HEADS_FUSED");
        f.arith = !getenv("SAYURI_NO_ARITH");
        if (const char* e = getenv("SAYURI_COMPUTE_STREAMS")) f.compute_streams = atoi(e) == 2 ? 2 : 1;
        if (const char* e = getenv("SAYURI_CHAINS")) f.chains = std::max(0, std::min(atoi(e), 4));
        if (const char* e = getenv("SAYURI_IO_INORDER")) f.io_inorder = atoi(e) != 0;
#ifdef SAYURI_EXPERIMENTS
        if (const char* e = getenv("SAYURI_BOARD_KOT")) f.board_kot = atoi(e);
        if (const char* e = getenv("SAYURI_ACT_OVERRIDE")) f.act_override = atoi(e);
        if (const char* e = getenv("SAYURI_BOARD_DBG")) f.board_dbg = atoi(e);
        if (getenv("SAYURI_HEADS_DBG")) f.heads_dbg = 1;
        if (f.board_dbg || f.heads_dbg || f.act_override >= 0) f.tower = false;
#endif
        return f;
    }
};

// allow > 64 KiB of dynamic LDS on the current device
template <typename T> static void enable_big_lds() {
    register_all_convs<T>();
    for (const auto& e : ConvKernelTable<T>::entries())
        (void)hipFuncSetAttribute((const void*)e.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds);
}

// candidate output-channel tiles, largest first
static int pick_wmt(int cout_s, bool fp16) {
    // KO_T = 32*WMT.  Smallest tile that covers cout_s, else the tile with least padding.
    const int opts16[] = {1, 2, 3, 4, 6, 8};
    const int opts32[] = {1, 2, 3, 4};
    const int* opts = fp16 ? opts16 : opts32;
    const int nopts = fp16 ? 6 : 4;
    for (int i = 0; i < nopts; ++i)
        if (opts[i] * 32 >= cout_s) return opts[i];
    int best = opts[nopts - 1], best_pad = 1 << 30;
    for (int i = nopts - 1; i >= 0; --i) {
        const int kot = opts[i] * 32, pad = round_up(cout_s, kot) - cout_s;
        if (pad < best_pad) { best_pad = pad; best = opts[i]; }
    }
    return best;
}

// ------------------------------------------------------------------ host-side geometry
struct HostGeom {
    std::vector<int> bsz, off;  // off has n+1 entries
    int n = 0, total = 0;
    // bs*bs when every sample has the same board size, else 0
    int uniform_sq() const {
        for (int i = 1; i < n; ++i)
            if (bsz[i] != bsz[0]) return 0;
        return n > 0 ? bsz[0] * bsz[0] : 0;
    }
    // worst-case LDS halo positions / subregions of any PT-pixel tile
    void tile_bounds(int PT, int* npos_out, int* nsub_out) const {
        int max_pos = 0, max_sub = 0;
        int s = 0;
        for (int g0 = 0; g0 < total; g0 += PT) {
            const int g1 = std::min(g0 + PT, total);
            while (s + 1 < n && off[s + 1] <= g0) ++s;
            int pos = 0, sub = 0;
            for (int m = s; m < n && off[m] < g1; ++m) {
                const int bs = bsz[m];
                const int a = std::max(g0, off[m]) - off[m], b = std::min(g1, off[m + 1]) - off[m];
                const int rows = (b - 1) / bs - a / bs + 3;
                pos += rows * (bs + 2);
                ++sub;
            }
            max_pos = std::max(max_pos, pos);
            max_sub = std::max(max_sub, sub);
        }
        *npos_out = round_up(std::max(max_pos, 16), 16);
        *nsub_out = max_sub;
    }
};

// Choose the tuned LDS-DMA kernel variant for an fp16 3x3 layer with `ko_pad` weight rows on
// this batch geometry; nullptr when none applies (the generic conv_mfma kernel is used then).
static const GldsEntry* pick_glds(const HostGeom& geom, int ko_pad, int* ntiles_out, const ConvOverride& ov) {
    if (ko_pad % 128 != 0) return nullptr;
    if (ov.v0) return nullptr;
    const int wmt = ko_pad % 256 == 0 ? 8 : 4;
    const int kot_tiles = ko_pad / (wmt * 32);
    const GldsEntry* best = nullptr;
    double best_cost = 1e30;
    for (const auto& e : glds_entries()) {
        if (e.wmt != wmt) continue;
        if (ov.wnt && e.wnt != ov.wnt) continue;
        const int PT = e.pt;
        int npos, nsub;
        geom.tile_bounds(PT, &npos, &nsub);
        if (npos > e.npos_cap || nsub > kMaxSub || e.lds > kMaxLds) continue;
        const int ntiles = (geom.total + PT - 1) / PT;
        const double waves = std::ceil((double)ntiles * kot_tiles / kNumCU);
        const double cost = waves * (PT + 24);
        if (cost < best_cost) { best_cost = cost; best = &e; *ntiles_out = ntiles; }
    }
    return best;
}

// The one-workgroup-per-board plan of a batch geometry (conv_board.h): consecutive samples packed greedily.
struct BoardPlan {
    int ntiles = 0, npos = 0;
    bool ok = false, single = false;  // single: one sample per tile
    int uniform_info = -1;            // every tile has this (column tiles | board size << 8), or -1
    double fill = 0;
    std::vector<int> tile_first;      // first sample of every tile, then the number of samples (ntiles + 1 entries)
};
static BoardPlan board_plan(const HostGeom& geom, const ConvOverride& ov) {
    BoardPlan bp;
    if (ov.no_board || geom.n <= 0) return bp;
    BoardPack pk;
    int max_pos = 0, info0 = -2;
    int tile_start = 0;
    auto close_tile = [&] {
        bp.tile_first.push_back(tile_start);
        max_pos = std::max(max_pos, pk.pos);
        const int info = ((pk.px + 15) / 16) | (pk.bs0 << 8);
        info0 = info0 == -2 ? info : (info0 == info ? info0 : -1);
        ++bp.ntiles;
    };
    for (int s = 0; s < geom.n; ++s) {
        const int bs = geom.bsz[s];
        if (!BoardPack{}.fits(bs)) return bp;  // a board that does not fit a tile on its own
        if (pk.cnt > 0 && !pk.fits(bs)) {
            close_tile();
            pk = BoardPack{};
            tile_start = s;
        }
        pk.add(bs);
    }
    close_tile();
    bp.tile_first.push_back(geom.n);
    bp.uniform_info = info0;
    bp.npos = round_up(max_pos, 64);
    bp.fill = (double)geom.total / ((double)bp.ntiles * kBoardPT);
    bp.single = bp.ntiles == geom.n;
    bp.ok = true;
    return bp;
}
static const BoardEntry* pick_board(const BoardPlan& bp, int ko_pad, int* kot_tiles, int force = 0) {
    if (!bp.ok) return nullptr;
    // the channel tile that needs the fewest rounds of workgroups over the 256 CUs (time of a round ~ its channel count);
    // ties go to the larger tile (the halo is staged once per workgroup)
    const BoardEntry* best = nullptr;
    long best_cost = 0;
    for (const auto& e : kBoardEntries) {
        if (ko_pad % e.kot != 0 || e.lds(bp.npos) > kMaxLds) continue;
        if (force && force != e.kot) continue;
        const long kts = ko_pad / e.kot, rounds = (bp.ntiles * kts + kNumCU - 1) / kNumCU, cost = rounds * e.kot;
        if (!best || cost < best_cost) { best = &e; best_cost = cost; *kot_tiles = (int)kts; }
    }
    return best;
}

struct Stat {
    int launches = 0;
    float ms = 0.f;
    double flops = 0, bytes = 0;
};

// ------------------------------------------------------------------ layers
struct ConvLayerDev {
    int cin = 0, cout = 0, k = 0;
    bool depthwise = false, with_bn_fold = false;
    std::vector<float> hw, hb;  // host tensors as handed over the ABI
    int cin_s = 0, cout_s = 0, wmt = 0, ko_pad = 0;
    void* w = nullptr;      // MFMA image, or [k*k][cs] fp32 for depthwise
    void* w_board = nullptr;  // the same image in board_row_channel order (fp16 3x3 layers a board kernel may run) ...
    float* bias_board = nullptr;  // ... and the bias in the same order: what a layer gets whose epilogue is the generated one
    float* bias = nullptr;  // [ko_pad] / [cs]
    float* w32 = nullptr;   // plain fp32 copy [cout][cin] for the tiny head convs
};
struct FcLayerDev {
    int in = 0, out = 0;
    std::vector<float> hw, hb;
    float* wt = nullptr;  // [in][out]
    float* b = nullptr;
    void* img16 = nullptr;  // SE units of the fp16 engine: the LDS-staging image of conv_board.h (BoardSeParams::w1h / w2h)
    int img_bytes = 0;      // bytes of one image (whole 1 KiB pieces)
    void* sx_img = nullptr; // ... and the per-channel-tile images of conv_board_sx.h (BoardSxParams::w1t / w2t), layers of 2-4 tiles of 128
    int sx_bytes = 0;
    FcDev dev() const { return FcDev{wt, b, in, out}; }
};

// ------------------------------------------------------------------ host-side images shared by the engine and the test taps
// fp16 images of an SE unit's two FCs for the LDS staging of board_se_stage (conv_board.h, BoardSeParams::w1h / w2h):
// the squeeze weights once per board size 2..board with the scaled-mean third of the pooled vector folded into the mean
// third (reference GlobalPooling<false>, se_unit.cc:9-40: pool = (mean, mean * (B-14)/10, max)), the excite weights with
// both bias vectors behind them.  sq_w [se][3C], ex_w [2C][se] as handed over the ABI.  false: the unit does not fit.
static bool make_se_images(int C, int se, int board, const float* sq_w, const float* sq_b, const float* ex_w, const float* ex_b,
                           std::vector<f16>* img1, std::vector<unsigned char>* img2, int* w1_bytes_out, int* w2_bytes_out) {
    if (se <= 0 || se % 4 || se > 512 || 2 * C > 512) return false;
    const int w1_bytes = round_up(2 * C * se * 2, 1024), w2_bytes = round_up(se * 2 * C * 2 + (2 * C + se) * 4, 1024);
    if ((size_t)w1_bytes + w2_bytes + 20 * 1024 > kMaxLds) return false;
    img1->assign((size_t)(board - 1) * (w1_bytes / 2), (f16)0.f);
    for (int bs = 2; bs <= board; ++bs) {
        const float sc = ((float)bs - 14.f) / 10.f;
        f16* d = img1->data() + (size_t)(bs - 2) * (w1_bytes / 2);
        for (int r = 0; r < 2 * C; ++r)
            for (int o = 0; o < se; ++o) {
                const float* w = sq_w + (size_t)o * 3 * C;
                d[(size_t)r * se + o] = (f16)(r < C ? w[r] + sc * w[C + r] : w[2 * C + (r - C)]);
            }
    }
    img2->assign(w2_bytes, 0);
    f16* h = (f16*)img2->data();
    for (int i = 0; i < se; ++i)
        for (int o = 0; o < 2 * C; ++o) h[((size_t)(i >> 2) * 2 * C + o) * 4 + (i & 3)] = (f16)ex_w[(size_t)o * se + i];
    float* bias = (float*)(img2->data() + (size_t)se * 2 * C * 2);
    std::copy(ex_b, ex_b + 2 * C, bias);
    std::copy(sq_b, sq_b + se, bias + 2 * C);
    *w1_bytes_out = w1_bytes;
    *w2_bytes_out = w2_bytes;
    return true;
}

// The same two FCs cut by 128-channel tile for conv_board_sx.h (a layer whose channels are split over kts workgroups): per tile kt
// the squeeze rows of its 128 channels (mean rows with the scaled-mean third folded in, once per board size; then the max rows),
// and the excite rows that produce its channels' gamma and beta, with their bias and the squeeze bias behind them.
static bool make_sx_images(int C, int se, int kts, int board, const float* sq_w, const float* sq_b, const float* ex_w, const float* ex_b,
                           std::vector<f16>* img1, std::vector<unsigned char>* img2, int* w1_bytes_out, int* w2_bytes_out) {
    if (se <= 0 || se % 4 || se > kSxSlots || kts < 2 || kts > 4 || C > kts * 128) return false;
    const int w1_bytes = round_up(256 * se * 2, 1024), w2_bytes = round_up(se * 256 * 2 + (256 + se) * 4, 1024);
    if (w1_bytes + w2_bytes > SxLds::stage_bytes) return false;
    img1->assign((size_t)kts * (board - 1) * (w1_bytes / 2), (f16)0.f);
    img2->assign((size_t)kts * w2_bytes, 0);
    for (int kt = 0; kt < kts; ++kt) {
        for (int bs = 2; bs <= board; ++bs) {
            const float sc = ((float)bs - 14.f) / 10.f;
            f16* d = img1->data() + ((size_t)kt * (board - 1) + (bs - 2)) * (w1_bytes / 2);
            for (int r = 0; r < 256; ++r) {
                const int c = kt * 128 + (r & 127);
                if (c >= C) continue;  // pad channels: x is 0 there, and their rows stay 0
                for (int o = 0; o < se; ++o) {
                    const float* w = sq_w + (size_t)o * 3 * C;
                    d[(size_t)r * se + o] = (f16)(r < 128 ? w[c] + sc * w[C + c] : w[2 * C + c]);
                }
            }
        }
        unsigned char* base = img2->data() + (size_t)kt * w2_bytes;
        f16* h = (f16*)base;
        float* bias = (float*)(base + (size_t)se * 256 * 2);
        for (int o = 0; o < 256; ++o) {
            const int c = kt * 128 + (o & 127);
            if (c >= C) continue;  // gamma = sigmoid(0), beta = 0 on x = 0
            const int row = o < 128 ? c : C + c;
            for (int i = 0; i < se; ++i) h[((size_t)(i >> 2) * 256 + o) * 4 + (i & 3)] = (f16)ex_w[(size_t)row * se + i];
            bias[o] = ex_b[row];
        }
        std::copy(sq_b, sq_b + se, bias + 256);
    }
    *w1_bytes_out = w1_bytes;
    *w2_bytes_out = w2_bytes;
    return true;
}

// Images of head_board_kernel (head_board.h): the stacked [policy | value] 1x1 head convolutions as one MFMA image
// (rows: policy channels rounded to a row tile of 16, then the value channels up to an even number of row tiles), the
// per-pixel weights (policy planes over the policy rows, ownership over the value rows) in the accumulator's channel
// order, the stacked bias.  Returns the kernel variant that fits, or nullptr (the separate head kernels run then).
struct HeadImages {
    std::vector<f16> img, img2;
    std::vector<float> bias;
    int PT = 0, VT = 0;
};
static HeadFn make_head_images(int C, int Cp, int Cv, int prob_ch, int board, const float* p_w, const float* p_b, const float* v_w,
                               const float* v_b, const float* prob_w, const float* own_w, HeadImages* out) {
    if (board * board > kHeadPix || prob_ch > 8) return nullptr;
    const int PT = round_up(Cp, 16), rows = round_up(PT + Cv, 32), VT = rows - PT, cs = round_up(C, 32), nch = cs / 32;
    HeadFn fn = nullptr;
    for (const auto& e : kHeadEntries)
        if (e.rt * 16 == rows && head_board_fits(rows, nch, e.depth)) { fn = e.fn; break; }
    if (!fn) return nullptr;
    // per-pixel weights in the accumulator's channel order: k-group kg of pair t holds stacked rows 32t + 4kg + s (s < 4)
    // and 32t + 16 + 4kg + (s - 4); row k < prob_ch = policy plane k over the policy rows, row prob_ch = ownership
    out->img2.assign((size_t)(rows / 32) * 4 * 16 * 8, (f16)0.f);
    for (int t = 0; t < rows / 32; ++t)
        for (int kg = 0; kg < 4; ++kg)
            for (int e = 0; e < 8; ++e) {
                const int ch = 32 * t + (e < 4 ? 4 * kg + e : 16 + 4 * kg + (e - 4));
                for (int k = 0; k < prob_ch; ++k)
                    if (ch < Cp) out->img2[(((size_t)t * 4 + kg) * 16 + k) * 8 + e] = (f16)prob_w[(size_t)k * Cp + ch];
                if (ch >= PT && ch - PT < Cv) out->img2[(((size_t)t * 4 + kg) * 16 + prob_ch) * 8 + e] = (f16)own_w[ch - PT];
            }
    out->img.assign((size_t)nch * 4 * rows * 8, (f16)0.f);
    out->bias.assign(rows, 0.f);
    for (int half = 0; half < 2; ++half) {
        const float* w = half ? v_w : p_w;
        const float* b = half ? v_b : p_b;
        const int cout = half ? Cv : Cp, r0 = half ? PT : 0;
        for (int ko = 0; ko < cout; ++ko) {
            out->bias[r0 + ko] = b[ko];
            for (int c = 0; c < C; ++c)
                out->img[(((size_t)(c / 32) * 4 + (c % 32) / 8) * rows + r0 + ko) * 8 + c % 8] = (f16)w[(size_t)ko * C + c];
        }
    }
    out->PT = PT;
    out->VT = VT;
    return fn;
}

// the persistent tower kernels (conv_tower.h) out of the embedded code object: [0] 256-channel tile, [1] 128-channel tile
}  // namespace sayuri
extern "C" const unsigned char sayuri_tower_hsaco[];
extern "C" const unsigned long long sayuri_tower_hsaco_size;
namespace sayuri {
static int load_tower_module(hipModule_t* mod, hipFunction_t fn[2]) {
    // an EMPTY blob: the build went on without the persistent kernel because tower_seam.py did not recognise the compiler's
    // assembly (sayuri_amd/_build.py tower_blob_from_asm); the caller reports the fallback and launches per layer
    if (sayuri_tower_hsaco_size == 0) return fail("this build carries no persistent tower kernel: tower_seam.py rejected the compiler's assembly at build time");
    HIP_OK(hipModuleLoadData(mod, sayuri_tower_hsaco));
    HIP_OK(hipModuleGetFunction(&fn[0], *mod, "_ZN6sayuri17conv_tower_kernelILi4EEEvPKNS_10TowerLayerE"));
    HIP_OK(hipModuleGetFunction(&fn[1], *mod, "_ZN6sayuri17conv_tower_kernelILi2EEEvPKNS_10TowerLayerE"));
    return 0;
}

}  // namespace sayuri
