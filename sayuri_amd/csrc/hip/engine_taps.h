// engine_taps.h -- layer-level test taps of the C-ABI (sayuri_hip_test_*): one kernel family at a time on host tensors, for the
// parity tests (tests/test_gpu_layers.py, test_gpu_smallops.py).  Not used by the pipe.
#pragma once
#include "engine_graph.h"

// ---------------------------------------------------------------------- layer-level test tap
// Drives ONE convolution kernel directly (host-side layout conversion in, out) so the parity
// tests can localise a defect to a layer kind.

namespace sayuri {

template <typename T>
static int test_conv_impl(int device, int n, const int* board_sizes, int max_board, int cin, int cout, int k,
                          int depthwise, int act, int post_residual, const float* x, const float* w, const float* bias,
                          const float* res, float* y) {
    HIP_OK(hipSetDevice(device));
    enable_big_lds<T>();
    HostGeom hg;
    hg.n = n;
    hg.bsz.assign(board_sizes, board_sizes + n);
    hg.off.resize(n + 1);
    hg.off[0] = 0;
    for (int i = 0; i < n; ++i) {
        if (hg.bsz[i] < 2 || hg.bsz[i] > max_board) return fail("test_conv: bad board size");
        hg.off[i + 1] = hg.off[i] + hg.bsz[i] * hg.bsz[i];
    }
    hg.total = hg.off[n];
    const int slot = max_board * max_board;
    const int cin_s = round_up(depthwise ? cout : cin, 32), cout_s = round_up(cout, 32);
    std::vector<void*> allocs;
    auto cleanup = [&] { for (void* p : allocs) (void)hipFree(p); };
    auto dalloc = [&](size_t bytes) -> void* {
        void* p = nullptr;
        if (hipMalloc(&p, std::max<size_t>(bytes, 256)) != hipSuccess) return nullptr;
        (void)hipMemset(p, 0, std::max<size_t>(bytes, 256));
        allocs.push_back(p);
        return p;
    };
    // host NCHW (compact per sample) -> compact NHWC
    auto to_nhwc = [&](const float* src, int C, int cs) {
        std::vector<T> h((size_t)n * slot * cs, (T)0.f);
        size_t so = 0;
        for (int i = 0; i < n; ++i) {
            const int S = hg.bsz[i] * hg.bsz[i];
            for (int c = 0; c < C; ++c)
                for (int p = 0; p < S; ++p) h[((size_t)i * slot + p) * cs + c] = (T)src[so + (size_t)c * S + p];
            so += (size_t)C * S;
        }
        return h;
    };
    const int xin_c = depthwise ? cout : cin;
    std::vector<T> hx = to_nhwc(x, xin_c, cin_s);
    T* dx = (T*)dalloc(hx.size() * sizeof(T) + kZeroPrefix);
    if (dx) dx += kZeroPrefix / sizeof(T);  // conv_board.h reads its halo cells from a zero prefix in front of the activations
    T* dy = (T*)dalloc((size_t)n * slot * cout_s * sizeof(T));
    T* dres = nullptr;
    if (!dx || !dy) { cleanup(); return fail("test_conv: hipMalloc failed"); }
    HIP_OK(hipMemcpy(dx, hx.data(), hx.size() * sizeof(T), hipMemcpyHostToDevice));
    if (res) {
        std::vector<T> hr = to_nhwc(res, cout, cout_s);
        dres = (T*)dalloc(hr.size() * sizeof(T));
        if (!dres) { cleanup(); return fail("test_conv: hipMalloc failed"); }
        HIP_OK(hipMemcpy(dres, hr.data(), hr.size() * sizeof(T), hipMemcpyHostToDevice));
    }
    int* d_off = (int*)dalloc(sizeof(int) * (n + 1));
    int* d_bsz = (int*)dalloc(sizeof(int) * n);
    HIP_OK(hipMemcpy(d_off, hg.off.data(), sizeof(int) * (n + 1), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_bsz, hg.bsz.data(), sizeof(int) * n, hipMemcpyHostToDevice));
    const BatchGeom g{d_off, d_bsz, n, hg.total, slot};

    if (depthwise) {
        g_test_conv_kind = 3;
        const int kk = k * k;
        std::vector<float> wt((size_t)kk * cout_s, 0.f), b(cout_s, 0.f);
        for (int c = 0; c < cout; ++c) {
            for (int t = 0; t < kk; ++t) wt[(size_t)t * cout_s + c] = w[(size_t)c * kk + t];
            b[c] = bias ? bias[c] : 0.f;
        }
        float* dw = (float*)dalloc(wt.size() * 4);
        float* db = (float*)dalloc(b.size() * 4);
        HIP_OK(hipMemcpy(dw, wt.data(), wt.size() * 4, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice));
        const int EPP = ElemTraits<T>::kPieceElems;
        const size_t total = (size_t)hg.total * (cout_s / EPP);
        hipLaunchKernelGGL(depthwise_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, 0, (const T*)dx,
                           post_residual ? (const T*)dres : (const T*)nullptr, dy, (const float*)dw, (const float*)db, g,
                           cout, cout_s, k, act);
    } else {
        const int wmt = pick_wmt(cout_s, sizeof(T) == 2);
        const int kot = wmt * 32, ko_pad = round_up(cout_s, kot), taps = k * k, nch = cin_s / 32;
        std::vector<T> img((size_t)taps * nch * 4 * ko_pad * 8, (T)0.f);
        for (int t = 0; t < taps; ++t)
            for (int ch = 0; ch < nch; ++ch)
                for (int kg = 0; kg < 4; ++kg)
                    for (int ko = 0; ko < cout; ++ko)
                        for (int e = 0; e < 8; ++e) {
                            const int c = ch * 32 + kg * 8 + e;
                            if (c >= cin) continue;
                            img[((((size_t)t * nch + ch) * 4 + kg) * ko_pad + ko) * 8 + e] =
                                (T)w[((size_t)ko * cin + c) * taps + t];
                        }
        std::vector<float> b(ko_pad, 0.f);
        if (bias) std::copy(bias, bias + cout, b.begin());
        T* dw = (T*)dalloc(img.size() * sizeof(T));
        float* db = (float*)dalloc(b.size() * 4);
        HIP_OK(hipMemcpy(dw, img.data(), img.size() * sizeof(T), hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice));
        int g_ntiles = 0;
        const GldsEntry* ge = nullptr;
        bool board_done = false;
        if (sizeof(T) == 2 && k == 3) {
            enable_big_lds_glds();
            const ConvOverride cov = EngineFlags::from_env().conv;
            const BoardPlan plan = board_plan(hg, cov);
            int kot_tiles = 0;
            const BoardEntry* be = plan.fill >= cov.board_min_fill ? pick_board(plan, ko_pad, &kot_tiles) : nullptr;
            if (be) {
                int* tsrc = (int*)dalloc(sizeof(int) * (size_t)plan.ntiles * plan.npos);
                int2* tpix = (int2*)dalloc(sizeof(int2) * (size_t)plan.ntiles * kBoardPT);
                int* tcols = (int*)dalloc(sizeof(int) * (size_t)plan.ntiles);
                if (!tsrc || !tpix || !tcols) { cleanup(); return fail("test_conv: hipMalloc failed"); }
                hipLaunchKernelGGL(board_setup_kernel, dim3(plan.ntiles), dim3(256), 0, 0, g, plan.npos, tsrc, tpix, tcols);
                BoardParams bp;
                std::memset(&bp, 0, sizeof(bp));
                bp.tab_src = tsrc; bp.tab_pix = tpix; bp.tab_cols = tcols; bp.npos = plan.npos; bp.dbg = nullptr;
                bp.uniform_info = plan.uniform_info;
                bp.arith = (plan.single && plan.uniform_info >= 0) ? 1 : 0;
                ConvParams& p = bp.c;
                p.in = dx; p.w = dw; p.bias = db; p.res = dres; p.out = dy; p.g = g;
                p.cin_s = cin_s; p.cout_s = cout_s; p.ko_pad = ko_pad; p.taps = 9; p.act = act; p.npos = 0;
                p.num_pix_tiles = plan.ntiles;
                hipLaunchKernelGGL(be->fn, dim3(plan.ntiles * kot_tiles), dim3(512), be->lds(plan.npos), 0, bp);
                HIP_OK(hipGetLastError());
                HIP_OK(hipDeviceSynchronize());
                board_done = true;
                g_test_conv_kind = 2;
            }
        }
        if (sizeof(T) == 2 && k == 3 && !board_done) {
            enable_big_lds_glds();
            ge = pick_glds(hg, ko_pad, &g_ntiles, EngineFlags::from_env().conv);
        }
        if (ge) {
            float* dz = (float*)dalloc(256);
            int* tsrc = (int*)dalloc(sizeof(int) * (size_t)g_ntiles * ge->npos);
            int2* tpix = (int2*)dalloc(sizeof(int2) * (size_t)g_ntiles * ge->pt);
            if (!dz || !tsrc || !tpix) { cleanup(); return fail("test_conv: hipMalloc failed"); }
            hipLaunchKernelGGL(ge->setup, dim3(g_ntiles), dim3(256), 0, 0, g, tsrc, tpix);
            GldsParams gp;
            gp.tab_src = tsrc;
            gp.tab_pix = tpix;
            ConvParams& p = gp.c;
            p.in = dx; p.w = dw; p.bias = db; p.res = dres; p.out = dy; p.g = g;
            p.cin_s = cin_s; p.cout_s = cout_s; p.ko_pad = ko_pad; p.taps = 9; p.act = act; p.npos = 0;
            p.num_pix_tiles = g_ntiles;
            gp.zeros = dz;
            hipLaunchKernelGGL(ge->fn, dim3(g_ntiles * (ko_pad / (ge->wmt * 32))), dim3(512), ge->lds, 0, gp);
            g_test_conv_kind = 1;
            HIP_OK(hipGetLastError());
            HIP_OK(hipDeviceSynchronize());
        }
        const typename ConvKernelTable<T>::Entry* best = nullptr;
        int best_npos = 0;
        for (const auto& e : ConvKernelTable<T>::entries()) {
            if (ge || board_done) break;
            if (e.wmt != wmt) continue;
            int npos, nsub;
            hg.tile_bounds(64 * e.wnt, &npos, &nsub);
            if (npos > e.npos_cap || nsub > kMaxSub || e.lds(npos) > kMaxLds) continue;
            if (!best || e.wnt > best->wnt) { best = &e; best_npos = npos; }
        }
        if (!best && !ge && !board_done) { cleanup(); return fail("test_conv: no tile configuration fits"); }
        if (best) {
        g_test_conv_kind = 0;
        ConvParams p;
        p.in = dx; p.w = dw; p.bias = db; p.res = dres; p.out = dy; p.g = g;
        p.cin_s = cin_s; p.cout_s = cout_s; p.ko_pad = ko_pad; p.taps = taps; p.act = act;
        p.npos = best_npos;
        const int PT = 64 * best->wnt;
        p.num_pix_tiles = (hg.total + PT - 1) / PT;
        hipLaunchKernelGGL(best->fn, dim3(p.num_pix_tiles * (ko_pad / kot)), dim3(512), best->lds(best_npos), 0, p);
        }
    }
    HIP_OK(hipGetLastError());
    HIP_OK(hipDeviceSynchronize());
    std::vector<T> hy((size_t)n * slot * cout_s);
    HIP_OK(hipMemcpy(hy.data(), dy, hy.size() * sizeof(T), hipMemcpyDeviceToHost));
    size_t so = 0;
    for (int i = 0; i < n; ++i) {
        const int S = hg.bsz[i] * hg.bsz[i];
        for (int c = 0; c < cout; ++c)
            for (int pp = 0; pp < S; ++pp) y[so + (size_t)c * S + pp] = (float)hy[((size_t)i * slot + pp) * cout_s + c];
        so += (size_t)cout * S;
    }
    cleanup();
    return 0;
}

}  // namespace sayuri

// ---------------------------------------------------------------------- small-op test taps
namespace sayuri {

struct TestGeom {
    HostGeom hg;
    int slot = 0;
    int* d_off = nullptr;
    int* d_bsz = nullptr;
    BatchGeom g{};
};

class TestArena {  // device allocations of one tap call, freed at scope exit
public:
    ~TestArena() { for (void* p : ptrs_) (void)hipFree(p); }
    void* alloc(size_t bytes) {
        void* p = nullptr;
        if (hipMalloc(&p, std::max<size_t>(bytes, 256)) != hipSuccess) return nullptr;
        (void)hipMemset(p, 0, std::max<size_t>(bytes, 256));
        ptrs_.push_back(p);
        return p;
    }
    template <typename U> U* upload(const std::vector<U>& h) {
        U* d = (U*)alloc(h.size() * sizeof(U));
        if (d && hipMemcpy(d, h.data(), h.size() * sizeof(U), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
        return d;
    }
private:
    std::vector<void*> ptrs_;
};

static int make_test_geom(TestArena& A, int n, const int* board_sizes, int max_board, TestGeom* tg) {
    tg->hg.n = n;
    tg->hg.bsz.assign(board_sizes, board_sizes + n);
    tg->hg.off.assign(n + 1, 0);
    for (int i = 0; i < n; ++i) {
        if (tg->hg.bsz[i] < 2 || tg->hg.bsz[i] > max_board) return fail("test tap: bad board size");
        tg->hg.off[i + 1] = tg->hg.off[i] + tg->hg.bsz[i] * tg->hg.bsz[i];
    }
    tg->hg.total = tg->hg.off[n];
    tg->slot = max_board * max_board;
    tg->d_off = A.upload(tg->hg.off);
    tg->d_bsz = A.upload(tg->hg.bsz);
    if (!tg->d_off || !tg->d_bsz) return fail("test tap: hipMalloc failed");
    tg->g = BatchGeom{tg->d_off, tg->d_bsz, n, tg->hg.total, tg->slot};
    return 0;
}
// host NCHW (compact per sample) <-> compact NHWC
template <typename T> static std::vector<T> nchw_to_nhwc(const TestGeom& tg, const float* src, int C, int cs) {
    std::vector<T> h((size_t)tg.hg.n * tg.slot * cs, (T)0.f);
    size_t so = 0;
    for (int i = 0; i < tg.hg.n; ++i) {
        const int S = tg.hg.bsz[i] * tg.hg.bsz[i];
        for (int c = 0; c < C; ++c)
            for (int p = 0; p < S; ++p) h[((size_t)i * tg.slot + p) * cs + c] = (T)src[so + (size_t)c * S + p];
        so += (size_t)C * S;
    }
    return h;
}
static std::vector<float> fc_transposed(const float* w, int in, int out) {  // [out][in] -> [in][out]
    std::vector<float> t((size_t)in * out);
    for (int o = 0; o < out; ++o)
        for (int i = 0; i < in; ++i) t[(size_t)i * out + o] = w[(size_t)o * in + i];
    return t;
}

template <typename T>
static int test_se_unit_impl(int device, int n, const int* board_sizes, int max_board, int C, int se, int act, const float* x,
                             const float* res, const float* w1, const float* b1, const float* w2, const float* b2, float* y,
                             float* gate_out) {
    HIP_OK(hipSetDevice(device));
    TestArena A;
    TestGeom tg;
    if (make_test_geom(A, n, board_sizes, max_board, &tg)) return -1;
    const int cs = round_up(C, 32);
    constexpr int EPP = ElemTraits<T>::kPieceElems;
    if (cs / EPP > 256) return fail("test_se_unit: too many channels");
    T* dx = A.upload(nchw_to_nhwc<T>(tg, x, C, cs));
    T* dres = res ? A.upload(nchw_to_nhwc<T>(tg, res, C, cs)) : nullptr;
    float* dw1 = A.upload(fc_transposed(w1, 3 * C, se));
    float* dw2 = A.upload(fc_transposed(w2, se, 2 * C));
    float* db1 = A.upload(std::vector<float>(b1, b1 + se));
    float* db2 = A.upload(std::vector<float>(b2, b2 + 2 * C));
    float* separt = (float*)A.alloc(sizeof(float) * (size_t)n * kSeSplit * 2 * cs);
    float* gate = (float*)A.alloc(sizeof(float) * (size_t)n * 2 * cs);
    if (!dx || (res && !dres) || !dw1 || !dw2 || !db1 || !db2 || !separt || !gate) return fail("test_se_unit: hipMalloc failed");
    const FcDev sq{dw1, db1, 3 * C, se}, ex{dw2, db2, se, 2 * C};
    hipLaunchKernelGGL(se_pool_kernel<T>, dim3(n * kSeSplit), dim3(256), 0, 0, (const T*)dx, separt, tg.g, cs);
    hipLaunchKernelGGL(se_fc_kernel, dim3(n), dim3(kSeFcThreads), sizeof(float) * (3 * C + se + kSeFcThreads), 0, (const float*)separt, gate, tg.g, C, cs, sq,
                       ex, act);
    const int ppr = cs / EPP;
    const dim3 grid((tg.slot * ppr + 256 * kScaleUnroll - 1) / (256 * kScaleUnroll), n);
    hipLaunchKernelGGL(se_scale_kernel<T>, grid, dim3(256), 0, 0, (const T*)dx, (const T*)dres, dx, (const float*)gate, tg.g, C, cs, act);
    HIP_OK(hipGetLastError());
    HIP_OK(hipDeviceSynchronize());
    std::vector<T> hy((size_t)n * tg.slot * cs);
    HIP_OK(hipMemcpy(hy.data(), dx, hy.size() * sizeof(T), hipMemcpyDeviceToHost));
    size_t so = 0;
    for (int i = 0; i < n; ++i) {
        const int S = tg.hg.bsz[i] * tg.hg.bsz[i];
        for (int c = 0; c < C; ++c)
            for (int pp = 0; pp < S; ++pp) y[so + (size_t)c * S + pp] = (float)hy[((size_t)i * tg.slot + pp) * cs + c];
        so += (size_t)C * S;
    }
    if (gate_out) {
        std::vector<float> hg((size_t)n * 2 * cs);
        HIP_OK(hipMemcpy(hg.data(), gate, hg.size() * 4, hipMemcpyDeviceToHost));
        for (int i = 0; i < n; ++i)
            for (int c = 0; c < C; ++c) {
                gate_out[(size_t)i * 2 * C + c] = hg[(size_t)i * 2 * cs + c];
                gate_out[(size_t)i * 2 * C + C + c] = hg[(size_t)i * 2 * cs + cs + c];
            }
    }
    return 0;
}

template <typename T>
static int test_head_tail_impl(int device, int n, const int* board_sizes, int max_board, int Cp, int Cv, int prob_ch, int pass_outs,
                               int misc_outs, int act, const float* pconv, const float* vconv, const float* const* w, float* prob,
                               float* pass, float* misc, float* own) {
    // w: p_inter_w, p_inter_b, pass_w, pass_b, v_inter_w, v_inter_b, v_misc_w, v_misc_b, prob_w, prob_b, own_w, own_b
    HIP_OK(hipSetDevice(device));
    TestArena A;
    TestGeom tg;
    if (make_test_geom(A, n, board_sizes, max_board, &tg)) return -1;
    if (prob_ch > 8) return fail("test_head_tail: too many policy planes");
    const int cs_p = round_up(Cp, 32), cs_v = round_up(Cv, 32), B2 = max_board * max_board;
    T* dp = A.upload(nchw_to_nhwc<T>(tg, pconv, Cp, cs_p));
    T* dv = A.upload(nchw_to_nhwc<T>(tg, vconv, Cv, cs_v));
    HeadParams h;
    float* d_pi = A.upload(fc_transposed(w[0], 3 * Cp, Cp));
    float* d_pib = A.upload(std::vector<float>(w[1], w[1] + Cp));
    float* d_pw = A.upload(fc_transposed(w[2], Cp, pass_outs));
    float* d_pwb = A.upload(std::vector<float>(w[3], w[3] + pass_outs));
    float* d_vi = A.upload(fc_transposed(w[4], 3 * Cv, 3 * Cv));
    float* d_vib = A.upload(std::vector<float>(w[5], w[5] + 3 * Cv));
    float* d_vm = A.upload(fc_transposed(w[6], 3 * Cv, misc_outs));
    float* d_vmb = A.upload(std::vector<float>(w[7], w[7] + misc_outs));
    float* d_prw = A.upload(std::vector<float>(w[8], w[8] + (size_t)prob_ch * Cp));
    float* d_prb = A.upload(std::vector<float>(w[9], w[9] + prob_ch));
    float* d_ow = A.upload(std::vector<float>(w[10], w[10] + Cv));
    float* d_ob = A.upload(std::vector<float>(w[11], w[11] + 1));
    float* d_prob = (float*)A.alloc(sizeof(float) * (size_t)n * prob_ch * B2);
    float* d_pass = (float*)A.alloc(sizeof(float) * (size_t)n * pass_outs);
    float* d_misc = (float*)A.alloc(sizeof(float) * (size_t)n * misc_outs);
    float* d_own = (float*)A.alloc(sizeof(float) * (size_t)n * B2);
    if (!dp || !dv || !d_pi || !d_pib || !d_pw || !d_pwb || !d_vi || !d_vib || !d_vm || !d_vmb || !d_prw || !d_prb || !d_ow || !d_ob ||
        !d_prob || !d_pass || !d_misc || !d_own)
        return fail("test_head_tail: hipMalloc failed");
    h.p_inter = FcDev{d_pi, d_pib, 3 * Cp, Cp};
    h.pass_fc = FcDev{d_pw, d_pwb, Cp, pass_outs};
    h.v_inter = FcDev{d_vi, d_vib, 3 * Cv, 3 * Cv};
    h.v_misc = FcDev{d_vm, d_vmb, 3 * Cv, misc_outs};
    h.prob_w = d_prw; h.prob_b = d_prb; h.own_w = d_ow; h.own_b = d_ob;
    h.Cp = Cp; h.cs_p = cs_p; h.Cv = Cv; h.cs_v = cs_v; h.prob_ch = prob_ch; h.act = act; h.board = max_board;
    h.prob = d_prob; h.pass = d_pass; h.misc = d_misc; h.own = d_own; h.perm = nullptr;
    const int maxc = std::max(Cp, Cv);
    hipLaunchKernelGGL(head_tail_kernel<T>, dim3(2 * n), dim3(256), sizeof(float) * (7 * maxc + 512), 0, (const T*)dp, (const T*)dv, tg.g, h);
    HIP_OK(hipGetLastError());
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpy(prob, d_prob, sizeof(float) * (size_t)n * prob_ch * B2, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(pass, d_pass, sizeof(float) * (size_t)n * pass_outs, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(misc, d_misc, sizeof(float) * (size_t)n * misc_outs, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(own, d_own, sizeof(float) * (size_t)n * B2, hipMemcpyDeviceToHost));
    return 0;
}

// The convolution with the SE unit inside it (conv_board_se_kernel, or the same stage inside the persistent tower kernel
// when via_tower != 0): C -> C 3x3 convolution + bias, then the unit's pool -> FC -> FC -> act(sigmoid(g) x + b + res).
// Returns 1 when the fused kernel does not apply to this batch (several samples per tile, channel tile not 128 / 256).
static int test_conv_se_impl(int device, int n, const int* board_sizes, int max_board, int C, int se, int act, int via_tower,
                             const float* x, const float* w, const float* bias, const float* res, const float* w1, const float* b1,
                             const float* w2, const float* b2, float* y) {
    typedef f16 T;
    HIP_OK(hipSetDevice(device));
    enable_big_lds_glds();
    TestArena A;
    TestGeom tg;
    if (make_test_geom(A, n, board_sizes, max_board, &tg)) return -1;
    const int cs = round_up(C, 32), wmt = pick_wmt(cs, true), ko_pad = round_up(cs, wmt * 32);
    const BoardPlan plan = board_plan(tg.hg, ConvOverride{});
    const BoardEntry* be = nullptr;
    if (plan.ok)
        for (const auto& e : kBoardEntries)
            if (e.fn_se && e.kot == ko_pad && e.lds(plan.npos) <= kMaxLds) be = &e;
    if (!be || !plan.single || C > be->kot) return 1;
    std::vector<f16> img1;
    std::vector<unsigned char> img2;
    int w1_bytes = 0, w2_bytes = 0;
    const bool staged = make_se_images(C, se, max_board, w1, b1, w2, b2, &img1, &img2, &w1_bytes, &w2_bytes);
    if (!staged && (se % 4 || se > 512 || (2 * C) % 4 || 512 % (se / 4) || 512 % (2 * C / 4))) return 1;
    // activations (with the zero prefix the board kernels read their halo cells from), weights image, tables
    std::vector<T> hx = nchw_to_nhwc<T>(tg, x, C, cs);
    T* dx = (T*)A.alloc(hx.size() * sizeof(T) + kZeroPrefix);
    if (!dx) return fail("test_conv_se: hipMalloc failed");
    dx += kZeroPrefix / sizeof(T);
    HIP_OK(hipMemcpy(dx, hx.data(), hx.size() * sizeof(T), hipMemcpyHostToDevice));
    T* dres = res ? A.upload(nchw_to_nhwc<T>(tg, res, C, cs)) : nullptr;
    T* dy = (T*)A.alloc((size_t)n * tg.slot * cs * sizeof(T));
    const int nch = cs / 32;
    std::vector<T> img((size_t)9 * nch * 4 * ko_pad * 8, (T)0.f);
    for (int t = 0; t < 9; ++t)
        for (int ko = 0; ko < C; ++ko)
            for (int c = 0; c < C; ++c)
                img[((((size_t)t * nch + c / 32) * 4 + (c % 32) / 8) * ko_pad + ko) * 8 + c % 8] = (T)w[((size_t)ko * C + c) * 9 + t];
    std::vector<float> hb(ko_pad, 0.f);
    if (bias) std::copy(bias, bias + C, hb.begin());
    // through the tower a Mish layer with computed table entries takes the generated epilogue: weights and bias in
    // board_row_channel order (Engine::board_row_order_ok)
    const bool row_order = via_tower && board_uses_row_order(be->kot) && (act == kMish || act == kReLU || act == kIdentity) && plan.single && plan.uniform_info >= 0 && cs == be->kot &&
                           !EngineFlags::off("SAYURI_TOWER_GEN_EPI");
    if (row_order) {
        std::vector<float> hbb(ko_pad);
        for (int r = 0; r < ko_pad; ++r) hbb[r] = hb[board_row_channel(r)];
        hb.swap(hbb);
    }
    T* dw = A.upload(row_order ? board_row_order(img, ko_pad) : img);
    float* db = A.upload(hb);
    float* dw1 = A.upload(fc_transposed(w1, 3 * C, se));
    float* dw2 = A.upload(fc_transposed(w2, se, 2 * C));
    float* db1 = A.upload(std::vector<float>(b1, b1 + se));
    float* db2 = A.upload(std::vector<float>(b2, b2 + 2 * C));
    f16* d1 = staged ? A.upload(img1) : nullptr;
    unsigned char* d2 = staged ? A.upload(img2) : nullptr;
    int* tsrc = (int*)A.alloc(sizeof(int) * (size_t)plan.ntiles * plan.npos);
    int2* tpix = (int2*)A.alloc(sizeof(int2) * (size_t)plan.ntiles * kBoardPT);
    int* tcols = (int*)A.alloc(sizeof(int) * (size_t)plan.ntiles);
    if ((res && !dres) || !dy || !dw || !db || !dw1 || !dw2 || !db1 || !db2 || (staged && (!d1 || !d2)) || !tsrc || !tpix || !tcols)
        return fail("test_conv_se: hipMalloc failed");
    hipLaunchKernelGGL(board_setup_kernel, dim3(plan.ntiles), dim3(256), 0, 0, tg.g, plan.npos, tsrc, tpix, tcols);
    BoardSeParams sp;
    std::memset(&sp, 0, sizeof(sp));
    BoardParams& bp = sp.b;
    bp.tab_src = tsrc; bp.tab_pix = tpix; bp.tab_cols = tcols; bp.npos = plan.npos;
    bp.uniform_info = plan.uniform_info;
    bp.arith = (plan.single && plan.uniform_info >= 0) ? 1 : 0;
    bp.row_order = row_order ? 1 : 0;
    ConvParams& p = bp.c;
    p.in = dx; p.w = dw; p.bias = db; p.res = dres; p.out = dy; p.g = tg.g;
    p.cin_s = cs; p.cout_s = cs; p.ko_pad = ko_pad; p.taps = 9; p.act = act; p.num_pix_tiles = plan.ntiles;
    sp.squeeze = FcDev{dw1, db1, 3 * C, se};
    sp.excite = FcDev{dw2, db2, se, 2 * C};
    sp.C = C;
    sp.w1h = d1; sp.w2h = d2; sp.w1_bytes = w1_bytes; sp.w2_bytes = w2_bytes;
    hipModule_t mod = nullptr;
    if (via_tower) {
        hipFunction_t fn[2] = {nullptr, nullptr};
        if (load_tower_module(&mod, fn)) return -1;
        TowerLayer t;
        std::memset(&t, 0, sizeof(t));
        TowerLayer* dt = (TowerLayer*)A.alloc(sizeof(TowerLayer));
        if (!dt) return fail("test_conv_se: hipMalloc failed");
        t.self = dt; t.last = 1; t.has_se = 1; t.sp = sp;
        HIP_OK(hipMemcpy(dt, &t, sizeof(t), hipMemcpyHostToDevice));
        const TowerLayer* arg = dt;
        void* params[] = {(void*)&arg};
        HIP_OK(hipModuleLaunchKernel(fn[be->kot == 256 ? 0 : 1], plan.ntiles, 1, 1, 512, 1, 1, 0, nullptr, params, nullptr));
    } else {
        hipLaunchKernelGGL(be->fn_se, dim3(plan.ntiles), dim3(512), be->lds(plan.npos), 0, sp);
    }
    HIP_OK(hipGetLastError());
    HIP_OK(hipDeviceSynchronize());
    if (mod) (void)hipModuleUnload(mod);
    std::vector<T> hy((size_t)n * tg.slot * cs);
    HIP_OK(hipMemcpy(hy.data(), dy, hy.size() * sizeof(T), hipMemcpyDeviceToHost));
    size_t so = 0;
    for (int i = 0; i < n; ++i) {
        const int S = tg.hg.bsz[i] * tg.hg.bsz[i];
        for (int c = 0; c < C; ++c)
            for (int pp = 0; pp < S; ++pp) y[so + (size_t)c * S + pp] = (float)hy[((size_t)i * tg.slot + pp) * cs + c];
        so += (size_t)C * S;
    }
    return 0;
}

// Both heads of a sample in one workgroup (head_board_kernel): trunk [n][C][bs*bs] -> the four output tensors.
// Returns 1 when no head_board_kernel variant fits these channel counts (the engine then runs conv1x1 x2 + head_tail).
static int test_head_board_impl(int device, int n, const int* board_sizes, int max_board, int C, int Cp, int Cv, int prob_ch,
                                int pass_outs, int misc_outs, int act, const float* trunk, const float* p_w, const float* p_b,
                                const float* v_w, const float* v_b, const float* const* w, float* prob, float* pass, float* misc,
                                float* own) {
    typedef f16 T;
    HIP_OK(hipSetDevice(device));
    enable_big_lds_glds();
    TestArena A;
    TestGeom tg;
    if (make_test_geom(A, n, board_sizes, max_board, &tg)) return -1;
    HeadImages hi;
    const HeadFn fn = make_head_images(C, Cp, Cv, prob_ch, max_board, p_w, p_b, v_w, v_b, w[8], w[10], &hi);
    if (!fn) return 1;
    const int cs = round_up(C, 32), B2 = max_board * max_board;
    T* dt = A.upload(nchw_to_nhwc<T>(tg, trunk, C, cs));
    HeadBoardParams hp;
    std::memset(&hp, 0, sizeof(hp));
    HeadParams& h = hp.h;
    float* d_pi = A.upload(fc_transposed(w[0], 3 * Cp, Cp));
    float* d_pib = A.upload(std::vector<float>(w[1], w[1] + Cp));
    float* d_pw = A.upload(fc_transposed(w[2], Cp, pass_outs));
    float* d_pwb = A.upload(std::vector<float>(w[3], w[3] + pass_outs));
    float* d_vi = A.upload(fc_transposed(w[4], 3 * Cv, 3 * Cv));
    float* d_vib = A.upload(std::vector<float>(w[5], w[5] + 3 * Cv));
    float* d_vm = A.upload(fc_transposed(w[6], 3 * Cv, misc_outs));
    float* d_vmb = A.upload(std::vector<float>(w[7], w[7] + misc_outs));
    float* d_prw = A.upload(std::vector<float>(w[8], w[8] + (size_t)prob_ch * Cp));
    float* d_prb = A.upload(std::vector<float>(w[9], w[9] + prob_ch));
    float* d_ow = A.upload(std::vector<float>(w[10], w[10] + Cv));
    float* d_ob = A.upload(std::vector<float>(w[11], w[11] + 1));
    f16* d_img = A.upload(hi.img);
    f16* d_img2 = A.upload(hi.img2);
    float* d_bias = A.upload(hi.bias);
    float* d_prob = (float*)A.alloc(sizeof(float) * (size_t)n * prob_ch * B2);
    float* d_pass = (float*)A.alloc(sizeof(float) * (size_t)n * pass_outs);
    float* d_misc = (float*)A.alloc(sizeof(float) * (size_t)n * misc_outs);
    float* d_own = (float*)A.alloc(sizeof(float) * (size_t)n * B2);
    if (!dt || !d_pi || !d_pib || !d_pw || !d_pwb || !d_vi || !d_vib || !d_vm || !d_vmb || !d_prw || !d_prb || !d_ow || !d_ob || !d_img ||
        !d_img2 || !d_bias || !d_prob || !d_pass || !d_misc || !d_own)
        return fail("test_head_board: hipMalloc failed");
    h.p_inter = FcDev{d_pi, d_pib, 3 * Cp, Cp};
    h.pass_fc = FcDev{d_pw, d_pwb, Cp, pass_outs};
    h.v_inter = FcDev{d_vi, d_vib, 3 * Cv, 3 * Cv};
    h.v_misc = FcDev{d_vm, d_vmb, 3 * Cv, misc_outs};
    h.prob_w = d_prw; h.prob_b = d_prb; h.own_w = d_ow; h.own_b = d_ob;
    h.Cp = Cp; h.cs_p = round_up(Cp, 32); h.Cv = Cv; h.cs_v = round_up(Cv, 32); h.prob_ch = prob_ch; h.act = act; h.board = max_board;
    h.prob = d_prob; h.pass = d_pass; h.misc = d_misc; h.own = d_own; h.perm = nullptr;
    hp.trunk = dt; hp.w = d_img; hp.w2 = d_img2; hp.bias = d_bias; hp.g = tg.g; hp.cs = cs; hp.PT = hi.PT; hp.VT = hi.VT; hp.dbg = nullptr;
    hipLaunchKernelGGL(fn, dim3(n), dim3(512), kMaxLds, 0, hp);
    HIP_OK(hipGetLastError());
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpy(prob, d_prob, sizeof(float) * (size_t)n * prob_ch * B2, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(pass, d_pass, sizeof(float) * (size_t)n * pass_outs, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(misc, d_misc, sizeof(float) * (size_t)n * misc_outs, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(own, d_own, sizeof(float) * (size_t)n * B2, hipMemcpyDeviceToHost));
    return 0;
}

}  // namespace sayuri

extern "C" int sayuri_hip_test_conv_se(int device, int n, const int* board_sizes, int max_board, int channels, int se_size, int act,
                                       int via_tower, const float* x, const float* w, const float* bias, const float* res,
                                       const float* w1, const float* b1, const float* w2, const float* b2, float* y) {
    if (!board_sizes || !x || !w || !w1 || !b1 || !w2 || !b2 || !y || n <= 0) return fail("test_conv_se: bad argument");
    return test_conv_se_impl(device, n, board_sizes, max_board, channels, se_size, act, via_tower, x, w, bias, res, w1, b1, w2, b2, y);
}

extern "C" int sayuri_hip_test_head_board(int device, int n, const int* board_sizes, int max_board, int channels, int policy_channels,
                                          int value_channels, int prob_channels, int pass_outs, int misc_outs, int act, const float* trunk,
                                          const float* p_w, const float* p_b, const float* v_w, const float* v_b,
                                          const float* const* weights12, float* prob, float* pass, float* misc, float* own) {
    if (!board_sizes || !trunk || !p_w || !p_b || !v_w || !v_b || !weights12 || !prob || !pass || !misc || !own || n <= 0)
        return fail("test_head_board: bad argument");
    return test_head_board_impl(device, n, board_sizes, max_board, channels, policy_channels, value_channels, prob_channels, pass_outs,
                                misc_outs, act, trunk, p_w, p_b, v_w, v_b, weights12, prob, pass, misc, own);
}

extern "C" int sayuri_hip_test_se_unit(int device, int use_fp16, int n, const int* board_sizes, int max_board, int channels, int se_size,
                                       int act, const float* x, const float* res, const float* w1, const float* b1, const float* w2,
                                       const float* b2, float* y, float* gate) {
    if (!board_sizes || !x || !w1 || !b1 || !w2 || !b2 || !y || n <= 0) return fail("test_se_unit: bad argument");
    if (use_fp16) return test_se_unit_impl<f16>(device, n, board_sizes, max_board, channels, se_size, act, x, res, w1, b1, w2, b2, y, gate);
    return test_se_unit_impl<float>(device, n, board_sizes, max_board, channels, se_size, act, x, res, w1, b1, w2, b2, y, gate);
}

extern "C" int sayuri_hip_test_head_tail(int device, int use_fp16, int n, const int* board_sizes, int max_board, int policy_channels,
                                         int value_channels, int prob_channels, int pass_outs, int misc_outs, int act, const float* pconv,
                                         const float* vconv, const float* const* weights12, float* prob, float* pass, float* misc,
                                         float* own) {
    if (!board_sizes || !pconv || !vconv || !weights12 || !prob || !pass || !misc || !own || n <= 0) return fail("test_head_tail: bad argument");
    if (use_fp16)
        return test_head_tail_impl<f16>(device, n, board_sizes, max_board, policy_channels, value_channels, prob_channels, pass_outs,
                                        misc_outs, act, pconv, vconv, weights12, prob, pass, misc, own);
    return test_head_tail_impl<float>(device, n, board_sizes, max_board, policy_channels, value_channels, prob_channels, pass_outs,
                                      misc_outs, act, pconv, vconv, weights12, prob, pass, misc, own);
}

extern "C" int sayuri_hip_test_last_conv_kind(void) { return sayuri::g_test_conv_kind; }

extern "C" int sayuri_hip_test_conv(int device, int use_fp16, int n, const int* board_sizes, int max_board, int cin,
                                    int cout, int k, int depthwise, int act, int post_residual, const float* x,
                                    const float* w, const float* bias, const float* res, float* y) {
    if (!board_sizes || !x || !w || !y || n <= 0) return fail("test_conv: bad argument");
    if (use_fp16)
        return test_conv_impl<f16>(device, n, board_sizes, max_board, cin, cout, k, depthwise, act, post_residual, x, w,
                                   bias, res, y);
    return test_conv_impl<float>(device, n, board_sizes, max_board, cin, cout, k, depthwise, act, post_residual, x, w,
                                 bias, res, y);
}
