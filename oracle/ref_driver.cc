// oracle/ref_driver.cc -- TEST INFRASTRUCTURE ONLY (never part of the product path).
//
// A thin extern "C" tap on the *unmodified* reference sources, compiled in THIS
// container only (where /root/reference is mounted) by oracle/Makefile into
// oracle/_ref/libsayuri_ref.so.  It lets tests and the fixture generator call the
// reference's own CPU pipe and weight loader on seeded inputs:
//
//   * DNNLoader::FromFile              (reference src/neural/loader.cc:26-65)
//   * BlasForwardPipe::Forward         (reference src/neural/blas/blas_forward_pipe.cc:314-563)
//   * DNNWeights tensors after folding (reference src/neural/loader.cc:775-831)
//
// No reference code is copied here: this file only *calls* reference classes through
// their public headers.  The built .so is git-ignored; it travels to the GPU box only
// as a prebuilt checker / "reference" CPU baseline.
#include <cstring>
#include <memory>
#include <thread>
#include <string>
#include <vector>

#include "config.h"
#include "neural/blas/blas_forward_pipe.h"
#include "neural/description.h"
#include "neural/loader.h"
#include "neural/network_basic.h"
#include "utils/option.h"

#ifdef WITH_HIP_PIPE
// The product pipe compiled IN the reference tree (-DSAYURI_IN_TREE: reference headers, reference
// types): the drop-in configuration of INTEGRATION.md, used by tests/test_gpu_dropin.py.
#include "hip_forward_pipe.h"
#endif

namespace {
std::shared_ptr<DNNWeights> g_weights;
std::unique_ptr<BlasForwardPipe> g_pipe;
bool g_args_ready = false;
std::string g_err;

ConvLayer* FindConv(BlockBasic* b, const std::string& n) {
    if (n == "conv1") return &b->conv1;
    if (n == "conv2") return &b->conv2;
    if (n == "conv3") return &b->conv3;
    if (n == "conv4") return &b->conv4;
    if (n == "pre_btl_conv") return &b->pre_btl_conv;
    if (n == "post_btl_conv") return &b->post_btl_conv;
    if (n == "dw_conv") return &b->dw_conv;
    return nullptr;
}
LinearLayer* FindFc(BlockBasic* b, const std::string& n) {
    if (n == "squeeze") return &b->squeeze;
    if (n == "excite") return &b->excite;
    return nullptr;
}
ConvLayer* FindTopConv(DNNWeights* w, const std::string& n) {
    if (n == "input_conv") return &w->input_conv;
    if (n == "p_hd_conv") return &w->p_hd_conv;
    if (n == "p_dw_conv") return &w->p_dw_conv;
    if (n == "p_pt_conv") return &w->p_pt_conv;
    if (n == "prob_conv") return &w->prob_conv;
    if (n == "v_hd_conv") return &w->v_hd_conv;
    if (n == "v_ownership") return &w->v_ownership;
    return nullptr;
}
LinearLayer* FindTopFc(DNNWeights* w, const std::string& n) {
    if (n == "p_inter_fc") return &w->p_inter_fc;
    if (n == "pass_fc") return &w->pass_fc;
    if (n == "v_inter_fc") return &w->v_inter_fc;
    if (n == "v_misc") return &w->v_misc;
    return nullptr;
}
} // namespace

extern "C" {

const char* ref_last_error() { return g_err.c_str(); }

// Load a weights file with the reference loader.  winograd=0 => "--no-winograd".
// Returns 0 on success, -1 when the reference refused the file.
// Build the reference's option map / tables once (ArgsParser, config.cc:336-381).
int ref_ensure_args(int winograd) {
    if (!g_args_ready) {
        std::vector<std::string> args = {"sayuri", "--quiet", "-t", "1", "-p", "1"};
        if (!winograd) args.emplace_back("--no-winograd");
        std::vector<char*> argv;
        for (auto& a : args) argv.push_back(a.data());
        ArgsParser(static_cast<int>(argv.size()), argv.data());
        g_args_ready = true;
    }
    return 0;
}

int ref_init(const char* weights_path, int winograd) {
    try {
        ref_ensure_args(winograd);
        SetOption("winograd", static_cast<bool>(winograd));
        g_weights = std::make_shared<DNNWeights>();
        DNNLoader::Get().FromFile(g_weights, weights_path);
        if (!g_weights->loaded) {
            g_err = "reference loader rejected the weights file";
            g_weights.reset();
            return -1;
        }
        g_pipe = std::make_unique<BlasForwardPipe>();
        g_pipe->Initialize(g_weights);
        return 0;
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    }
}

// info[0..11] = version, input_channels, residual_blocks, residual_channels,
// policy_head_channels, value_head_channels, probabilities_channels,
// pass_probability_outputs, ownership_channels, value_misc_outputs,
// default_act, policy_head_type(0 normal / 1 RepLK)
int ref_info(int* info) {
    if (!g_weights) return -1;
    auto* w = g_weights.get();
    info[0] = w->version;
    info[1] = w->input_channels;
    info[2] = w->residual_blocks;
    info[3] = w->residual_channels;
    info[4] = w->policy_head_channels;
    info[5] = w->value_head_channels;
    info[6] = w->probabilities_channels;
    info[7] = w->pass_probability_outputs;
    info[8] = w->ownership_channels;
    info[9] = w->value_misc_outputs;
    info[10] = static_cast<int>(w->default_act);
    info[11] = w->policy_head_type == PolicyHeadType::kRepLK ? 1 : 0;
    return 0;
}

// binfo[0..4] = type(1 res,2 btl,3 nested,4 mixer), apply_se, se_size,
// bottleneck_channels, feedforward_channels
int ref_block_info(int idx, int* binfo) {
    if (!g_weights || idx < 0 || idx >= g_weights->residual_blocks) return -1;
    auto* b = g_weights->tower[idx].get();
    binfo[0] = static_cast<int>(b->type);
    binfo[1] = b->apply_se ? 1 : 0;
    binfo[2] = b->se_size;
    binfo[3] = b->bottleneck_channels;
    binfo[4] = b->feedforward_channels;
    return 0;
}

// Fetch a post-ProcessWeights tensor by name: "<layer>.<w|b|u>" for top-level
// layers, "tower.<i>.<layer>.<w|b|u>" for block layers ("u" = Winograd U of a
// 3x3 conv).  Returns the element count (copying min(count, cap) floats), -1 if
// unknown.
long ref_get_tensor(const char* name_c, float* dst, long cap) {
    if (!g_weights) return -1;
    std::string name = name_c;
    std::vector<float>* v = nullptr;
    auto pick_conv = [&](ConvLayer* c, const std::string& kind) -> std::vector<float>* {
        if (!c) return nullptr;
        if (kind == "w") return &c->GetWeights();
        if (kind == "b") return &c->GetBiases();
        if (kind == "u") return &c->GetTransformF();
        return nullptr;
    };
    auto pick_fc = [&](LinearLayer* c, const std::string& kind) -> std::vector<float>* {
        if (!c) return nullptr;
        if (kind == "w") return &c->GetWeights();
        if (kind == "b") return &c->GetBiases();
        return nullptr;
    };
    const auto last_dot = name.rfind('.');
    if (last_dot == std::string::npos) return -1;
    const std::string kind = name.substr(last_dot + 1);
    std::string path = name.substr(0, last_dot);
    if (path.rfind("tower.", 0) == 0) {
        const auto d2 = path.find('.', 6);
        if (d2 == std::string::npos) return -1;
        const int idx = std::stoi(path.substr(6, d2 - 6));
        if (idx < 0 || idx >= g_weights->residual_blocks) return -1;
        const std::string lname = path.substr(d2 + 1);
        auto* b = g_weights->tower[idx].get();
        v = pick_conv(FindConv(b, lname), kind);
        if (!v) v = pick_fc(FindFc(b, lname), kind);
    } else {
        v = pick_conv(FindTopConv(g_weights.get(), path), kind);
        if (!v) v = pick_fc(FindTopFc(g_weights.get(), path), kind);
    }
    if (!v) return -1;
    const long n = static_cast<long>(v->size());
    if (dst) std::memcpy(dst, v->data(), sizeof(float) * static_cast<size_t>(std::min(n, cap)));
    return n;
}

// One evaluation through BlasForwardPipe::Forward.  planes = [C_in][bs*bs] packed
// with the sample's own board stride (InputData layout, network_basic.h:23-34).
// out (raw, pre-activation, 2*bs*bs + 9 floats):
//   prob[bs*bs], own[bs*bs], pass, wdl[3], stm_winrate, final_score, q_error,
//   score_error, (float)offset
int ref_forward(int board_size, float komi, int side_to_move, int offset, const float* planes,
                float* out) {
    if (!g_pipe) return -1;
    try {
        auto inp = std::make_unique<InputData>();
        inp->board_size = board_size;
        inp->komi = komi;
        inp->side_to_move = side_to_move;
        inp->offset = static_cast<PolicyBufferOffset>(offset);
        const int n = g_weights->input_channels * board_size * board_size;
        std::memcpy(inp->planes.data(), planes, sizeof(float) * n);
        const OutputResult r = g_pipe->Forward(*inp);
        const int s = board_size * board_size;
        std::memcpy(out, r.probabilities.data(), sizeof(float) * s);
        std::memcpy(out + s, r.ownership.data(), sizeof(float) * s);
        float* t = out + 2 * s;
        t[0] = r.pass_probability;
        t[1] = r.wdl[0];
        t[2] = r.wdl[1];
        t[3] = r.wdl[2];
        t[4] = r.stm_winrate;
        t[5] = r.final_score;
        t[6] = r.q_error;
        t[7] = r.score_error;
        t[8] = static_cast<float>(static_cast<int>(r.offset));
        return 0;
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    }
}

#ifdef WITH_HIP_PIPE
namespace {
std::unique_ptr<HipForwardPipe> g_hip;
}

// Build HipForwardPipe on the weights the REFERENCE loader parsed (ref_init must have run).
int ref_hip_init(int board, int batch, int fp16, int device) {
    if (!g_weights) { g_err = "ref_init first"; return -1; }
    try {
        HipPipeConfig cfg;
        cfg.batch_size = batch;
        cfg.fp16 = fp16 != 0;
        cfg.default_boardsize = board;
        if (device >= 0) cfg.gpus = {device};
        g_hip = std::make_unique<HipForwardPipe>(cfg);
        g_hip->Initialize(g_weights);
        return 0;
    } catch (const std::exception& e) {
        g_err = e.what();
        g_hip.reset();
        return -1;
    }
}

// n concurrent NetworkForwardPipe::Forward calls (reference InputData / OutputResult) through
// the MI355X pipe; same packing as ref_forward, stride 2*361+9 per sample.
int ref_hip_forward(int n, const int* board_sizes, const int* offsets, const float* planes, float* out) {
    if (!g_hip) { g_err = "ref_hip_init first"; return -1; }
    try {
        std::vector<InputData> in(n);
        std::vector<OutputResult> res(n);
        const int PL = kInputChannels * kNumIntersections, OL = 2 * kNumIntersections + 9;
        for (int i = 0; i < n; ++i) {
            in[i].board_size = board_sizes[i];
            in[i].komi = 7.5f;
            in[i].offset = static_cast<PolicyBufferOffset>(offsets[i]);
            std::memcpy(in[i].planes.data(), planes + static_cast<size_t>(i) * PL, sizeof(float) * PL);
        }
        std::vector<std::thread> th;
        std::vector<std::string> errs(n);
        for (int i = 0; i < n; ++i)
            th.emplace_back([&, i] {
                try {
                    res[i] = g_hip->Forward(in[i]);
                } catch (const std::exception& e) {
                    errs[i] = e.what();
                }
            });
        for (auto& t : th) t.join();
        for (auto& e : errs)
            if (!e.empty()) throw std::runtime_error(e);
        for (int i = 0; i < n; ++i) {
            float* o = out + static_cast<size_t>(i) * OL;
            const OutputResult& r = res[i];
            const int s = board_sizes[i] * board_sizes[i];
            std::memset(o, 0, sizeof(float) * OL);
            std::memcpy(o, r.probabilities.data(), sizeof(float) * s);
            std::memcpy(o + kNumIntersections, r.ownership.data(), sizeof(float) * s);
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

void ref_hip_destroy() { g_hip.reset(); }
#endif

} // extern "C"
