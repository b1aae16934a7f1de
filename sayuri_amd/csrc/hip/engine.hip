// engine.hip -- the C-ABI of include/sayuri_hip.h: one translation unit in three parts
//   engine_plan.h   kernel registries, environment switches, batch geometry and tile plans, device images of the layers
//   engine_graph.h  Engine<T>: the per-GPU forward graph -- counterpart of the reference's CudaForwardPipe::NNGraph
//                   (src/neural/cuda/cuda_forward_pipe.cc:133-1090) and its layer objects (src/neural/cuda/cuda_layers.cc),
//                   re-designed for MI355X: compact NHWC activations (common.h), one workgroup per board with the whole residual
//                   tower as ONE persistent launch (conv_board.h, conv_tower.h), SE units inside the convolutions (conv_board.h,
//                   conv_board_sx.h), both heads in one kernel (head_board.h), mixed board sizes without masked work
//   engine_taps.h   layer-level test taps (sayuri_hip_test_*)
// and, in this file, the entry points themselves.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../../include/sayuri_hip.h"
#include "common.h"
#include "conv_mfma.h"
#include "conv_glds.h"
#include "conv_board.h"
#include "conv_board_sx.h"
#include "conv_tower.h"
#include "head_board.h"
#include "small_ops.h"

#include "engine_plan.h"
#include "engine_graph.h"

// ====================================================================== C ABI
using namespace sayuri;

struct sayuri_hip_ctx {
    std::unique_ptr<EngineBase> eng;
};

extern "C" {

const char* sayuri_hip_last_error(void) { return g_err.c_str(); }

int sayuri_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

sayuri_hip_ctx* sayuri_hip_create(int device, const sayuri_hip_netdesc* desc, int max_batch, int board, int use_fp16) {
    if (!desc || !desc->blocks || max_batch <= 0 || board < 2 || board > 25) {
        fail("sayuri_hip_create: bad arguments");
        return nullptr;
    }
    int ndev = sayuri_hip_device_count();
    if (device < 0 || device >= ndev) {
        fail("sayuri_hip_create: no such HIP device (found " + std::to_string(ndev) + ")");
        return nullptr;
    }
    auto ctx = std::make_unique<sayuri_hip_ctx>();
    int rc;
    const EngineFlags flags = EngineFlags::from_env();
    if (use_fp16) {
        auto e = std::make_unique<Engine<f16>>(device, *desc, max_batch, board, flags);
        rc = e->init();
        ctx->eng = std::move(e);
    } else {
        auto e = std::make_unique<Engine<float>>(device, *desc, max_batch, board, flags);
        rc = e->init();
        ctx->eng = std::move(e);
    }
    if (rc) return nullptr;
    return ctx.release();
}

int sayuri_hip_load_tensor(sayuri_hip_ctx* ctx, int layer_id, int kind, const float* host, size_t n) {
    if (!ctx || !host) return fail("load_tensor: null argument");
    return ctx->eng->load_tensor(layer_id, kind, host, n);
}

int sayuri_hip_upload(sayuri_hip_ctx* ctx, int n, const float* planes, const int* board_sizes) {
    if (!ctx || !planes) return fail("upload: null argument");
    return ctx->eng->upload(n, planes, board_sizes);
}
int sayuri_hip_run(sayuri_hip_ctx* ctx) { return ctx ? ctx->eng->run() : fail("run: null ctx"); }
int sayuri_hip_sync(sayuri_hip_ctx* ctx) { return ctx ? ctx->eng->sync() : fail("sync: null ctx"); }
int sayuri_hip_download(sayuri_hip_ctx* ctx, float* prob, float* pass, float* misc, float* own) {
    return ctx ? ctx->eng->download(prob, pass, misc, own) : fail("download: null ctx");
}

int sayuri_hip_forward(sayuri_hip_ctx* ctx, int n, const float* planes, const int* board_sizes, float* prob,
                       float* pass, float* misc, float* own) {
    if (!ctx) return fail("forward: null ctx");
    if (ctx->eng->upload(n, planes, board_sizes)) return -1;
    if (ctx->eng->run()) return -1;
    return ctx->eng->download(prob, pass, misc, own);
}

int sayuri_hip_forward_packed(sayuri_hip_ctx* ctx, int n, const unsigned* records, int binary_planes, const int* board_sizes,
                              float* prob, float* pass, float* misc, float* own) {
    if (!ctx || !records) return fail("forward_packed: null argument");
    if (ctx->eng->upload(n, nullptr, board_sizes, records, binary_planes)) return -1;
    if (ctx->eng->run()) return -1;
    return ctx->eng->download(prob, pass, misc, own);
}
int sayuri_hip_submit_packed(sayuri_hip_ctx* ctx, int n, const unsigned* records, int binary_planes, const int* board_sizes,
                             float* prob, float* pass, float* misc, float* own, int* ticket) {
    if (!ctx || !records || !ticket) return fail("submit_packed: null argument");
    return ctx->eng->submit(n, nullptr, board_sizes, prob, pass, misc, own, ticket, records, binary_planes);
}
int sayuri_hip_submit(sayuri_hip_ctx* ctx, int n, const float* planes, const int* board_sizes, float* prob,
                      float* pass, float* misc, float* own, int* ticket) {
    if (!ctx || !planes || !prob || !pass || !misc || !own || !ticket) return fail("submit: null argument");
    return ctx->eng->submit(n, planes, board_sizes, prob, pass, misc, own, ticket);
}
int sayuri_hip_wait(sayuri_hip_ctx* ctx, int ticket) { return ctx ? ctx->eng->wait(ticket) : fail("wait: null ctx"); }
int sayuri_hip_query(sayuri_hip_ctx* ctx, int ticket) { return ctx ? ctx->eng->query(ticket) : fail("query: null ctx"); }

int sayuri_hip_time_runs(sayuri_hip_ctx* ctx, int iters, float* total_ms) {
    if (!ctx || !total_ms || iters <= 0) return fail("time_runs: bad argument");
    return ctx->eng->time_runs(iters, total_ms);
}

int sayuri_hip_mark_kernel(sayuri_hip_ctx* ctx, const char* name) {
    return ctx ? ctx->eng->mark_kernel(name) : fail("mark_kernel: null ctx");
}
int sayuri_hip_timed_stat(sayuri_hip_ctx* ctx, sayuri_hip_kernel_stat* row) {
    if (!ctx || !row) return fail("timed_stat: bad argument");
    return ctx->eng->timed_stat(row);
}

int sayuri_hip_profile_run(sayuri_hip_ctx* ctx, sayuri_hip_kernel_stat* rows, int cap) {
    if (!ctx || !rows) return fail("profile_run: bad argument");
    return ctx->eng->profile_run(rows, cap);
}

void* sayuri_hip_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
        fail("hipHostMalloc failed");
        return nullptr;
    }
    return p;
}
void sayuri_hip_host_free(void* p) {
    if (!p) return;
    g_host_free_gen.fetch_add(1, std::memory_order_release);
    (void)hipHostFree(p);
}

size_t sayuri_hip_device_bytes(const sayuri_hip_ctx* ctx) { return ctx ? ctx->eng->device_bytes() : 0; }
int sayuri_hip_last_chains(const sayuri_hip_ctx* ctx) { return ctx ? ctx->eng->last_chains() : 0; }
int sayuri_hip_tower_state(const sayuri_hip_ctx* ctx) { return ctx ? ctx->eng->tower_state() : -1; }
// debugging tap (not part of the ABI, not in the header): activation buffer `buf` (0..5) of ticket 0 as it stands after the last forward
extern "C" int sayuri_hip_debug_read_activations(sayuri_hip_ctx* ctx, int buf, void* host, size_t bytes) {
    return ctx ? ctx->eng->debug_read(buf, host, bytes) : -1;
}

void sayuri_hip_destroy(sayuri_hip_ctx* ctx) { delete ctx; }

}

#include "engine_taps.h"
