// The non-GEMM kernels of the forward pipe (all HBM/latency-bound, fp32 math):
//   pack_input_kernel  planes [n][C][B*B] (NN grid, fp32) -> compact NHWC activations
//   se_gate_kernel     global pooling + squeeze FC + excite FC of one SE unit, one WG per sample
//                      (reference se_unit.cc:9-128 / cuda_kernels.cu:241-321 + two cuBLAS gemms)
//   se_scale_kernel    act(sigmoid(gamma)*x + beta + residual)   (cuda_kernels.cu:391-440)
//   depthwise_kernel   k x k depthwise conv, bias, act, optional post-activation residual
//                      (cuda_kernels.cu:708-770)
//   head_tail_kernel   everything after the two 1x1 head convolutions, one WG per sample:
//                      policy pooling -> inter FC -> pass FC, value pooling -> inter FC -> misc FC,
//                      per-pixel policy planes (with the inter vector added as a channel bias)
//                      and ownership, written straight into the NN-grid fp32 output tensors
//                      (cuda_forward_pipe.cc:930-981)
#pragma once
#include "common.h"

namespace sayuri {

template <typename T>
__global__ void pack_input_kernel(const float* __restrict__ planes, T* __restrict__ out, BatchGeom g,
                                  int cin, int cs, int board) {
    const int gi = blockIdx.x * blockDim.x + threadIdx.x;
    if (gi >= g.total_pix) return;
    int lo = 0, hi = g.n_samples;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (g.sample_off[mid] <= gi) lo = mid; else hi = mid;
    }
    const int n = lo, bs = g.bsz[n], pp = gi - g.sample_off[n];
    const int y = pp / bs, x = pp - y * bs;
    const float* src = planes + (size_t)n * cin * board * board + y * board + x;
    T* dst = out + ((size_t)n * g.slot_pix + pp) * cs;
    for (int c = 0; c < cs; ++c) dst[c] = from_float<T>(c < cin ? src[(size_t)c * board * board] : 0.f);
}

struct FcDev {
    const float* wt;  // transposed [in][out]
    const float* b;   // [out]
    int in, out;
};

// y[o] = act(b[o] + sum_i x[i] * wt[i][o]); x and y in LDS; all threads of the block cooperate.
__device__ __forceinline__ void block_fc(const FcDev fc, const float* x, float* y, int act, int tid, int nt) {
    for (int o = tid; o < fc.out; o += nt) {
        float s = 0.f;
        for (int i = 0; i < fc.in; ++i) s += x[i] * fc.wt[(size_t)i * fc.out + o];
        y[o] = activate(fc.b[o] + s, act);
    }
}

// Per-channel (mean, scaled mean, max | value-head third moment) over the sample's pixels into
// LDS `pool[3*C]`, layout [mean...][scaled...][max...] (se_unit.cc:9-68).  blockDim.x threads.
template <typename T>
__device__ __forceinline__ void block_pool(const T* __restrict__ x, int npix, int C, int cs, int bs,
                                           bool value_head, float* pool, float* scratch, int tid, int nt) {
    // thread -> (channel lane cl, pixel part): consecutive threads read consecutive channels
    const int cl_n = cs < nt ? cs : nt;
    const int parts = nt / cl_n;
    const int cl = tid % cl_n, part = tid / cl_n;
    for (int c0 = 0; c0 < cs; c0 += cl_n) {
        const int c = c0 + cl;
        float sum = 0.f, mx = -5000.f;
        if (part < parts && c < cs) {
            for (int p = part; p < npix; p += parts) {
                const float v = to_float(x[(size_t)p * cs + c]);
                sum += v;
                mx = fmaxf(mx, v);
            }
        }
        scratch[tid] = sum;
        scratch[nt + tid] = mx;
        __syncthreads();
        if (part == 0 && c < C) {
            for (int q = 1; q < parts; ++q) {
                sum += scratch[q * cl_n + cl];
                mx = fmaxf(mx, scratch[nt + q * cl_n + cl]);
            }
            const float mean = sum / (float)npix;
            const float bd = (float)bs - 14.f;
            pool[c] = mean;
            pool[C + c] = mean * (bd / 10.f);
            pool[2 * C + c] = value_head ? mean * (bd * bd / 100.f - 0.1f) : mx;
        }
        __syncthreads();
    }
}

// One SE unit's gate: pooling + squeeze + excite.  gate[n][0..C) = sigmoid(gamma), [C..2C) = beta.
template <typename T>
__global__ __launch_bounds__(256) void se_gate_kernel(const T* __restrict__ x, float* __restrict__ gate,
                                                      BatchGeom g, int C, int cs, FcDev squeeze, FcDev excite,
                                                      int act) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* pool = (float*)smem;           // [3C]
    float* mid = pool + 3 * C;            // [se_size]
    float* scratch = mid + squeeze.out;   // [2*256]
    const int n = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int bs = g.bsz[n];
    block_pool<T>(x + (size_t)n * g.slot_pix * cs, bs * bs, C, cs, bs, false, pool, scratch, tid, nt);
    block_fc(squeeze, pool, mid, act, tid, nt);
    __syncthreads();
    for (int o = tid; o < excite.out; o += nt) {
        float s = 0.f;
        for (int i = 0; i < excite.in; ++i) s += mid[i] * excite.wt[(size_t)i * excite.out + o];
        s += excite.b[o];
        gate[(size_t)n * 2 * C + o] = o < C ? 1.0f / (1.0f + fast_exp(-s)) : s;
    }
}

// out = act(gate_gamma[n][c] * x + beta[n][c] + res), 8 channels (fp16) / 4 (fp32) per thread.
template <typename T>
__global__ void se_scale_kernel(const T* __restrict__ x, const T* __restrict__ res, T* __restrict__ out,
                                const float* __restrict__ gate, BatchGeom g, int C, int cs, int act) {
    constexpr int EPP = ElemTraits<T>::kPieceElems;
    const int ppp = cs / EPP;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)g.total_pix * ppp;
    if (idx >= total) return;
    const int gi = (int)(idx / ppp), piece = (int)(idx - (size_t)gi * ppp);
    int lo = 0, hi = g.n_samples;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (g.sample_off[mid] <= gi) lo = mid; else hi = mid;
    }
    const int n = lo, pp = gi - g.sample_off[n];
    const size_t o = ((size_t)n * g.slot_pix + pp) * cs + piece * EPP;
    const float* gm = gate + (size_t)n * 2 * C;
    T vx[EPP], vr[EPP], vo[EPP];
    *(uint4*)vx = *(const uint4*)(x + o);
    if (res) *(uint4*)vr = *(const uint4*)(res + o);
#pragma unroll
    for (int e = 0; e < EPP; ++e) {
        const int c = piece * EPP + e;
        float v = 0.f;
        if (c < C) {
            v = gm[c] * to_float(vx[e]) + gm[C + c];
            if (res) v += to_float(vr[e]);
            v = activate(v, act);
        }
        vo[e] = from_float<T>(v);
    }
    *(uint4*)(out + o) = *(uint4*)vo;
}

// Depthwise k x k convolution.  wt = [k*k][cs] fp32 (transposed), bias [cs].
// post_res: out = act(conv + b) + res (AddSpatialBiasesPost, biases.cc:49-77); else act(conv + b).
template <typename T>
__global__ void depthwise_kernel(const T* __restrict__ x, const T* __restrict__ res, T* __restrict__ out,
                                 const float* __restrict__ wt, const float* __restrict__ bias, BatchGeom g,
                                 int C, int cs, int k, int act) {
    constexpr int EPP = ElemTraits<T>::kPieceElems;
    const int ppp = cs / EPP;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)g.total_pix * ppp;
    if (idx >= total) return;
    const int gi = (int)(idx / ppp), piece = (int)(idx - (size_t)gi * ppp);
    int lo = 0, hi = g.n_samples;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (g.sample_off[mid] <= gi) lo = mid; else hi = mid;
    }
    const int n = lo, bs = g.bsz[n], pp = gi - g.sample_off[n];
    const int y = pp / bs, xx = pp - y * bs, pad = k / 2;
    const T* xs = x + (size_t)n * g.slot_pix * cs + piece * EPP;
    float acc[EPP];
#pragma unroll
    for (int e = 0; e < EPP; ++e) acc[e] = 0.f;
    for (int kr = 0; kr < k; ++kr) {
        const int iy = y + kr - pad;
        if (iy < 0 || iy >= bs) continue;
        for (int kc = 0; kc < k; ++kc) {
            const int ix = xx + kc - pad;
            if (ix < 0 || ix >= bs) continue;
            T v[EPP];
            *(uint4*)v = *(const uint4*)(xs + (size_t)(iy * bs + ix) * cs);
            const float* w = wt + (size_t)(kr * k + kc) * cs + piece * EPP;
#pragma unroll
            for (int e = 0; e < EPP; ++e) acc[e] += to_float(v[e]) * w[e];
        }
    }
    const size_t o = ((size_t)n * g.slot_pix + pp) * cs + piece * EPP;
    T vr[EPP], vo[EPP];
    if (res) *(uint4*)vr = *(const uint4*)(res + o);
#pragma unroll
    for (int e = 0; e < EPP; ++e) {
        const int c = piece * EPP + e;
        float v = 0.f;
        if (c < C) {
            v = activate(acc[e] + bias[c], act);
            if (res) v += to_float(vr[e]);
        }
        vo[e] = from_float<T>(v);
    }
    *(uint4*)(out + o) = *(uint4*)vo;
}

struct HeadParams {
    FcDev p_inter, pass_fc, v_inter, v_misc;
    const float* prob_w;  // [prob_ch][Cp]
    const float* prob_b;  // [prob_ch]
    const float* own_w;   // [Cv]  (ownership_channels == 1)
    const float* own_b;   // [1]
    int Cp, cs_p, Cv, cs_v, prob_ch, act, board;
    float* prob;  // [n][prob_ch][board*board]
    float* pass;  // [n][pass_outs]
    float* misc;  // [n][misc_outs]
    float* own;   // [n][board*board]
};

template <typename T>
__global__ __launch_bounds__(256) void head_tail_kernel(const T* __restrict__ pconv, const T* __restrict__ vconv,
                                                        BatchGeom g, HeadParams h) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int maxc = h.Cp > h.Cv ? h.Cp : h.Cv;
    float* pool = (float*)smem;          // [3*maxc]
    float* inter = pool + 3 * maxc;      // [3*maxc]
    float* pinter = inter + 3 * maxc;    // [Cp] policy intermediate (kept for the per-pixel pass)
    float* scratch = pinter + maxc;      // [2*256]
    const int n = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int bs = g.bsz[n], npix = bs * bs, B2 = h.board * h.board;
    const T* pc = pconv + (size_t)n * g.slot_pix * h.cs_p;
    const T* vc = vconv + (size_t)n * g.slot_pix * h.cs_v;

    // policy: pooling -> intermediate FC (act) -> pass FC
    block_pool<T>(pc, npix, h.Cp, h.cs_p, bs, false, pool, scratch, tid, nt);
    block_fc(h.p_inter, pool, pinter, h.act, tid, nt);
    __syncthreads();
    block_fc(h.pass_fc, pinter, h.pass + (size_t)n * h.pass_fc.out, kIdentity, tid, nt);

    // value: pooling -> intermediate FC (act) -> misc FC
    block_pool<T>(vc, npix, h.Cv, h.cs_v, bs, true, pool, scratch, tid, nt);
    block_fc(h.v_inter, pool, inter, h.act, tid, nt);
    __syncthreads();
    block_fc(h.v_misc, inter, h.misc + (size_t)n * h.v_misc.out, kIdentity, tid, nt);

    // per-pixel outputs in the NN grid (off-board cells of a smaller sample = 0)
    for (int cell = tid; cell < B2; cell += nt) {
        const int y = cell / h.board, x = cell - y * h.board;
        const bool on = y < bs && x < bs;
        const int pp = y * bs + x;
        float pr[8];
        for (int k = 0; k < h.prob_ch; ++k) pr[k] = on ? h.prob_b[k] : 0.f;
        float ow = on ? h.own_b[0] : 0.f;
        if (on) {
            for (int c = 0; c < h.Cp; ++c) {
                const float v = to_float(pc[(size_t)pp * h.cs_p + c]) + pinter[c];
                for (int k = 0; k < h.prob_ch; ++k) pr[k] += v * h.prob_w[k * h.Cp + c];
            }
            for (int c = 0; c < h.Cv; ++c) ow += to_float(vc[(size_t)pp * h.cs_v + c]) * h.own_w[c];
        }
        for (int k = 0; k < h.prob_ch; ++k) h.prob[((size_t)n * h.prob_ch + k) * B2 + cell] = pr[k];
        h.own[(size_t)n * B2 + cell] = ow;
    }
}

}  // namespace sayuri
