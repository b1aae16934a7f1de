// The non-GEMM kernels of the forward pipe (all HBM/latency-bound, fp32 math):
//   pack_input_kernel  planes [n][C][B*B] (NN grid, fp32) -> compact NHWC activations
//   se_pool / se_fc    global pooling, then squeeze FC + excite FC of one SE unit
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
#include <algorithm>

#include "common.h"

namespace sayuri {

// kPackSplit workgroups per sample (grid = n_samples * kPackSplit, each a range of the sample's pixels).  The planes are read along pixels (thread = pixel, eight planes
// per 16-byte piece: every load of a wave is 256 contiguous bytes), transposed through LDS ([pixel][cs] with a 16-byte pad per
// row: 16-byte writes of consecutive pixels then fall on distinct banks) and leave as whole NHWC rows (consecutive lanes
// write consecutive 16-byte pieces).  15.9 MB in + 11.8 MB out for a 256-batch: a streaming kernel.
constexpr int kPackThreads = 128;
constexpr int kPackSplit = 4;  // workgroups per sample (pixel ranges): several per CU, so that one's loads overlap another's stores
template <typename T>
__global__ __launch_bounds__(kPackThreads) void pack_input_kernel(const float* __restrict__ planes, T* __restrict__ out,
                                                                  BatchGeom g, int cin, int cs, int board, const int* __restrict__ perm,
                                                                  int chunk, int n0 = 0) {  // n0: first sample of the launch
    constexpr int EPP = ElemTraits<T>::kPieceElems;
    extern __shared__ __attribute__((aligned(16))) unsigned char pk_smem[];
    const int n = n0 + blockIdx.x / kPackSplit, part = blockIdx.x % kPackSplit, tid = threadIdx.x;
    const int bs = g.bsz[n], npix = bs * bs, ppr = cs / EPP, stride = cs * (int)sizeof(T) + 16;
    const int per = (npix + kPackSplit - 1) / kPackSplit, pbeg = part * per, pend = min(npix, pbeg + per);
    const size_t B2 = (size_t)board * board;
    const float* src = planes + (size_t)(perm ? perm[n] : n) * cin * B2;  // perm: device sample -> caller's slot
    T* dst = out + (size_t)n * g.slot_pix * cs;
    for (int p0 = pbeg; p0 < pend; p0 += chunk) {  // `chunk` pixels fit the LDS
        const int np = min(chunk, pend - p0);
        if (p0 != pbeg) __syncthreads();
        for (int q = tid; q < np; q += kPackThreads) {
            const int pp = p0 + q, y = pp / bs, x = pp - y * bs;
            const float* s0 = src + y * board + x;
            // six pieces (48 planes) at a time: all their loads go out before the first conversion (a thread has its
            // whole pixel in flight at once -- with one workgroup per CU this kernel lives on memory-level parallelism)
            for (int pb = 0; pb < ppr; pb += 6) {
                float f[6 * EPP];
#pragma unroll
                for (int k = 0; k < 6 * EPP; ++k) {
                    const int c = pb * EPP + k;
                    f[k] = c < cin ? s0[(size_t)c * B2] : 0.f;
                }
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    if (pb + j < ppr) {
                        T v[EPP];
#pragma unroll
                        for (int e = 0; e < EPP; ++e) v[e] = from_float<T>(f[j * EPP + e]);
                        *(uint4*)(pk_smem + (size_t)q * stride + (pb + j) * 16) = *(uint4*)v;
                    }
                }
            }
        }
        __syncthreads();
        for (int it = tid; it < np * ppr; it += kPackThreads) {
            const int q = it / ppr, piece = it - q * ppr;
            *(uint4*)(dst + ((size_t)p0 * ppr + it) * EPP) = *(const uint4*)(pk_smem + (size_t)q * stride + piece * 16);
        }
    }
}
// The same when a whole sample fits the LDS (fp16 activations of a 19x19 board: 361 rows of 144 bytes): ONE workgroup of 512
// threads per sample.  A sample's planes are one contiguous run of cin * board^2 floats: thread t takes elements t, t + 512, ...
// (every load of a wave is 256 contiguous bytes, up to kPackFlatLoads of them in flight per thread before the first is used),
// drops each into its place of the transposed image ([pixel][channel], 16 bytes of pad per row) and the image leaves as whole
// NHWC rows, pad channels zeroed on the way.  27 us -> the time the bytes take (15.9 MB in, 11.8 MB out per 256-batch).
constexpr int kPackFlatThreads = 512;
constexpr int kPackFlatLoads = 32;  // x 512 threads = 16 384 floats per pass (43 planes x 361 = 15 523)
template <typename T>
__global__ __launch_bounds__(kPackFlatThreads) void pack_input_flat_kernel(const float* __restrict__ planes, T* __restrict__ out,
                                                                            BatchGeom g, int cin, int cs, int board,
                                                                            const int* __restrict__ perm, int n0 = 0) {
    constexpr int EPP = ElemTraits<T>::kPieceElems;
    extern __shared__ __attribute__((aligned(16))) unsigned char pk_smem[];
    const int n = n0 + blockIdx.x, tid = threadIdx.x;
    const int bs = g.bsz[n], npix = bs * bs, ppr = cs / EPP, stride = cs * (int)sizeof(T) + 16;
    const int B2 = board * board, total = cin * B2;
    const float* src = planes + (size_t)(perm ? perm[n] : n) * total;  // perm: device sample -> caller's slot
    T* dst = out + (size_t)n * g.slot_pix * cs;
    const float r_b2 = 1.0f / (float)B2, r_b = 1.0f / (float)board;
    for (int base = 0; base < total; base += kPackFlatThreads * kPackFlatLoads) {
        float f[kPackFlatLoads];
#pragma unroll
        for (int k = 0; k < kPackFlatLoads; ++k) {
            const int e = base + k * kPackFlatThreads + tid;
            f[k] = e < total ? src[e] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < kPackFlatLoads; ++k) {
            const int e = base + k * kPackFlatThreads + tid;
            // e = (c * board + y) * board + x; quotients by a reciprocal multiply ((e + 0.5) / d is never within 1e-3 of an
            // integer, e < 2^16)
            const int c = (int)(((float)e + 0.5f) * r_b2), r = e - c * B2;
            const int y = (int)(((float)r + 0.5f) * r_b), x = r - y * board;
            if (e < total && y < bs && x < bs) *(T*)(pk_smem + (size_t)(y * bs + x) * stride + c * (int)sizeof(T)) = from_float<T>(f[k]);
        }
    }
    __syncthreads();
    for (int it = tid; it < npix * ppr; it += kPackFlatThreads) {
        const int q = it / ppr, piece = it - q * ppr;
        T v[EPP];
        *(uint4*)v = *(const uint4*)(pk_smem + (size_t)q * stride + piece * 16);
#pragma unroll
        for (int e = 0; e < EPP; ++e)
            if (piece * EPP + e >= cin) v[e] = from_float<T>(0.f);  // the pad channels were never written
        *(uint4*)(dst + (size_t)it * EPP) = *(uint4*)v;
    }
}
static inline size_t pack_input_flat_lds(int slot_pix, int cs, int elem) { return (size_t)slot_pix * (cs * elem + 16); }

// pixels per pass and the LDS they take (<= 64 KiB: no opt-in needed)
static inline int pack_input_chunk(int slot_pix, int cs, int elem) {
    return std::min((slot_pix + kPackSplit - 1) / kPackSplit, (64 * 1024) / (cs * elem + 16));
}

// The same activations from PACKED planes (csrc/host/packed_planes.h): record of a sample = uint32
// bits[nbin][12] (bit y*bs+x of a 0/1 plane, the sample's OWN cell order -- no re-padding into the NN grid) followed by 8
// floats (the value of each broadcast plane).  1.8 KB per sample instead of 62 KB over PCIe and out of HBM.
// kPackSplit workgroups per sample (pixel ranges): the record goes to LDS with one coalesced load, then thread = (pixel,
// 16-byte piece) in output order -- every store of a wave is 1 KiB contiguous.
// The geometry arrays of a MIXED batch, moved from the engine's pinned staging ring to device memory by the forward's own
// stream (one workgroup; `stage` is host memory the device reads over PCIe: off[n+1] | bsz[n] | perm[n] at strides of
// max_batch + 1 / max_batch).  Not a hipMemcpyAsync: see Engine::enqueue_inputs.
__global__ __launch_bounds__(256) void geom_stage_kernel(const int* __restrict__ stage, int max_batch, int n, int* __restrict__ off,
                                                         int* __restrict__ bsz, int* __restrict__ perm) {
    for (int i = threadIdx.x; i <= n; i += blockDim.x) off[i] = stage[i];
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        bsz[i] = stage[max_batch + 1 + i];
        perm[i] = stage[2 * max_batch + 1 + i];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void pack_bits_kernel(const unsigned* __restrict__ records, int rec_words, int nbin,
                                                        T* __restrict__ out, BatchGeom g, int cin, int cs,
                                                        const int* __restrict__ perm, int n0 = 0, int split = kPackSplit) {
    constexpr int EPP = ElemTraits<T>::kPieceElems;
    __shared__ unsigned rec[40 * 13 + 8];  // 13 words per plane: the eight planes a wave reads at once fall on different banks
    const int n = n0 + blockIdx.x / split, part = blockIdx.x % split, tid = threadIdx.x;  // split: workgroups per sample
    const int bs = g.bsz[n], npix = bs * bs, ppr = cs / EPP;
    const int per = (npix + split - 1) / split, pbeg = part * per, pend = min(npix, pbeg + per);
    const unsigned* src = records + (size_t)(perm ? perm[n] : n) * rec_words;  // perm: device sample -> caller's slot
    for (int i = tid; i < rec_words; i += 256) rec[i < nbin * 12 ? (i / 12) * 13 + i % 12 : nbin * 13 + (i - nbin * 12)] = src[i];
    __syncthreads();
    T* dst = out + (size_t)n * g.slot_pix * cs;
    for (int it = pbeg * ppr + tid; it < pend * ppr; it += 256) {
        const int pp = it / ppr, piece = it - pp * ppr;
        const int word = pp >> 5, sh = pp & 31;
        T v[EPP];
#pragma unroll
        for (int e = 0; e < EPP; ++e) {
            const int c = piece * EPP + e;
            float f = 0.f;
            if (c < nbin) f = (float)((rec[c * 13 + word] >> sh) & 1u);
            else if (c < cin) f = __uint_as_float(rec[nbin * 13 + (c - nbin)]);
            v[e] = from_float<T>(f);
        }
        *(uint4*)(dst + (size_t)it * EPP) = *(uint4*)v;
    }
}

struct FcDev {
    const float* wt;  // transposed [in][out]
    const float* b;   // [out]
    int in, out;
};

// y[o] = act(b[o] + sum_i x[i] * wt[i][o]); x and y in LDS; all threads of the block cooperate.
// The i-loop is unrolled so that 8 independent weight loads are in flight per thread: these
// launches run one workgroup per CU and are bound by L2 latency, not bandwidth.
__device__ __forceinline__ void block_fc(const FcDev fc, const float* x, float* y, int act, int tid, int nt) {
    for (int o = tid; o < fc.out; o += nt) {
        float s0 = 0.f, s1 = 0.f;
        const float* w = fc.wt + o;
        int i = 0;
        for (; i + 8 <= fc.in; i += 8) {
            float wv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) wv[u] = w[(size_t)(i + u) * fc.out];
#pragma unroll
            for (int u = 0; u < 8; u += 2) { s0 += x[i + u] * wv[u]; s1 += x[i + u + 1] * wv[u + 1]; }
        }
        for (; i < fc.in; ++i) s0 += x[i] * w[(size_t)i * fc.out];
        y[o] = activate(fc.b[o] + (s0 + s1), act);
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
#pragma unroll 8
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

// ---- SE unit, three bandwidth-shaped launches -------------------------------------------------
// se_pool_kernel : grid (n, SPLIT): each workgroup sums / maxes a slice of the sample's pixels with
//                  16-byte loads and writes per-channel partials  part[n][split][2][cs]
// se_fc_kernel   : grid n: folds the partials into (mean, scaled mean, max), squeeze FC, excite FC
//                  -> gate[n][0..C) = sigmoid(gamma), gate[n][C..2C) = beta
// se_scale_kernel: out = act(gamma * x + beta + residual), 16-byte loads/stores, no index search
constexpr int kSeSplit = 4;

template <typename T>
__global__ __launch_bounds__(256) void se_pool_kernel(const T* __restrict__ x, float* __restrict__ part, BatchGeom g,
                                                      int cs, int n0 = 0) {  // n0: first sample of the launch
    constexpr int EPP = ElemTraits<T>::kPieceElems;
    __shared__ float red[2][256 * 8 / 8 * 8];  // [sum|max][thread][EPP<=8]
    const int n = n0 + blockIdx.x / kSeSplit, sp = blockIdx.x % kSeSplit;
    const int tid = threadIdx.x;
    const int bs = g.bsz[n], npix = bs * bs;
    const int ppr = cs / EPP;                 // 16-byte pieces per pixel row
    const int prows = 256 / ppr;              // pixel rows handled concurrently (>=1 when cs <= 256*EPP)
    const int piece = tid % ppr, prow = tid / ppr;
    const int per = (npix + kSeSplit - 1) / kSeSplit;
    const int p0 = sp * per, p1 = min(npix, p0 + per);
    float sum[EPP], mx[EPP];
#pragma unroll
    for (int e = 0; e < EPP; ++e) { sum[e] = 0.f; mx[e] = -5000.f; }
    if (prow < prows) {
        const T* xs = x + (size_t)n * g.slot_pix * cs + piece * EPP;
#pragma unroll 4
        for (int p = p0 + prow; p < p1; p += prows) {
            T v[EPP];
            *(uint4*)v = *(const uint4*)(xs + (size_t)p * cs);
#pragma unroll
            for (int e = 0; e < EPP; ++e) {
                const float f = to_float(v[e]);
                sum[e] += f;
                mx[e] = fmaxf(mx[e], f);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < EPP; ++e) { red[0][tid * EPP + e] = sum[e]; red[1][tid * EPP + e] = mx[e]; }
    __syncthreads();
    if (prow == 0 && piece < ppr) {
        for (int q = 1; q < prows; ++q) {
#pragma unroll
            for (int e = 0; e < EPP; ++e) {
                sum[e] += red[0][(q * ppr + piece) * EPP + e];
                mx[e] = fmaxf(mx[e], red[1][(q * ppr + piece) * EPP + e]);
            }
        }
        float* dst = part + ((size_t)(n * kSeSplit + sp) * 2) * cs + piece * EPP;
#pragma unroll
        for (int e = 0; e < EPP; ++e) { dst[e] = sum[e]; dst[cs + e] = mx[e]; }
    }
}

// kSeFcThreads threads per workgroup: the kernel is a chain of dependent L2 loads per thread (the squeeze FC's 3C rows split over the row
// groups of the workgroup), so its time is the length of that chain -- 115 rows per thread with 256 threads (27 us for a 384-channel
// layer), 28 with 1024.
constexpr int kSeFcThreads = 1024;
__global__ __launch_bounds__(kSeFcThreads) void se_fc_kernel(const float* __restrict__ part, float* __restrict__ gate,
                                                    BatchGeom g, int C, int cs, FcDev squeeze, FcDev excite, int act, int n0 = 0) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* pool = (float*)smem;          // [3C]
    float* mid = pool + 3 * C;           // [se_size]
    float* scratch = mid + squeeze.out;  // [NT]
    constexpr int NT = kSeFcThreads;
    const int n = n0 + blockIdx.x, tid = threadIdx.x;
    const int bs = g.bsz[n];
    const float npix = (float)(bs * bs), bd = (float)bs - 14.f;
    for (int c = tid; c < C; c += NT) {
        float s = 0.f, m = -5000.f;
        for (int sp = 0; sp < kSeSplit; ++sp) {
            const float* src = part + ((size_t)(n * kSeSplit + sp) * 2) * cs;
            s += src[c];
            m = fmaxf(m, src[cs + c]);
        }
        const float mean = s / npix;
        pool[c] = mean;
        pool[C + c] = mean * (bd / 10.f);
        pool[2 * C + c] = m;
    }
    __syncthreads();
    // squeeze: NT threads = (out/4 column quads) x (row groups); each thread streams 16-byte weight
    // loads (unrolled: 8 in flight), partial sums are folded through LDS
    const int so = squeeze.out;
    if ((so & 3) == 0 && so <= 256) {
        const int quads = so / 4, groups = NT / quads;
        const int oq = tid % quads, grp = tid / quads;
        f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
        if (grp < groups) {
            const float* w = squeeze.wt + oq * 4;
#pragma unroll 8
            for (int i = grp; i < squeeze.in; i += groups) s4 += pool[i] * *(const f32x4*)(w + (size_t)i * so);
        }
        __syncthreads();  // pool no longer needed below this point except via s4
        float* red = scratch;  // [groups][so] would exceed NT floats: fold in rounds of NT floats
        for (int g0 = 0; g0 < groups; g0 += NT / so) {
            if (grp < groups && grp >= g0 && grp < g0 + NT / so) *(f32x4*)(red + (grp - g0) * so + oq * 4) = s4;
            __syncthreads();
            if (tid < so) {
                float a = g0 == 0 ? squeeze.b[tid] : mid[tid];
                for (int q = 0; q < NT / so && g0 + q < groups; ++q) a += red[q * so + tid];
                mid[tid] = a;
            }
            __syncthreads();
        }
        if (tid < so) mid[tid] = activate(mid[tid], act);
    } else {
        block_fc(squeeze, pool, mid, act, tid, NT);
    }
    __syncthreads();
    for (int o = tid; o < excite.out; o += NT) {
        float s = excite.b[o];
        const float* w = excite.wt + o;
#pragma unroll 16
        for (int i = 0; i < excite.in; ++i) s += mid[i] * w[(size_t)i * excite.out];
        // gate[n][0][c] = sigmoid(gamma_c), gate[n][1][c] = beta_c; channel stride cs, pads stay 0
        gate[(size_t)n * 2 * cs + (o < C ? o : cs + (o - C))] = o < C ? 1.0f / (1.0f + fast_exp(-s)) : s;
    }
}

// out = act(gate_gamma[n][c] * x + beta[n][c] + res).  grid (ceil(slot_pix*ppr / (256*4)), n);
// each thread owns 4 16-byte pieces (all loads issued before the math).
constexpr int kScaleUnroll = 4;
template <typename T>
__global__ __launch_bounds__(256) void se_scale_kernel(const T* __restrict__ x, const T* __restrict__ res,
                                                       T* __restrict__ out, const float* __restrict__ gate,
                                                       BatchGeom g, int C, int cs, int act, int n0 = 0) {
    constexpr int EPP = ElemTraits<T>::kPieceElems;
    const int n = n0 + blockIdx.y;
    const int bs = g.bsz[n], npix = bs * bs;
    const int ppr = cs / EPP, total = npix * ppr;
    const size_t base = (size_t)n * g.slot_pix * cs;
    const float* gm = gate + (size_t)n * 2 * cs;  // [2][cs], pad channels hold 0
    uint4 vx[kScaleUnroll], vr[kScaleUnroll];
    int idx[kScaleUnroll];
#pragma unroll
    for (int u = 0; u < kScaleUnroll; ++u) {
        idx[u] = (blockIdx.x * kScaleUnroll + u) * 256 + threadIdx.x;
        if (idx[u] < total) {
            vx[u] = *(const uint4*)(x + base + (size_t)idx[u] * EPP);
            if (res) vr[u] = *(const uint4*)(res + base + (size_t)idx[u] * EPP);
        }
    }
#pragma unroll
    for (int u = 0; u < kScaleUnroll; ++u) {
        if (idx[u] >= total) continue;
        const int c0 = (idx[u] % ppr) * EPP;
        const T* px = (const T*)&vx[u];
        const T* pr = (const T*)&vr[u];
        float ga[EPP], be[EPP];
#pragma unroll
        for (int e = 0; e < EPP; e += 4) {
            *(f32x4*)(ga + e) = *(const f32x4*)(gm + c0 + e);
            *(f32x4*)(be + e) = *(const f32x4*)(gm + cs + c0 + e);
        }
        T vo[EPP];
#pragma unroll
        for (int e = 0; e < EPP; ++e) {
            float v = ga[e] * to_float(px[e]) + be[e];  // pad channels: 0 * 0 + 0
            if (res) v += to_float(pr[e]);
            vo[e] = from_float<T>(activate(v, act));
        }
        *(uint4*)(out + base + (size_t)idx[u] * EPP) = *(uint4*)vo;
    }
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
    const int* perm;  // device sample -> caller's slot (outputs are written in the caller's order); null = identity
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
    // two workgroups per sample: the policy half and the value half share nothing, and each is a chain of dependent
    // small steps (pool -> FC -> FC -> per-pixel pass) that one workgroup can only run one after the other
    const int n = blockIdx.x >> 1, role = blockIdx.x & 1, tid = threadIdx.x, nt = blockDim.x;
    const int bs = g.bsz[n], npix = bs * bs, B2 = h.board * h.board;
    const T* pc = pconv + (size_t)n * g.slot_pix * h.cs_p;
    const T* vc = vconv + (size_t)n * g.slot_pix * h.cs_v;
    const int on_ = h.perm ? h.perm[n] : n;  // the caller's slot

    if (role == 0) {
        // policy: pooling -> intermediate FC (act) -> pass FC
        block_pool<T>(pc, npix, h.Cp, h.cs_p, bs, false, pool, scratch, tid, nt);
        block_fc(h.p_inter, pool, pinter, h.act, tid, nt);
        __syncthreads();
        block_fc(h.pass_fc, pinter, h.pass + (size_t)on_ * h.pass_fc.out, kIdentity, tid, nt);
    } else {
        // value: pooling -> intermediate FC (act) -> misc FC
        block_pool<T>(vc, npix, h.Cv, h.cs_v, bs, true, pool, scratch, tid, nt);
        block_fc(h.v_inter, pool, inter, h.act, tid, nt);
        __syncthreads();
        block_fc(h.v_misc, inter, h.misc + (size_t)on_ * h.v_misc.out, kIdentity, tid, nt);
    }

    // per-pixel outputs in the NN grid (off-board cells of a smaller sample = 0)
    for (int cell = tid; cell < B2; cell += nt) {
        const int y = cell / h.board, x = cell - y * h.board;
        const bool on = y < bs && x < bs;
        const int pp = y * bs + x;
        if (role == 0) {
            float pr[8];
            for (int k = 0; k < h.prob_ch; ++k) pr[k] = on ? h.prob_b[k] : 0.f;
            if (on) {
                for (int c = 0; c < h.Cp; ++c) {
                    const float v = to_float(pc[(size_t)pp * h.cs_p + c]) + pinter[c];
                    for (int k = 0; k < h.prob_ch; ++k) pr[k] += v * h.prob_w[k * h.Cp + c];
                }
            }
            for (int k = 0; k < h.prob_ch; ++k) h.prob[((size_t)on_ * h.prob_ch + k) * B2 + cell] = pr[k];
        } else {
            float ow = on ? h.own_b[0] : 0.f;
            if (on)
                for (int c = 0; c < h.Cv; ++c) ow += to_float(vc[(size_t)pp * h.cs_v + c]) * h.own_w[c];
            h.own[(size_t)on_ * B2 + cell] = ow;
        }
    }
}

}  // namespace sayuri
