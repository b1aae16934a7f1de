// head_board.h -- both heads of one sample in one workgroup (fp16 engine, normal policy head).
//
// Replaces, per batch, the two 1x1 head convolutions (each re-reading the 47 MB trunk) + head_tail_kernel
// (110 us of a 4.08 ms step at batch 256) by ONE launch of one workgroup per sample:
//   trunk rows of the sample -> LDS by LDS-DMA (32-channel chunks, DEPTH in flight) -> [policy | value] head convolution on
//   the matrix cores (the two weight matrices stacked: rows 0..PT-1 policy, PT.. value) -> bias + activation in the
//   accumulators -> (a) the per-pixel planes (policy planes, ownership) as a second tiny MFMA product straight from the
//   accumulators, (b) global pooling of both heads by register / DPP reduction -> the four FCs -> spatial bias folded into
//   the per-pixel row bias -> NN-grid fp32 outputs in the caller's order.  The head planes never leave the registers.
// Reference: blas_forward_pipe.cc:449-580 (policy head :449-536, value head :538-580), GlobalPooling<false/true>
// se_unit.cc:9-68.  The head planes stay fp32 here (the separate kernels round them to fp16 in between).
#pragma once
#include "common.h"
#include "conv_board.h"
#include "small_ops.h"

namespace sayuri {

constexpr int kHeadPix = 384;  // pixel slots (24 column tiles of 16): boards up to 19x19

struct HeadBoardParams {
    const void* trunk;    // [slot][pix][cs] fp16, with the kZeroPrefix bytes in front (not needed here)
    const void* w;        // stacked head weights, MFMA image [chunk][4][rows][8] fp16, rows = PT + VT
    const void* w2;       // per-pixel weights (policy planes over the policy rows, ownership over the value rows), MFMA image
                          // [pair of row tiles][4][16][8] fp16 in the accumulator's channel order (see head_board_kernel)
    const float* bias;    // [rows]
    BatchGeom g;
    int cs;               // trunk channel stride (multiple of 32)
    int PT, VT;           // policy / value rows, each a multiple of 16 (>= Cp / Cv)
    HeadParams h;         // FCs, per-pixel weights, outputs, perm, act (small_ops.h)
    unsigned long long* dbg;  // SAYURI_HEADS_DBG: s_memtime stamps of workgroups 0-3, wave 0 ([wg][8])
    int n0;                   // first sample of the launch (a chain of the batch, Engine::forward)
};

// floats of the small-vector area at the END of the LDS (outside the rings):
// pool[3 ROWS] | inter[3 ROWS] | pinter[ROWS] | rowbias[16] | red[max(16 ROWS, 1024)]
__host__ __device__ inline int head_vec_floats(int rows) { return 7 * rows + 16 + (16 * rows > 1024 ? 16 * rows : 1024); }
// LDS budget of head_board_kernel<RT, DEPTH>: weights + DEPTH ring slots + vectors
__host__ inline bool head_board_fits(int rows, int nchunks, int depth) {
    return (size_t)nchunks * 4 * rows * 16 + (size_t)depth * 4 * kHeadPix * 16 + sizeof(float) * head_vec_floats(rows) <= 160 * 1024;
}

// Partial sums of y = W^T x by `nt` threads (thread = (output, slice of the inputs); loads of a slice go out eight at a
// time): red[slice * out + o].  fold_fc adds the slices and the bias.  Latency-bound FCs: every workgroup reads the same
// few KB from L2, what counts is the number of dependent round trips (block_fc: in / 8 of them; here in / (8 parts)).
__device__ __forceinline__ int split_fc(const FcDev fc, const float* x, float* red, int t, int nt) {
    const int parts = max(1, min(nt / fc.out, (fc.in + 7) / 8));
    const int per = (fc.in + parts - 1) / parts;
    for (int idx = t; idx < parts * fc.out; idx += nt) {  // one pass unless out > nt
        const int o = idx % fc.out, part = idx / fc.out;
        const int i0 = part * per, i1 = min(fc.in, i0 + per);
        const float* w = fc.wt + o;
        float a = 0.f;
        for (int i = i0; i < i1; i += 8) {
            float wv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) wv[u] = i + u < i1 ? w[(size_t)(i + u) * fc.out] : 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) a += (i + u < i1 ? x[i + u] : 0.f) * wv[u];
        }
        red[idx] = a;
    }
    return parts;
}
__device__ __forceinline__ void fold_fc(const FcDev fc, const float* red, int parts, float* y, int act, int t, int nt) {
    for (int o = t; o < fc.out; o += nt) {
        float a = fc.b[o];
        for (int k = 0; k < parts; ++k) a += red[k * fc.out + o];
        y[o] = activate(a, act);
    }
}

// everything after the K loop, with the activation as a template parameter (no per-element switch)
template <int RT, int ACT>
__device__ __forceinline__ void head_board_tail(const HeadBoardParams& hp, unsigned char* smem, f32x4 (&acc)[RT][3], const f16x8 (&a2)[RT / 2],
                                                int wave, int lane, int n, int bs, unsigned long long* dbg) {
    constexpr int ROWS = RT * 16;
    const HeadParams& h = hp.h;
    const int tid = wave * 64 + lane, q = lane >> 4, px = lane & 15;
    const int npix = bs * bs, B2 = h.board * h.board;
    const int on_ = h.perm ? h.perm[n] : n;
    float* vec = (float*)(smem + 160 * 1024) - head_vec_floats(ROWS);
    float* pool = vec;
    float* inter = pool + 3 * ROWS;
    float* pinter = inter + 3 * ROWS;
    float* rb = pinter + ROWS;  // bias of the per-pixel rows: prob_b[k] + <prob_w[k], pinter> | own_b
    float* red = rb + 16;

    // ---- activation in the accumulators
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if constexpr (ACT == kMish) {
                const f32x2 lo = mish2(f32x2{acc[i][j][0], acc[i][j][1]}), hi = mish2(f32x2{acc[i][j][2], acc[i][j][3]});
                acc[i][j] = f32x4{lo[0], lo[1], hi[0], hi[1]};
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = activate(acc[i][j][r], ACT);
            }
        }
    // ---- per-pixel rows (policy planes, ownership) = W2 x planes on the matrix cores.  The B operand of k-group q is
    // the lane's own 4 + 4 channels of a PAIR of row tiles (rows 32t + 4q + r and 32t + 16 + 4q + r); the host laid W2
    // out in the same order, so no lane exchange is needed.  fp16 planes, fp32 accumulation (what the separate kernels do).
    f32x4 d[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        d[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < RT / 2; ++t) {
            f16x8 b;
#pragma unroll
            for (int r = 0; r < 4; ++r) { b[r] = (f16)acc[2 * t][j][r]; b[4 + r] = (f16)acc[2 * t + 1][j][r]; }
            d[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2[t], b, d[j], 0, 0, 0);
        }
    }
    // ---- global pooling: lane over its three column tiles (valid pixels only), DPP over the 16 pixel lanes, LDS over the waves
    float* wsum = red;
    float* wmax = red + 8 * ROWS;
    bool valid[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) valid[j] = (wave + 8 * j) * 16 + px < npix;
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        f32x4 s4, m4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float a = 0.f, b = -5000.f;
#pragma unroll
            for (int j = 0; j < 3; ++j) { a += valid[j] ? acc[i][j][r] : 0.f; b = valid[j] ? fmaxf(b, acc[i][j][r]) : b; }
            a += row_ror<8>(a); b = fmaxf(b, row_ror<8>(b));
            a += row_ror<4>(a); b = fmaxf(b, row_ror<4>(b));
            a += row_ror<2>(a); b = fmaxf(b, row_ror<2>(b));
            a += row_ror<1>(a); b = fmaxf(b, row_ror<1>(b));
            s4[r] = a; m4[r] = b;
        }
        if (px == 0) {
            *(f32x4*)(wsum + wave * ROWS + i * 16 + 4 * q) = s4;
            *(f32x4*)(wmax + wave * ROWS + i * 16 + 4 * q) = m4;
        }
    }
    __syncthreads();
    if (dbg) dbg[2] = __builtin_amdgcn_s_memtime();
    float* vpool = pool + 3 * hp.PT;
    if (tid < ROWS) {
        float ss = 0.f, mm = -5000.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) { ss += wsum[k * ROWS + tid]; mm = fmaxf(mm, wmax[k * ROWS + tid]); }
        const float mean = ss / (float)npix, bd = (float)bs - 14.f;
        if (tid < hp.PT) {  // policy rows: GlobalPooling<false> over Cp channels
            if (tid < h.Cp) { pool[tid] = mean; pool[h.Cp + tid] = mean * (bd / 10.f); pool[2 * h.Cp + tid] = mm; }
        } else {            // value rows: GlobalPooling<true> over Cv channels, kept behind the policy vector
            const int c = tid - hp.PT;
            if (c < h.Cv) { vpool[c] = mean; vpool[h.Cv + c] = mean * (bd / 10.f); vpool[2 * h.Cv + c] = mean * (bd * bd / 100.f - 0.1f); }
        }
    }
    __syncthreads();
    if (dbg) dbg[3] = __builtin_amdgcn_s_memtime();
    // ---- the four FCs: waves 0-1 the policy chain, waves 2-7 the value chain (three times the weights)
    const bool pol = tid < 128;
    const int t2 = pol ? tid : tid - 128, nt2 = pol ? 128 : 384;
    float* red2 = pol ? red : red + 256;  // split_fc keeps parts * out <= nt2 (or one slice of `out` sums)
    float* vinter = inter + 3 * hp.PT;
    const FcDev f1 = pol ? h.p_inter : h.v_inter;
    const int parts = split_fc(f1, pol ? pool : vpool, red2, t2, nt2);
    __syncthreads();
    fold_fc(f1, red2, parts, pol ? pinter : vinter, h.act, t2, nt2);
    __syncthreads();
    if (dbg) dbg[4] = __builtin_amdgcn_s_memtime();
    const FcDev f2 = pol ? h.pass_fc : h.v_misc;
    const int parts2 = split_fc(f2, pol ? pinter : vinter, red2, t2, nt2);
    // row bias of the per-pixel product (wave 7): the spatial bias pinter[c] goes through prob_w once per sample
    if (wave == 7) {
        const int k = lane >> 3, part = lane & 7;
        float a = 0.f;
        if (k < h.prob_ch)
            for (int c = part; c < h.Cp; c += 8) a += h.prob_w[k * h.Cp + c] * pinter[c];
        a += __shfl_xor(a, 1);
        a += __shfl_xor(a, 2);
        a += __shfl_xor(a, 4);
        if (part == 0) rb[k] = k < h.prob_ch ? a + h.prob_b[k] : 0.f;
        if (lane == 0) rb[h.prob_ch] = h.own_b[0];
    }
    __syncthreads();
    fold_fc(f2, red2, parts2, pol ? h.pass + (size_t)on_ * f2.out : h.misc + (size_t)on_ * f2.out, kIdentity, t2, nt2);
    if (dbg) dbg[5] = __builtin_amdgcn_s_memtime();
    // ---- per-pixel outputs into the NN grid: lane (px, q) holds rows 4q .. 4q+3 of its pixels
    if (q * 4 <= h.prob_ch) {
        const f32x4 rb4 = *(const f32x4*)(rb + 4 * q);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int p = (wave + 8 * j) * 16 + px;
            if (p >= npix) continue;
            const int y = p / bs, cell = y * h.board + (p - y * bs);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * q + r;
                const float v = d[j][r] + rb4[r];
                if (row < h.prob_ch) h.prob[((size_t)on_ * h.prob_ch + row) * B2 + cell] = v;
                else if (row == h.prob_ch) h.own[(size_t)on_ * B2 + cell] = v;
            }
        }
    }
    if (bs < h.board)  // off-board cells of a smaller sample = 0
        for (int cell = tid; cell < B2; cell += 512) {
            const int y = cell / h.board, x = cell - y * h.board;
            if (y < bs && x < bs) continue;
            for (int k = 0; k < h.prob_ch; ++k) h.prob[((size_t)on_ * h.prob_ch + k) * B2 + cell] = 0.f;
            h.own[(size_t)on_ * B2 + cell] = 0.f;
        }
}

template <int RT, int DEPTH>  // row tiles of 16 = (PT + VT) / 16 (even); trunk chunks in flight
__global__ __launch_bounds__(512) void head_board_kernel(const HeadBoardParams hp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int ROWS = RT * 16;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = hp.n0 + blockIdx.x;
    const int bs = hp.g.bsz[n], npix = bs * bs;
    const int nchunks = hp.cs / kChunk;

    // LDS: [A image: nchunks x 4 planes x ROWS x 16 B][B ring: DEPTH x (4 planes x 384 x 16 B)] ... [vectors]
    const uint32_t a_lds = (uint32_t)(uintptr_t)smem;
    const int a_bytes = nchunks * 4 * ROWS * 16;
    const uint32_t b_lds = a_lds + a_bytes;
    constexpr int B_BYTES = 4 * kHeadPix * 16;

    // ---- DMA: weights (linear), trunk chunk c: instruction q = wave + 8*i covers plane q & 3 of pixel block q >> 2
    unsigned long long* dbg = hp.dbg && n < 4 && tid == 0 ? hp.dbg + n * 8 : nullptr;
    if (dbg) dbg[0] = __builtin_amdgcn_s_memtime();
    const unsigned char* gx = (const unsigned char*)hp.trunk + (size_t)n * hp.g.slot_pix * hp.cs * 2;
    const int a_instr = a_bytes / 1024;
    for (int q = wave; q < a_instr; q += 8) glds16_s(lane * 16, (const unsigned char*)hp.w + (size_t)q * 1024, a_lds + q * 1024);
    uint32_t boff[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int q = wave + 8 * i, kgq = q & 3, blk = q >> 2;  // 24 instructions: 6 pixel blocks x 4 planes
        const int p = blk * 64 + lane;
        boff[i] = (uint32_t)((p < npix ? p : 0) * hp.cs * 2 + kgq * 16);
    }
    auto issue_b = [&](int chunk) {
        const int slot = chunk % DEPTH;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int q = wave + 8 * i;
            glds16_s(boff[i], gx + chunk * (kChunk * 2), b_lds + slot * B_BYTES + ((q & 3) * kHeadPix + (q >> 2) * 64) * 16);
        }
    };
    for (int c = 0; c < DEPTH - 1 && c < nchunks; ++c) issue_b(c);

    const int kg = lane >> 4;
    f16x8 a2[RT / 2];  // per-pixel weights of this lane (row lane & 15, k-group kg)
#pragma unroll
    for (int t = 0; t < RT / 2; ++t) a2[t] = *(const f16x8*)((const unsigned char*)hp.w2 + ((t * 4 + kg) * 16 + (lane & 15)) * 16);
    f32x4 acc[RT][3];
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        const f32x4 b4 = *(const f32x4*)(hp.bias + i * 16 + 4 * kg);
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = b4;
    }
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        // chunks chunk+1 .. chunk+DEPTH-2 (3 instructions each per wave) may stay in flight; at the tail everything is waited for
        if (chunk + DEPTH - 2 < nchunks) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (DEPTH - 2)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // the chunk's tile (and, the first time, the weights) landed; slot (chunk - 1) % DEPTH is free
        if (chunk + DEPTH - 1 < nchunks) issue_b(chunk + DEPTH - 1);
        const unsigned char* A = smem + (size_t)(chunk * 4 + kg) * ROWS * 16 + (lane & 15) * 16;
        const unsigned char* Bc = smem + a_bytes + (chunk % DEPTH) * B_BYTES + (size_t)kg * kHeadPix * 16 + (lane & 15) * 16;
        f16x8 bf[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) bf[j] = *(const f16x8*)(Bc + (wave + 8 * j) * 256);  // column tile ct = wave + 8 j
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const f16x8 af = *(const f16x8*)(A + i * 256);
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf[j], acc[i][j], 0, 0, 0);
        }
    }
    if (dbg) dbg[1] = __builtin_amdgcn_s_memtime();
    switch (hp.h.act) {
    case kMish: head_board_tail<RT, kMish>(hp, smem, acc, a2, wave, lane, n, bs, dbg); break;
    case kReLU: head_board_tail<RT, kReLU>(hp, smem, acc, a2, wave, lane, n, bs, dbg); break;
    case kIdentity: head_board_tail<RT, kIdentity>(hp, smem, acc, a2, wave, lane, n, bs, dbg); break;
    case kSwish: head_board_tail<RT, kSwish>(hp, smem, acc, a2, wave, lane, n, bs, dbg); break;
    case kELU: head_board_tail<RT, kELU>(hp, smem, acc, a2, wave, lane, n, bs, dbg); break;
    case kSELU: head_board_tail<RT, kSELU>(hp, smem, acc, a2, wave, lane, n, bs, dbg); break;
    case kGELU: head_board_tail<RT, kGELU>(hp, smem, acc, a2, wave, lane, n, bs, dbg); break;
    default: head_board_tail<RT, kHardSwish>(hp, smem, acc, a2, wave, lane, n, bs, dbg); break;
    }
    if (dbg) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        dbg[6] = __builtin_amdgcn_s_memtime();
    }
}

}  // namespace sayuri
