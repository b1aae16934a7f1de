// conv_wino.h -- fp16 3x3 convolution as a FUSED Winograd F(2x2,3x3): input transform, the sixteen
// [output channel x tile x input channel] products on the matrix cores, output transform, bias, residual and
// activation in ONE kernel, nothing but the layer's input and output ever touches HBM.
//
// The reference evaluates its 3x3 layers as Winograd too (F(4x4,3x3), three kernels with the V / M tensors in global
// memory: src/neural/cuda/cuda_kernels.cu:521-667, cuda_layers.cc:455-619; CPU: winograd_convolution3.cc:12-278).
// Un-fused, that moves 6x the activation bytes and is HBM-bound on this chip (conv_mfma.h header); fused, F(2x2) does
// 2.25x fewer multiplies than the direct form (16 instead of 36 per 2x2 outputs; 2.03x on 19x19 boards, whose 10x10
// tiles cover 20x20) and keeps fp16 products at the error level of a plain dot product (the transform matrices hold
// only 0, +-1, +-1/2).
//
// Work split.  A workgroup owns KO_T = 64 output channels x NTL = 48 tiles (192 output pixels) and ALL sixteen
// transform points: 4 waves, wave xi holds M[xi][nu = 0..3][64 ch][48 tiles] in 192 accumulator registers per lane.
//   * raw input: per 32-channel chunk the 4x4 patches of the tile block (their union: a few board rows with a one /
//     two cell frame, <= 512 positions) are DMA'd into LDS once (`global_load_lds_dwordx4`, 64 contiguous bytes per
//     position, so the global side is coalesced) and double-buffered; layout and bank spreading: wino_pitch / wino_rot;
//   * input transform in registers: V = B^T d B.  The K loop runs in stages of (chunk, nu pair); per stage and 16 tiles
//     wave xi reads the two patch rows B^T's row xi combines (6 x ds_read_b128, two items ahead of their use), forms
//     t[j] = d[ia][j] +- d[ib][j] and V[xi][nu] = t[ja] +- t[jb] with packed fp16 FMAs: the transformed tensor is never
//     written anywhere (LDS stores are the slow port on this chip);
//   * weights U = G g G^T are transformed once at load time and stored in MFMA fragment order; a wave's eight
//     fragments of a stage are 8 KiB of contiguous global memory, loaded straight into a three-deep register ring two
//     stages ahead (no other wave needs them, so LDS would only add a round trip), waited for with hand-counted vmcnt;
//   * output transform: the nu half (A^T on the right) in registers, the xi half across the four waves through an
//     fp32 staging tile in LDS; then bias, residual, activation and 128-byte row segments out.
// Measured state and the list of what is still slow: DESIGN.md "Kernel 3".
#pragma once
#include "common.h"
#include "conv_glds.h"

namespace sayuri {

constexpr int kZeroPrefix = 4096;  // zero bytes every activation buffer carries in front of row 0

struct WinoCfg {
    static constexpr int KO_T = 64, NF = 3, NTL = 16 * NF, NWAVE = 4, NT = 256;  // NF = 16-tile fragment columns
    static constexpr int NPOS = 512;                       // raw positions of a tile block (multiple of 64)
    static constexpr int RAW_BYTES = NPOS * 64;            // [position][k-group][8 halves]
    static constexpr int Z_RS = KO_T * 4 + 16;             // staging row: 64 fp32 + pad
    static constexpr int STAGE_BYTES = 4 * 2 * NTL * Z_RS; // [xi][b][tile][ch]
    static constexpr int OUTROW_OFF = STAGE_BYTES;         // int out_row[NTL][4] behind the staging tile
    static constexpr size_t lds_bytes() { return STAGE_BYTES + NTL * 4 * 4; }
    static constexpr int ITEMS = NTL * 4 * 8 / NT;          // epilogue items (tile, a, b, 8 channels) per thread
    static_assert(NTL * 4 * 8 % NT == 0 && NTL * 4 <= NT, "epilogue split");
    static_assert(2 * RAW_BYTES <= STAGE_BYTES, "the raw ring lives inside the staging area");
};

struct WinoParams {
    ConvParams c;          // c.w = Winograd image [kt][chunk][xi][nu][m][lane][8], c.ko_pad multiple of 64
    const void* zeros;     // unused by the kernel proper (kept for symmetry with GldsParams)
    const int* tab_src;    // [block][NPOS]  activation row feeding each raw position, -1 = zero
    const int* tab_tile;   // [block][NTL]   lpos | lstr << 16 of each tile's 4x4 patch (top-left position, row stride)
    const int* tab_out;    // [block][NTL][4] activation row of output (a, b) of each tile, -1 = outside the board
    int num_blocks;        // tile blocks of the batch
    unsigned long long* dbg;  // optional s_memtime timeline [wg < 64][wave][16] (SAYURI_WINO_DBG)
};

__host__ __device__ inline int wino_tiles_per_side(int bs) { return (bs + 1) >> 1; }
// LDS image of a tile block's raw input: per raw row `pitch` positions of 64 bytes -- the tw+1 even patch columns, then
// the tw+1 odd ones, then one pad position.  The odd pitch and the rotation of the four 16-byte k-group slots of a
// position by ((pos >> 1) + row) & 3 were found by search (a bank simulation of ds_read_b128's lane groups): the
// sixteen tiles of a fragment column then hit sixteen different 16-byte bank groups for every patch cell (4.0 LDS cycles
// per read inside a sample, 4.5 averaged over blocks that cross samples; 10.5 with pitch 2tw+2 and no rotation).
__host__ __device__ inline int wino_pitch(int tw) { return 2 * tw + 3; }
__host__ __device__ inline int wino_rot(int pos, int row) { return ((pos >> 1) + row) & 3; }

// tile_off[i] = number of Winograd tiles of samples 0..i-1
__global__ __launch_bounds__(1024) void wino_prefix_kernel(BatchGeom g, int* __restrict__ tile_off) {
    __shared__ int s[1024];
    __shared__ int carry;
    const int tid = threadIdx.x;
    if (tid == 0) { carry = 0; tile_off[0] = 0; }
    __syncthreads();
    for (int base = 0; base < g.n_samples; base += 1024) {
        const int i = base + tid;
        int v = 0;
        if (i < g.n_samples) { const int tw = wino_tiles_per_side(g.bsz[i]); v = tw * tw; }
        s[tid] = v;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {
            const int add = tid >= d ? s[tid - d] : 0;
            __syncthreads();
            s[tid] += add;
            __syncthreads();
        }
        if (i < g.n_samples) tile_off[i + 1] = carry + s[tid];
        __syncthreads();
        if (tid == 1023) carry += s[1023];
        __syncthreads();
    }
}

// Index tables of one tile block (NTL consecutive tiles, sample-major, row-major inside a sample).  Subregion s =
// the raw rows one sample contributes: tile rows trlo..trhi need board rows 2*trlo-1 .. 2*trhi+2, columns -1 .. 2*tw.
__global__ __launch_bounds__(256) void wino_setup_kernel(BatchGeom g, const int* __restrict__ tile_off, int total_tiles,
                                                         int* __restrict__ tab_src, int* __restrict__ tab_tile,
                                                         int* __restrict__ tab_out) {
    constexpr int NTL = WinoCfg::NTL, NPOS = WinoCfg::NPOS;
    __shared__ int hdr[8 + 8 * kMaxSub];
    const int blk = blockIdx.x, tid = threadIdx.x;
    const int t0 = blk * NTL;
    if (tid == 0) {
        const int t1 = min(t0 + NTL, total_tiles);
        int lo = 0, hi = g.n_samples;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (tile_off[mid] <= t0) lo = mid; else hi = mid;
        }
        int n = lo, base = 0, cnt = 0;
        while (n < g.n_samples && cnt < kMaxSub) {
            const int off = tile_off[n];
            if (off >= t1) break;
            const int bs = g.bsz[n], tw = wino_tiles_per_side(bs);
            const int a = max(t0, off) - off, b = min(t1, tile_off[n + 1]) - off;
            const int trlo = a / tw, trhi = (b - 1) / tw, rows = 2 * (trhi - trlo + 1) + 2;
            int* sb = hdr + 8 + 8 * cnt;
            sb[0] = base; sb[1] = trlo; sb[2] = bs; sb[3] = n;
            sb[4] = off + a; sb[5] = off + b; sb[6] = rows; sb[7] = off;
            base += rows * wino_pitch(tw);
            ++cnt; ++n;
        }
        hdr[0] = cnt;
    }
    __syncthreads();
    const int nsub = hdr[0];
    for (int pos = tid; pos < NPOS; pos += blockDim.x) {
        int src = -1;
        for (int s = 0; s < nsub; ++s) {
            const int* sb = hdr + 8 + 8 * s;
            const int bs = sb[2], tw = wino_tiles_per_side(bs), w2 = wino_pitch(tw), rel = pos - sb[0];
            if (rel >= 0 && rel < sb[6] * w2) {
                const int r = rel / w2, idx = rel - r * w2, half = tw + 1;
                const int xc = idx < half ? 2 * idx : 2 * (idx - half) + 1;  // even columns first, then the odd ones (+ 1 pad cell)
                const int y = 2 * sb[1] - 1 + r, x = xc - 1;
                // bits 28-29: k-group rotation of this position in LDS (bank spreading, see conv_wino_kernel)
                if (y >= 0 && y < bs && x >= 0 && x < bs) src = (sb[3] * g.slot_pix + y * bs + x) | (wino_rot(pos, r) << 28);
                break;
            }
        }
        tab_src[(size_t)blk * NPOS + pos] = src;
    }
    if (tid < NTL) {
        const int gi = t0 + tid;
        int lstr = wino_pitch(wino_tiles_per_side(hdr[8 + 2])), lpos = 0, trl = 0;  // a dummy tile reads real positions, stores nothing
        int orow[4] = {-1, -1, -1, -1};
        if (gi < total_tiles) {
            for (int s = 0; s < nsub; ++s) {
                const int* sb = hdr + 8 + 8 * s;
                if (gi >= sb[4] && gi < sb[5]) {
                    const int bs = sb[2], tw = wino_tiles_per_side(bs), tt = gi - sb[7];
                    const int ty = tt / tw, tx = tt - ty * tw;
                    lstr = wino_pitch(tw);
                    trl = ty - sb[1];
                    lpos = sb[0] + 2 * trl * lstr + tx;  // patch column j sits at tx + (j >> 1) of the even / odd half
#pragma unroll
                    for (int ab = 0; ab < 4; ++ab) {
                        const int y = 2 * ty + (ab >> 1), x = 2 * tx + (ab & 1);
                        orow[ab] = (y < bs && x < bs) ? sb[3] * g.slot_pix + y * bs + x : -1;
                    }
                    break;
                }
            }
        }
        tab_tile[(size_t)blk * NTL + tid] = lpos | (lstr << 16) | ((trl & 1) << 24);
#pragma unroll
        for (int ab = 0; ab < 4; ++ab) tab_out[((size_t)blk * NTL + tid) * 4 + ab] = orow[ab];
    }
}

// Weight fragment load the compiler does not track: with LDS-DMA in flight hipcc waits for its own global loads with
// vmcnt(0), i.e. also for the loads issued a moment ago for two stages ahead -- a full memory latency per chunk.
// The fragments are waited for by hand with the exact count of younger VMEM instructions instead (wait_a).
template <int OFF> __device__ __forceinline__ void gload16(f16x8& dst, const void* uniform_base, uint32_t voff) {
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(uniform_base), "n"(OFF));
}
template <int N> __device__ __forceinline__ void wait_a(f16x8 (&a)[2][4]) {
    asm volatile("s_waitcnt vmcnt(%8)"
                 : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[0][2]), "+v"(a[0][3]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[1][2]), "+v"(a[1][3])
                 : "n"(N));
}

// Program order of the VMEM batches (8 instructions each) of conv_wino_kernel<NCH>, for the hand-counted vmcnt
// waits.  Batches: raw DMA of chunk c, weight loads of stage s.  Prologue: raw(0), A(0), A(1), raw(1).  Item k (stage
// k / NF, column k % NF): first, when item k+2 opens chunk c2 >= 1 ... wait, the barrier, raw(c2+1); then, at
// column 0, A(stage+2).
template <int NCH, int NF, int BI> struct WinoOrder {
    static constexpr int nstages = 2 * NCH, nitems = nstages * NF;
    // batches issued after batch (kind, idx) up to and including item `upto`'s own batches (`upto` = -1: prologue
    // only); with before_dma the raw batch of item `upto` (and its A batch) are not counted
    static constexpr int after(int kind, int idx, int upto, bool before_dma) {
        int cnt = 0;
        bool seen = false;
        auto ev = [&](int k2, int i2) {
            if (seen) cnt += k2 == 0 ? BI : 8;
            if (k2 == kind && i2 == idx) seen = true;
        };
        ev(0, 0); ev(1, 0); ev(1, 1);
        if (NCH > 1) ev(0, 1);
        for (int k = 0; k <= upto; ++k) {
            const bool last = k == upto;
            if (last && before_dma) break;
            if (k + 3 < nitems && (k + 3) % (2 * NF) == 0) {
                const int c2 = (k + 3) / (2 * NF);
                if (c2 + 1 < NCH) ev(0, c2 + 1);
            }
            if (k % NF == 0 && k / NF + 2 < nstages) ev(1, k / NF + 2);
        }
        return cnt;
    }
};

// wait until at most YOUNG younger LDS reads are outstanding (LDS returns in order)
template <int YOUNG> __device__ __forceinline__ void wait_patch(f16x8 (&r)[6]) {
    asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]) : "n"(YOUNG));
}

template <int ACT>
__device__ __forceinline__ void wino_store(const WinoParams& wp, const unsigned char* stage, const int* out_row, int kt, int tid) {
    constexpr int NTL = WinoCfg::NTL, RS = WinoCfg::Z_RS;
    const ConvParams& p = wp.c;
    f16* __restrict__ gout = (f16*)p.out;
    const f16* __restrict__ gres = (const f16*)p.res;
    const int cg = tid & 7;
    const int ko = kt * WinoCfg::KO_T + cg * 8;
    if (ko >= p.cout_s) return;  // cout_s is a multiple of 32: 8-channel groups never straddle it
    const f32x4 b0 = *(const f32x4*)(p.bias + ko), b1 = *(const f32x4*)(p.bias + ko + 4);
    // item = k*256 + tid: channel group = tid & 7, (tile, ab) = item >> 3
    constexpr int ITEMS = WinoCfg::ITEMS;
    int rows[ITEMS];
    f16x8 rr[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const int it = (k * 256 + tid) >> 3;
        rows[k] = out_row[it];  // [tile][ab] with it = tile*4 + ab
        if (gres) rr[k] = rows[k] >= 0 ? *(const f16x8*)(gres + (size_t)rows[k] * p.cout_s + ko) : f16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const int it = (k * 256 + tid) >> 3;
        const int tile = it >> 2, a = (it >> 1) & 1, b = it & 1;
        // Y[a][b] = sum_xi At[a][xi] Z[xi][b];  At = [[1,1,1,0],[0,1,-1,-1]]
        const unsigned char* z = stage + ((size_t)(b * NTL + tile) * RS) + cg * 32;
        constexpr int XS = 2 * NTL * RS;  // bytes between xi slabs
        const unsigned char* z1 = z + XS;
        const unsigned char* z2 = z + 2 * XS;
        const unsigned char* zo = a ? z + 3 * XS : z;
        const f32x4 p0 = *(const f32x4*)zo, p1 = *(const f32x4*)(zo + 16);
        const f32x4 q0 = *(const f32x4*)z1, q1 = *(const f32x4*)(z1 + 16);
        const f32x4 r0 = *(const f32x4*)z2, r1 = *(const f32x4*)(z2 + 16);
        f32x4 v0, v1;
        if (a) { v0 = q0 - r0 - p0; v1 = q1 - r1 - p1; }
        else { v0 = p0 + q0 + r0; v1 = p1 + q1 + r1; }
        v0 += b0; v1 += b1;
        float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        if (gres) {
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] += (float)rr[k][q];
        }
        f16x8 h;
#pragma unroll
        for (int q = 0; q < 8; ++q) h[q] = (f16)activate(v[q], ACT);
        if (rows[k] >= 0) *(f16x8*)(gout + (size_t)rows[k] * p.cout_s + ko) = h;
    }
}

// NCH = 32-channel chunks of the input (cin_s / 32): the K loop is fully unrolled, every ring index is static.
template <int NCH, int BI>
__global__ __launch_bounds__(256) void conv_wino_kernel(const WinoParams wp) {
    using Cfg = WinoCfg;
    constexpr int NPOS = Cfg::NPOS, NTL = Cfg::NTL, NF = Cfg::NF;  // BI: DMA instructions per wave and chunk (64 positions each)
    const ConvParams& p = wp.c;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int xi = __builtin_amdgcn_readfirstlane(tid >> 6);  // this wave's transform row; also its DMA plane
    // blockIdx -> (tile block, channel tile): the KT workgroups of a tile block sit on one XCD (blockIdx & 7), next to
    // each other in dispatch order, so the block's raw input is fetched into that XCD's L2 once
    const int kts = p.ko_pad / Cfg::KO_T;
    const int xcd = blockIdx.x & 7, rest = blockIdx.x >> 3;
    const int kt = rest % kts;
    const int blk = (rest / kts) * 8 + xcd;
    if (blk >= wp.num_blocks) return;
    unsigned long long* dbg = (wp.dbg && lane == 0 && blockIdx.x < 64) ? wp.dbg + ((size_t)blockIdx.x * 4 + xi) * 16 : nullptr;
    if (dbg) dbg[0] = __builtin_amdgcn_s_memtime();

    // ---- DMA role: plane kq = xi of every 64-position piece
    // DMA instruction q = xi + 4*i moves positions 16q .. 16q+15, four lanes (k-groups) per position: 64 contiguous
    // bytes of an activation row per position, 1 KiB of LDS per instruction.  Sources are 32-bit offsets from a
    // uniform base kZeroPrefix bytes in front of the activations: every activation buffer starts with that many
    // zero bytes, which is where halo / unused positions read from -- no per-lane pointer select, no address VALU
    // work per chunk (the chunk advance goes into the scalar base).
    const unsigned char* gin0 = (const unsigned char*)p.in - kZeroPrefix;
    uint32_t voff[BI];
#pragma unroll
    for (int i = 0; i < BI; ++i) {
        const int src = wp.tab_src[(size_t)blk * NPOS + (xi + 4 * i) * 16 + (lane >> 2)];
        // LDS slot (lane & 3) of a position holds k-group (slot - rot) & 3, rot = the raw row's rotation
        const uint32_t kq = (uint32_t)((lane & 3) - (src >> 28)) & 3u;
        voff[i] = src >= 0 ? (uint32_t)kZeroPrefix + (uint32_t)(src & 0x0fffffff) * (uint32_t)(p.cin_s * 2) + kq * 16u : 0u;
    }
    int* out_row = (int*)(smem + Cfg::OUTROW_OFF);
    if (tid < NTL * 4) out_row[tid] = wp.tab_out[(size_t)blk * NTL * 4 + tid];

    // ---- fragment addressing: lane -> tile (lane & 15) of each of the four 16-tile columns, k-group lane >> 4
    const int kg = lane >> 4;
    const int ia = xi == 0 ? 0 : (xi == 2 ? 2 : 1);
    const int ib = xi == 0 ? 2 : (xi == 1 ? 2 : (xi == 2 ? 1 : 3));
    // byte address of patch cell (row i of the 4x4 patch, column cell c): c = 0..3 -> even tx, odd tx, even tx+1, odd tx+1
    uint32_t cell[NF][2][4];
#pragma unroll
    for (int n = 0; n < NF; ++n) {
        const int pk = wp.tab_tile[(size_t)blk * NTL + n * 16 + (lane & 15)];
        const int lpos = pk & 0xffff, pitch = (pk >> 16) & 0xff, trl = pk >> 24, half = (pitch - 1) >> 1;
#pragma unroll
        for (int ab = 0; ab < 2; ++ab) {
            const int i = ab ? ib : ia, row = 2 * trl + i;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int pos = lpos + i * pitch + (c & 1) * half + (c >> 1);
                cell[n][ab][c] = (uint32_t)(uintptr_t)smem + (uint32_t)(pos * 64 + ((kg + wino_rot(pos, row)) & 3) * 16);
            }
        }
    }
    const f16 sgn = xi == 1 ? (f16)1.f : (f16)-1.f;
    const f16x8 sgn8 = {sgn, sgn, sgn, sgn, sgn, sgn, sgn, sgn};

    constexpr int nchunks = NCH;
    // weights: [kt][chunk][xi][nu][m] fragments of 1 KiB
    const unsigned char* gw = (const unsigned char*)p.w + ((size_t)kt * nchunks * 4 + xi) * 16 * 1024;  // uniform
    const uint32_t wlane = (uint32_t)lane * 16u;
    constexpr size_t W_CHUNK = 4 * 16 * 1024;

    auto issue_raw = [&](int chunk, int slot) {
        const unsigned char* gb = gin0 + chunk * (kChunk * 2);  // uniform
#if defined(SAYURI_WINO_ABL) && (SAYURI_WINO_ABL & 8)  // timing only: every DMA reads the zero prefix
        if (chunk > 1) {
#pragma unroll
            for (int i = 0; i < BI; ++i) glds16(gin0 + (lane & 3) * 16, smem + slot * Cfg::RAW_BYTES + (xi + 4 * i) * 1024);
            return;
        }
#endif
#pragma unroll
        for (int i = 0; i < BI; ++i) glds16(gb + voff[i], smem + slot * Cfg::RAW_BYTES + (xi + 4 * i) * 1024);
    };

    f32x4 acc[4][4][NF];  // [nu][m][n]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int c = 0; c < NF; ++c) acc[a][b][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    // The K loop runs in STAGES of (32-channel chunk, nu pair): stage s = 2*chunk + h covers nu = 2h, 2h+1.  A stage
    // needs 8 weight fragments (8 KiB per wave, contiguous); they are loaded two stages ahead into a three-deep
    // register ring (96 registers instead of the 128 a chunk-deep double buffer would take next to the 256
    // accumulators).
    f16x8 A[3][2][4];  // [ring slot][nu & 1][m]
    auto load_a = [&](int stage, auto slotc) {
        constexpr int slot = decltype(slotc)::value;
        const unsigned char* src = gw + (size_t)(stage >> 1) * W_CHUNK + (stage & 1) * 8 * 1024;  // uniform
#if defined(SAYURI_WINO_ABL) && (SAYURI_WINO_ABL & 4)  // timing only: the weight ring is filled once
        if (stage > 2) {
            // keep the hand-counted vmcnt consistent: eight cheap loads of one cached line
#pragma unroll
            for (int q = 0; q < 8; ++q) gload16<0>(A[slot][q >> 2][q & 3], gw, 0u);
            return;
        }
#endif
        gload16<0>(A[slot][0][0], src, wlane);
        gload16<1024>(A[slot][0][1], src, wlane);
        gload16<2048>(A[slot][0][2], src, wlane);
        gload16<3072>(A[slot][0][3], src, wlane);
        gload16<0>(A[slot][1][0], src + 4096, wlane);
        gload16<1024>(A[slot][1][1], src + 4096, wlane);
        gload16<2048>(A[slot][1][2], src + 4096, wlane);
        gload16<3072>(A[slot][1][3], src + 4096, wlane);
    };
    constexpr int nstages = 2 * nchunks;

    // Flat software pipeline over the items k = (stage, 16-tile column n): while the MFMAs of item k run, the
    // patch fragments of item k+1 (read one item earlier) are transformed and the reads of item k+2 go out.  The
    // chunk boundary work (DMA wait, barrier, next DMA) sits in front of the first read of a new chunk, i.e. two
    // items before its first MFMA.
    constexpr int nitems = nstages * NF;
    // -1 the optimiser cannot see through: a visible constant turns a + (-1)*b back into a vector fsub, which this
    // target expands to three scalar-half instructions per register instead of one v_pk_fma_f16
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    uint32_t negbits = 0xBC00BC00u;
    asm volatile("" : "+s"(negbits));
    const f16x8 neg8 = __builtin_bit_cast(f16x8, (u32x4){negbits, negbits, negbits, negbits});
    f16x8 R[2][6];  // patch fragments of item j live in R[j & 1]; reads run TWO items ahead of their transform
    auto read_patch = [&](auto kc) {
        constexpr int k = decltype(kc)::value;
        constexpr int stage = k / NF, n = k % NF, H = stage & 1, RB = (stage >> 1) & 1, B = k & 1;
        constexpr int SO = RB * Cfg::RAW_BYTES;
#if defined(SAYURI_WINO_ABL) && (SAYURI_WINO_ABL & 2)  // timing only: patch fragments read once per kernel
        if (k > 2) { asm volatile("" : "+v"(R[B][0]), "+v"(R[B][1]), "+v"(R[B][2]), "+v"(R[B][3]), "+v"(R[B][4]), "+v"(R[B][5])); return; }
#endif
        // patch columns j = H .. H+2: cells (even tx, odd tx, even tx+1) or (odd tx, even tx+1, odd tx+1)
        ds_read16<SO>(R[B][0], cell[n][0][H + 0]);
        ds_read16<SO>(R[B][1], cell[n][0][H + 1]);
        ds_read16<SO>(R[B][2], cell[n][0][H + 2]);
        ds_read16<SO>(R[B][3], cell[n][1][H + 0]);
        ds_read16<SO>(R[B][4], cell[n][1][H + 1]);
        ds_read16<SO>(R[B][5], cell[n][1][H + 2]);
    };
    // t[j] = d[ia][j] + sgn * d[ib][j];  V[nu] = (t0 - t2, t1 + t2, t2 - t1, t1 - t3); a stage holds nu = 2H, 2H+1
    auto transform = [&](auto hc, f16x8 (&Rb)[6], f16x8 (&V)[2]) {
        constexpr int H = decltype(hc)::value;
#if defined(SAYURI_WINO_ABL) && (SAYURI_WINO_ABL & 1)  // timing only: no transform arithmetic
        V[0] = Rb[0]; V[1] = Rb[4];
        asm volatile("" : "+v"(Rb[1]), "+v"(Rb[2]), "+v"(Rb[3]), "+v"(Rb[5]));
        return;
#endif
        const f16x8 ta = Rb[0] + sgn8 * Rb[3], tb = Rb[1] + sgn8 * Rb[4], tc = Rb[2] + sgn8 * Rb[5];
        if constexpr (H == 0) { V[0] = ta + neg8 * tc; V[1] = tb + tc; }   // ta, tb, tc = t0, t1, t2
        else { V[0] = tb + neg8 * ta; V[1] = ta + neg8 * tc; }             // ta, tb, tc = t1, t2, t3
    };
    using Order = WinoOrder<NCH, NF, BI>;
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;

    issue_raw(0, 0);
    load_a(0, I0{});
    load_a(1, I1{});
    if (dbg) dbg[1] = __builtin_amdgcn_s_memtime();
    wait_vmcnt<16>();  // raw(0) landed; the two weight batches may still be in flight
    __builtin_amdgcn_s_barrier();
    if (dbg) dbg[2] = __builtin_amdgcn_s_memtime();
    if constexpr (nchunks > 1) issue_raw(1, 1);
    f16x8 Vc[2], Vn[2];
    unsigned long long wait_a_cycles = 0;
    read_patch(I0{});
    read_patch(I1{});
    wait_patch<6>(R[0]);
    transform(I0{}, R[0], Vc);
    read_patch(I2{});

    static_for<nitems>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        constexpr int stage = k / NF, n = k % NF, SLOT = stage % 3, H = stage & 1;
        if constexpr (k + 1 < nitems) {
            // patch(k+1) was read two items ago; only patch(k+2) is younger
            wait_patch<(k + 2 < nitems ? 6 : 0)>(R[(k + 1) & 1]);
            transform(std::integral_constant<int, ((k + 1) / NF) & 1>{}, R[(k + 1) & 1], Vn);
        }
        if constexpr (k + 3 < nitems) {
            if constexpr ((k + 3) % (2 * NF) == 0) {
                constexpr int chunk2 = (k + 3) / (2 * NF);
                constexpr int young = Order::after(0, chunk2, k, true);
                if constexpr (chunk2 == 4) { if (dbg) dbg[13] = __builtin_amdgcn_s_memtime(); }
                // this wave's reads of chunk2-1 are done (the DMA below overwrites that slot), raw(chunk2) has landed
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(young) : "memory");
                if constexpr (chunk2 == 4) { if (dbg) dbg[14] = __builtin_amdgcn_s_memtime(); }
                __builtin_amdgcn_s_barrier();
                if constexpr (chunk2 == 4) { if (dbg) dbg[15] = __builtin_amdgcn_s_memtime(); }
                if constexpr (chunk2 + 1 < nchunks) issue_raw(chunk2 + 1, (chunk2 + 1) & 1);
                if constexpr (chunk2 < 7) { if (dbg) dbg[2 + chunk2] = __builtin_amdgcn_s_memtime(); }
            }
            read_patch(std::integral_constant<int, k + 3>{});
        }
        if constexpr (n == 0 && stage + 2 < nstages) load_a(stage + 2, std::integral_constant<int, (SLOT + 2) % 3>{});
        if constexpr (n == 0) {
            unsigned long long t0 = 0;
            if (dbg) t0 = __builtin_amdgcn_s_memtime();
            wait_a<Order::after(1, stage, k, false)>(A[SLOT]);
            if (dbg) wait_a_cycles += __builtin_amdgcn_s_memtime() - t0;
        }
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int m = 0; m < 4; ++m)
                acc[2 * H + v][m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[SLOT][v][m], Vc[v], acc[2 * H + v][m][n], 0, 0, 0);
        if constexpr (k + 1 < nitems) { Vc[0] = Vn[0]; Vc[1] = Vn[1]; }
    });

    // ---- output transform.  nu half in registers: Z[b] = M A, A^T = [[1,1,1,0],[0,1,-1,-1]]; xi half through LDS.
    auto lds_barrier = [] {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    if (dbg) { dbg[10] = __builtin_amdgcn_s_memtime(); dbg[9] = dbg[0] + wait_a_cycles; }
    lds_barrier();  // raw ring no longer read
    unsigned char* stage = smem;
    constexpr int RS = Cfg::Z_RS;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int kol = m * 16 + 4 * kg;
#pragma unroll
        for (int n = 0; n < NF; ++n) {
            const int tile = n * 16 + (lane & 15);
            const f32x4 z0 = acc[0][m][n] + acc[1][m][n] + acc[2][m][n];
            const f32x4 z1 = acc[1][m][n] - acc[2][m][n] - acc[3][m][n];
            *(f32x4*)(stage + ((size_t)((xi * 2 + 0) * NTL + tile) * RS) + kol * 4) = z0;
            *(f32x4*)(stage + ((size_t)((xi * 2 + 1) * NTL + tile) * RS) + kol * 4) = z1;
        }
    }
    lds_barrier();
    if (dbg) dbg[11] = __builtin_amdgcn_s_memtime();
    switch (p.act) {
    case kMish: wino_store<kMish>(wp, stage, out_row, kt, tid); break;
    case kIdentity: wino_store<kIdentity>(wp, stage, out_row, kt, tid); break;
    case kReLU: wino_store<kReLU>(wp, stage, out_row, kt, tid); break;
    case kSwish: wino_store<kSwish>(wp, stage, out_row, kt, tid); break;
    case kELU: wino_store<kELU>(wp, stage, out_row, kt, tid); break;
    case kSELU: wino_store<kSELU>(wp, stage, out_row, kt, tid); break;
    case kGELU: wino_store<kGELU>(wp, stage, out_row, kt, tid); break;
    default: wino_store<kHardSwish>(wp, stage, out_row, kt, tid); break;
    }
    if (dbg) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); dbg[12] = __builtin_amdgcn_s_memtime(); }
}

// ---------------------------------------------------------------------------------------------------------------
// Eight-wave variant: two waves per SIMD, so that one wave's transform arithmetic, LDS waits and epilogue overlap the
// other's MFMAs (the four-wave kernel above spends half of its life issuing VALU instructions with nothing else on
// the SIMD: DESIGN.md "Kernel 3").  Wave (xi, h) owns transform row xi and the nu PAIR h -- (0,1) or (3,2) -- for all 64
// output channels and 48 tiles: 96 accumulators, its own eight weight fragments per chunk (no weight is loaded
// twice), a two-chunk weight ring.  With the patch columns of pair 1 taken in the order j = 3,2,1 both pairs use one
// formula, Vx = ta - tc, Vy = tb + sy*tc (sy = +1 / -1); pair 1 then accumulates -M[xi][3], which the output transform
// undoes.
struct Wino8Cfg {
    static constexpr int KO_T = 64, NF = 3, NTL = 16 * NF, NWAVE = 8, NT = 512;
    static constexpr int NPOS = WinoCfg::NPOS, RAW_BYTES = NPOS * 64;
    static constexpr int Z_RS = KO_T * 4 + 16;
    static constexpr int STAGE_BYTES = 8 * NTL * Z_RS;     // one b at a time: [wave][tile][ch]
    static constexpr int OUTROW_OFF = STAGE_BYTES > 2 * RAW_BYTES ? STAGE_BYTES : 2 * RAW_BYTES;
    static constexpr size_t lds_bytes() { return OUTROW_OFF + NTL * 4 * 4; }
    static_assert(NTL == WinoCfg::NTL, "both kernels share the block tables");
};

template <int ACT>
__device__ __forceinline__ void wino8_store(const WinoParams& wp, const unsigned char* stage, const int* out_row, int kt, int tid,
                                            int b, const f32x4& bias0, const f32x4& bias1) {
    constexpr int NTL = Wino8Cfg::NTL, RS = Wino8Cfg::Z_RS, SLAB = NTL * RS;
    const ConvParams& p = wp.c;
    f16* __restrict__ gout = (f16*)p.out;
    const f16* __restrict__ gres = (const f16*)p.res;
    const int cg = tid & 7;
    const int ko = kt * Wino8Cfg::KO_T + cg * 8;
    // items (tile, a) x 8 channel groups = 768 per phase: two rounds of 512 threads, the second half full
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int it = (k * 512 + tid) >> 3;  // tile*2 + a
        if (it >= NTL * 2) break;
        const int tile = it >> 1, a = it & 1;
        const int row = out_row[tile * 4 + a * 2 + b];
        f16x8 rr = {0, 0, 0, 0, 0, 0, 0, 0};
        if (gres && row >= 0) rr = *(const f16x8*)(gres + (size_t)row * p.cout_s + ko);
        // Y[a][b] = sum_xi At[a][xi] (P[xi,0][b] + P[xi,1][b]);  At = [[1,1,1,0],[0,1,-1,-1]]
        const unsigned char* z = stage + (size_t)tile * RS + cg * 32;
        f32x4 v0 = bias0, v1 = bias1;
#pragma unroll
        for (int x = 0; x < 3; ++x) {
            const int xi = a ? x + 1 : x;
            const float sg = (a && x > 0) ? -1.f : 1.f;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const unsigned char* zs = z + (size_t)(xi * 2 + h) * SLAB;
                v0 += sg * *(const f32x4*)zs;
                v1 += sg * *(const f32x4*)(zs + 16);
            }
        }
        float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        if (gres) {
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] += (float)rr[q];
        }
        f16x8 hh;
#pragma unroll
        for (int q = 0; q < 8; ++q) hh[q] = (f16)activate(v[q], ACT);
        if (row >= 0) *(f16x8*)(gout + (size_t)row * p.cout_s + ko) = hh;
    }
}

// NCH = 32-channel chunks (fully unrolled), DI = DMA instructions per wave and chunk (3: up to 384 raw positions, 4: 512)
template <int NCH, int DI>
__global__ __launch_bounds__(512) void conv_wino8_kernel(const WinoParams wp) {
    using Cfg = Wino8Cfg;
    constexpr int NPOS = Cfg::NPOS, NTL = Cfg::NTL, NF = Cfg::NF;
    const ConvParams& p = wp.c;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xi = wave >> 1, H = wave & 1;
    const int kts = p.ko_pad / Cfg::KO_T;
    const int xcd = blockIdx.x & 7, rest = blockIdx.x >> 3;
    const int kt = rest % kts;
    const int blk = (rest / kts) * 8 + xcd;
    if (blk >= wp.num_blocks) return;
    unsigned long long* dbg = (wp.dbg && lane == 0 && blockIdx.x < 64) ? wp.dbg + ((size_t)blockIdx.x * 4 + (wave & 3)) * 16 : nullptr;
    if (dbg && wave < 4) dbg[0] = __builtin_amdgcn_s_memtime();

    // ---- DMA role: instruction q = wave + 8*i moves positions 16q .. 16q+15 (see conv_wino_kernel)
    const unsigned char* gin0 = (const unsigned char*)p.in - kZeroPrefix;
    uint32_t voff[DI];
#pragma unroll
    for (int i = 0; i < DI; ++i) {
        const int src = wp.tab_src[(size_t)blk * NPOS + (wave + 8 * i) * 16 + (lane >> 2)];
        const uint32_t kq = (uint32_t)((lane & 3) - (src >> 28)) & 3u;
        voff[i] = src >= 0 ? (uint32_t)kZeroPrefix + (uint32_t)(src & 0x0fffffff) * (uint32_t)(p.cin_s * 2) + kq * 16u : 0u;
    }
    int* out_row = (int*)(smem + Cfg::OUTROW_OFF);
    if (tid < NTL * 4) out_row[tid] = wp.tab_out[(size_t)blk * NTL * 4 + tid];

    // ---- patch cells: rows ia / ib of B^T's row xi; columns j = 0,1,2 (pair 0) or 3,2,1 (pair 1)
    const int kg = lane >> 4;
    const int ia = xi == 0 ? 0 : (xi == 2 ? 2 : 1);
    const int ib = xi == 0 ? 2 : (xi == 1 ? 2 : (xi == 2 ? 1 : 3));
    uint32_t cell[NF][2][3];
#pragma unroll
    for (int n = 0; n < NF; ++n) {
        const int pk = wp.tab_tile[(size_t)blk * NTL + n * 16 + (lane & 15)];
        const int lpos = pk & 0xffff, pitch = (pk >> 16) & 0xff, trl = pk >> 24, half = (pitch - 1) >> 1;
#pragma unroll
        for (int ab = 0; ab < 2; ++ab) {
            const int i = ab ? ib : ia, row = 2 * trl + i;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int j = H ? 3 - c : c;  // patch column; cell = even/odd half (j & 1), tile column tx + (j >> 1)
                const int pos = lpos + i * pitch + (j & 1) * half + (j >> 1);
                cell[n][ab][c] = (uint32_t)(uintptr_t)smem + (uint32_t)(pos * 64 + ((kg + wino_rot(pos, row)) & 3) * 16);
            }
        }
    }
    const f16 sgn = xi == 1 ? (f16)1.f : (f16)-1.f;
    const f16x8 sgn8 = {sgn, sgn, sgn, sgn, sgn, sgn, sgn, sgn};
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    uint32_t negbits = 0xBC00BC00u, sybits = H ? 0xBC00BC00u : 0x3C003C00u;
    asm volatile("" : "+s"(negbits), "+s"(sybits));
    const f16x8 neg8 = __builtin_bit_cast(f16x8, (u32x4){negbits, negbits, negbits, negbits});
    const f16x8 sy8 = __builtin_bit_cast(f16x8, (u32x4){sybits, sybits, sybits, sybits});

    constexpr int nchunks = NCH;
    // weights [kt][chunk][xi][nu][m]: this wave's Vx pairs with nu = 0 / 3, Vy with nu = 1 / 2
    const unsigned char* gw = (const unsigned char*)p.w + ((size_t)kt * nchunks * 4 + xi) * 16 * 1024;
    const int nux = H ? 3 : 0, nuy = H ? 2 : 1;
    const uint32_t wlane = (uint32_t)lane * 16u;
    constexpr size_t W_CHUNK = 4 * 16 * 1024;

    auto issue_raw = [&](int chunk, int slot) {
        const unsigned char* gb = gin0 + chunk * (kChunk * 2);
#pragma unroll
        for (int i = 0; i < DI; ++i) glds16(gb + voff[i], smem + slot * Cfg::RAW_BYTES + (wave + 8 * i) * 1024);
    };
    f32x4 acc[2][4][NF];  // [x / y][m][n]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int c = 0; c < NF; ++c) acc[a][b][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    f16x8 A[2][2][4];  // [ring slot][x / y][m]
    auto load_a = [&](int chunk, auto slotc) {
        constexpr int slot = decltype(slotc)::value;
        const unsigned char* sx = gw + (size_t)chunk * W_CHUNK + nux * 4096;
        const unsigned char* sy = gw + (size_t)chunk * W_CHUNK + nuy * 4096;
        gload16<0>(A[slot][0][0], sx, wlane);
        gload16<1024>(A[slot][0][1], sx, wlane);
        gload16<2048>(A[slot][0][2], sx, wlane);
        gload16<3072>(A[slot][0][3], sx, wlane);
        gload16<0>(A[slot][1][0], sy, wlane);
        gload16<1024>(A[slot][1][1], sy, wlane);
        gload16<2048>(A[slot][1][2], sy, wlane);
        gload16<3072>(A[slot][1][3], sy, wlane);
    };
    f16x8 R[6];
    auto read_patch = [&](auto slotc, int n) {
        constexpr int SO = decltype(slotc)::value * Cfg::RAW_BYTES;
        ds_read16<SO>(R[0], cell[n][0][0]);
        ds_read16<SO>(R[1], cell[n][0][1]);
        ds_read16<SO>(R[2], cell[n][0][2]);
        ds_read16<SO>(R[3], cell[n][1][0]);
        ds_read16<SO>(R[4], cell[n][1][1]);
        ds_read16<SO>(R[5], cell[n][1][2]);
    };

    unsigned long long w_dma = 0, w_bar = 0, w_issue = 0, w_a = 0, w_patch = 0;
    issue_raw(0, 0);
    load_a(0, std::integral_constant<int, 0>{});
    static_for<nchunks>([&](auto cc) {
        constexpr int c = decltype(cc)::value, S = c & 1;
        if constexpr (c == 0) { if (dbg && wave < 4) dbg[1] = __builtin_amdgcn_s_memtime(); }
        unsigned long long tq0 = 0, tq1 = 0, tq2 = 0, tq3 = 0;
        if (dbg) tq0 = __builtin_amdgcn_s_memtime();
        wait_vmcnt<8>();               // raw(c) landed; the weights of chunk c were issued after it
        if (dbg) tq1 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_barrier();  // every wave's pieces are in LDS; every wave is done with the other ring slot
        if (dbg) tq2 = __builtin_amdgcn_s_memtime();
        if constexpr (c < 8) { if (dbg && wave < 4) dbg[2 + c] = __builtin_amdgcn_s_memtime(); }
        if constexpr (c + 1 < nchunks) {
            issue_raw(c + 1, S ^ 1);
            load_a(c + 1, std::integral_constant<int, S ^ 1>{});
        }
        if (dbg) tq3 = __builtin_amdgcn_s_memtime();
        wait_a<(c + 1 < nchunks ? DI + 8 : 0)>(A[S]);
        if (dbg) { const unsigned long long t4 = __builtin_amdgcn_s_memtime(); w_dma += tq1 - tq0; w_bar += tq2 - tq1; w_issue += tq3 - tq2; w_a += t4 - tq3; }
        read_patch(std::integral_constant<int, S>{}, 0);
#pragma unroll
        for (int n = 0; n < NF; ++n) {
            unsigned long long tp = 0;
            if (dbg) tp = __builtin_amdgcn_s_memtime();
            wait_patch<0>(R);
            if (dbg) w_patch += __builtin_amdgcn_s_memtime() - tp;
            // t[j] = d[ia][j] + sgn * d[ib][j] for this pair's three columns; Vx = ta - tc, Vy = tb + sy * tc
            const f16x8 ta = R[0] + sgn8 * R[3], tb = R[1] + sgn8 * R[4], tc = R[2] + sgn8 * R[5];
            f16x8 V[2];
            V[0] = ta + neg8 * tc;
            V[1] = tb + sy8 * tc;
            if (n + 1 < NF) read_patch(std::integral_constant<int, S>{}, n + 1);
#pragma unroll
            for (int v = 0; v < 2; ++v)
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    acc[v][m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[S][v][m], V[v], acc[v][m][n], 0, 0, 0);
        }
    });

    // ---- output transform.  Pair 0 holds (M0, M1), pair 1 holds (-M3, M2).  Z[b] = M A with A^T = [[1,1,1,0],[0,1,-1,-1]]:
    // partial sums P[b=0] = M0 + M1 | M2,  P[b=1] = M1 | -M2 - M3; the xi sum and the pair sum go through LDS, one b at a time.
    auto lds_barrier = [] {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    if (dbg && wave < 4) { dbg[10] = __builtin_amdgcn_s_memtime(); dbg[9] = w_dma; dbg[13] = w_bar; dbg[14] = w_issue; dbg[15] = w_a; dbg[8] = w_patch; }
    unsigned char* stage = smem;
    constexpr int RS = Cfg::Z_RS;
    const int cgo = tid & 7, koo = kt * Cfg::KO_T + cgo * 8;
    const bool ko_ok = koo < p.cout_s;  // cout_s is a multiple of 32: 8-channel groups never straddle it
    f32x4 bias0 = {0.f, 0.f, 0.f, 0.f}, bias1 = bias0;
    if (ko_ok) { bias0 = *(const f32x4*)(p.bias + koo); bias1 = *(const f32x4*)(p.bias + koo + 4); }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        lds_barrier();  // raw ring / previous phase no longer read
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int kol = m * 16 + 4 * kg;
#pragma unroll
            for (int n = 0; n < NF; ++n) {
                const int tile = n * 16 + (lane & 15);
                f32x4 z;
                if (b == 0) z = H ? acc[1][m][n] : acc[0][m][n] + acc[1][m][n];
                else z = H ? acc[0][m][n] - acc[1][m][n] : acc[1][m][n];
                *(f32x4*)(stage + ((size_t)(wave * NTL + tile) * RS) + kol * 4) = z;
            }
        }
        lds_barrier();
        if (b == 1) { if (dbg && wave < 4) dbg[11] = __builtin_amdgcn_s_memtime(); }
        if (ko_ok) {
            switch (p.act) {
            case kMish: wino8_store<kMish>(wp, stage, out_row, kt, tid, b, bias0, bias1); break;
            case kIdentity: wino8_store<kIdentity>(wp, stage, out_row, kt, tid, b, bias0, bias1); break;
            case kReLU: wino8_store<kReLU>(wp, stage, out_row, kt, tid, b, bias0, bias1); break;
            case kSwish: wino8_store<kSwish>(wp, stage, out_row, kt, tid, b, bias0, bias1); break;
            case kELU: wino8_store<kELU>(wp, stage, out_row, kt, tid, b, bias0, bias1); break;
            case kSELU: wino8_store<kSELU>(wp, stage, out_row, kt, tid, b, bias0, bias1); break;
            case kGELU: wino8_store<kGELU>(wp, stage, out_row, kt, tid, b, bias0, bias1); break;
            default: wino8_store<kHardSwish>(wp, stage, out_row, kt, tid, b, bias0, bias1); break;
            }
        }
    }
    if (dbg && wave < 4) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); dbg[12] = __builtin_amdgcn_s_memtime(); }
}

}  // namespace sayuri
