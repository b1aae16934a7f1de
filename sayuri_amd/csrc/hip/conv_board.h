// conv_board.h -- the tower convolution (fp16 3x3 implicit GEMM) with ONE WORKGROUP PER BOARD.
//
// A Go board is small: 19 x 19 = 361 pixels x 256 channels of fp32 accumulators is 370 KB, which fits the 512 KB
// register file of one CU.  So a workgroup owns whole samples (one 19x19 board, two 13x13, four 9x9 ...: up to 384
// pixels = 24 column tiles of 16) and ALL output channels of the layer (KO_T = 256 / 192 / 128):
//   * a 256-sample batch of 19x19 boards is exactly 256 workgroups = one full wave of the chip's 256 CUs, no tail
//     (conv_glds.h's 192-pixel tiles gave 482 workgroups = 1.88 waves);
//   * the halo of a tile is the boards' own zero frame -- no neighbouring tile is ever read, and a layer's output
//     tile is exactly the next layer's input tile;
//   * every weight byte is staged once per board (half the L2 -> LDS weight traffic of the 192-pixel tiles) and a
//     K group (3 taps x 32 channels) is 4.4 k MFMA cycles per SIMD between workgroup barriers instead of 2.3 k;
//   * the workgroup sees every pixel and every channel of its samples, so the squeeze-and-excitation unit that
//     follows a block's second convolution (global pooling -> FC -> FC -> scale + residual + activation,
//     reference se_unit.cc:70-128) runs INSIDE this kernel's epilogue on the accumulators (conv_board_se.h).
//
// Waves: 4 along M x 2 along N.  Wave (m, n) owns output channels [m*WMT*16, (m+1)*WMT*16) and the column tiles
// [col0, col0 + nj) of the tile, nj = 12 or 11 for a 19x19 board: waves w and w + 4 share a SIMD, so each SIMD gets
// 23 column tiles x WMT row tiles -- balanced without padding the 23rd tile to a 24th.
// K loop, LDS rings and the hand-counted ds_read_b128 stream follow conv_glds.h: weights and halo reach LDS by
// global_load_lds_dwordx4 only, one s_barrier per K group, fragment reads are inline asm with compile-time lgkmcnt.
// Epilogue: no LDS staging.  v_permlane16_swap pairs the 4-channel accumulator quads of two row tiles into 8
// consecutive channels per lane (16 bytes of fp16), four lanes cover 64 contiguous bytes of one NHWC pixel row.
#pragma once
#include "common.h"
#include "conv_glds.h"
#include "conv_mfma.h"
#include "small_ops.h"

namespace sayuri {

constexpr int kBoardCols = 24;              // 16-pixel column tiles per workgroup
constexpr int kBoardPT = kBoardCols * 16;   // pixel slots per tile
constexpr int kBoardMaxPos = 512;           // halo positions per tile (DMA blocks of 64)
constexpr int kBoardMaxSub = 32;            // samples per tile
constexpr int kBoardNJ = kBoardCols / 2;    // column tiles per wave

// Row order of the weight image for the GENERATED epilogue of the persistent launch (tower_seam.py epi_hook).  A lane of the
// 16x16 accumulator layout holds rows 4q .. 4q+3 (q = lane >> 4) of each 16-row tile; a 16-byte store wants 8 CONSECUTIVE
// channels per lane.  The compiled epilogue pairs the quads of two row tiles with four v_permlane16_swap per store -- 14 SIMD
// cycles each, a seventh of the epilogue's VALU time (tools/ubench/trans_rate.hip).  With the image's rows permuted inside every
// group of 32 (= a pair of row tiles; a wave's rows start at a multiple of 32 for the even tile counts) -- row (tile t, m)
// carries channel 8 (m >> 2) + 4 t + (m & 3) -- lane q of the pair holds channels 8q .. 8q+3 (tile 0) and 8q+4 .. 8q+7 (tile 1)
// without any exchange.  The bias travels in the same order (the main loop indexes it by row).  BoardParams::row_order says
// which image a layer was given; only the generated code reads it.
__host__ __device__ constexpr int board_row_channel(int row) {
    return (row & ~31) + 8 * ((row & 15) >> 2) + 4 * ((row >> 4) & 1) + (row & 3);
}

struct BoardParams {
    ConvParams c;          // num_pix_tiles = number of board tiles; in must carry the kZeroPrefix zero bytes in front
    const int* tab_src;    // [tile][npos]   activation row feeding each halo position, -1 = zero
    const int2* tab_pix;   // [tile][384]    x = lpos | lstr << 16, y = output activation row (-1 = none)
    const int* tab_cols;   // [tile]         column tiles in use (1..24) | board size << 8
    int npos;              // halo positions per tile of this launch (multiple of 64, <= 512)
    int uniform_info;      // >= 0: every tile's tab_cols entry has this value (a uniform batch: one table read less)
    unsigned long long* dbg;  // DBG kernels only: s_memtime timeline [workgroup < 4][wave][8]
    int arith;                // 1: every tile is ONE sample and all samples have one size (tile = sample): the kernels compute
                              // the table entries they need (a few VALU ops) instead of waiting for them at the head of the
                              // prologue and of the epilogue
    // persistent tower launch only (conv_tower.h, CHAIN main loop): w_next = the weights of the run's next layer, whose first
    // group this layer's last K group brings into ring slot 0 (null: no hand-over); w_ready = 1: the previous layer did that
    // for this one, the prologue does not ask for group 0 again
    const void* w_next;
    int w_ready;
    int row_order;  // persistent tower launch only: 1 = weights and bias are in board_row_channel order (the generated epilogue runs)
};

// Which samples share a tile: consecutive samples OF ONE BOARD SIZE, greedily, while pixels <= 384, halo positions
// <= 512 and samples <= 32 (one size per tile keeps the halo row pitch wave-uniform: the B-fragment addresses of a
// kernel row are then the previous row's plus a scalar).  The same scan runs on the host (grid size, LDS size) and
// in board_setup_kernel.
struct BoardPack {
    int px = 0, pos = 0, cnt = 0, bs0 = 0;
    __host__ __device__ bool fits(int bs) const {
        return (cnt == 0 || bs == bs0) && cnt < kBoardMaxSub && px + bs * bs <= kBoardPT &&
               pos + (bs + 2) * (bs + 2) <= kBoardMaxPos;
    }
    __host__ __device__ void add(int bs) { px += bs * bs; pos += (bs + 2) * (bs + 2); ++cnt; bs0 = bs; }
};

// Index tables of one batch geometry, one workgroup per tile (shared by every layer of the forward).
__global__ __launch_bounds__(256) void board_setup_kernel(BatchGeom g, int npos, int* __restrict__ tab_src,
                                                          int2* __restrict__ tab_pix, int* __restrict__ tab_cols) {
    __shared__ int sub[kBoardMaxSub][4];  // {first halo position, first pixel slot, board size, sample}
    __shared__ int hdr[2];                // {samples, pixels}
    const int tile = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) {
        int t = 0, s = 0;
        BoardPack pk;
        int cnt = 0;
        for (; s < g.n_samples; ++s) {
            const int bs = g.bsz[s];
            if (pk.cnt > 0 && !pk.fits(bs)) {
                if (t == tile) break;
                ++t;
                pk = BoardPack{};
            }
            if (t == tile) {
                sub[cnt][0] = pk.pos; sub[cnt][1] = pk.px; sub[cnt][2] = bs; sub[cnt][3] = s;
                ++cnt;
            }
            pk.add(bs);
        }
        hdr[0] = cnt;
        hdr[1] = cnt > 0 ? sub[cnt - 1][1] + sub[cnt - 1][2] * sub[cnt - 1][2] : 0;
        tab_cols[tile] = ((hdr[1] + 15) / 16) | (sub[0][2] << 8);
    }
    __syncthreads();
    const int nsub = hdr[0], npx = hdr[1];
    for (int pos = tid; pos < npos; pos += blockDim.x) {
        int src = -1;
        for (int s = 0; s < nsub; ++s) {
            const int bs = sub[s][2], w2 = bs + 2, rel = pos - sub[s][0];
            if (rel >= 0 && rel < w2 * w2) {
                const int r = rel / w2, xc = rel - r * w2;
                const int y = r - 1, x = xc - 1;
                if (y >= 0 && y < bs && x >= 0 && x < bs) src = sub[s][3] * g.slot_pix + y * bs + x;
                break;
            }
        }
        tab_src[(size_t)tile * npos + pos] = src;
    }
    for (int i = tid; i < kBoardPT; i += blockDim.x) {
        int lstr = sub[0][2] + 2, lpos = lstr + 1, orow = -1;  // an interior cell: every tap stays inside the tile
        if (i < npx) {
            for (int s = 0; s < nsub; ++s) {
                const int bs = sub[s][2], pp = i - sub[s][1];
                if (pp >= 0 && pp < bs * bs) {
                    const int y = pp / bs, x = pp - y * bs;
                    lstr = bs + 2;
                    lpos = sub[s][0] + (y + 1) * (bs + 2) + x + 1;
                    orow = sub[s][3] * g.slot_pix + pp;
                    break;
                }
            }
        }
        tab_pix[(size_t)tile * kBoardPT + i] = make_int2(lpos | (lstr << 16), orow);
    }
}

template <int WMT_> struct BoardCfg {
    static constexpr int WMT = WMT_, WAVM = 4, WAVN = 2, NWAVE = 8, NJ = kBoardNJ;
    static constexpr int KO_T = WAVM * WMT * 16;
    static constexpr int A_TAP_BYTES = KO_T * 64;   // one tap x 32 channels of weights
    static constexpr int A_BYTES = 3 * A_TAP_BYTES; // one K group
    static constexpr int A_INSTR = KO_T / 16;       // 1 KiB DMA instructions per tap tile
    static constexpr int AI = (A_INSTR + NWAVE - 1) / NWAVE;
    static constexpr int KO_PARTS = KO_T / 64;      // 64-row pieces per k-group plane
    // K-loop rings; the launch always asks for the whole 160 KiB (one workgroup per CU either way): the epilogue hands
    // each wave 20 KiB of it for its residual rows
    static constexpr size_t ring_bytes(int npos) { return 2 * (size_t)A_BYTES + 2 * (size_t)npos * 64; }
    static constexpr size_t lds_bytes(int npos) { return ring_bytes(npos) <= 160 * 1024 ? 160 * 1024 : ring_bytes(npos); }
};

// Measuring builds only (tools/gpu/lds_streams.sh): -DSAYURI_DROP_STREAM=1 leaves out the K loop's A-fragment reads, =2 its
// B-fragment reads (the MFMAs then multiply stale registers: wrong results, same instruction stream otherwise), so that the
// LDS counters of the two streams can be told apart.
#ifndef SAYURI_DROP_STREAM
#define SAYURI_DROP_STREAM 0
#endif
template <int OFF> __device__ __forceinline__ void board_read_a(f16x8& dst, uint32_t addr) {
    if constexpr (SAYURI_DROP_STREAM == 1) asm volatile("" : "+v"(dst) : "v"(addr));
    else ds_read16<OFF>(dst, addr);
}
template <int OFF> __device__ __forceinline__ void board_read_b(f16x8& dst, uint32_t addr) {
    if constexpr (SAYURI_DROP_STREAM == 2) asm volatile("" : "+v"(dst) : "v"(addr));
    else ds_read16<OFF>(dst, addr);
}

namespace board_sched {
// The fragment stream of one K group: MFMA block b = 12*dx + j (tap-in-row dx, column tile j) issues WMT MFMAs
// with A(dx, 0..WMT-1) and B(b).  LDS reads in program order:
//   before block 0:  B(0), A(0,0) .. A(0,WMT-1), B(1)
//   block b:         B(b+2) | wait | MFMA 0 .. WMT-1; in the last block of a tap (j == 11) A(dx+1, i) follows MFMA i
// A fragment i always lives in afr[i] (the next tap's fragment lands in the register its MFMA has just read),
// B fragments in a ring of 3.
constexpr int kNB = 3 * kBoardNJ;
template <int WMT> constexpr int late_a(int b) { return (b >= 0 && b % kBoardNJ == kBoardNJ - 1 && b / kBoardNJ < 2) ? WMT : 0; }
// LDS reads younger than B(b) when block b waits for it (the first block of a tap waits per MFMA instead)
template <int WMT> constexpr int young_b(int b) {
    if (b < 2) return 2;  // block 0 waits for A(0, WMT-1): B(1), B(2) are younger; block 1: B(2), B(3)
    return late_a<WMT>(b - 2) + (b + 1 < kNB ? 1 : 0) + late_a<WMT>(b - 1) + (b + 2 < kNB ? 1 : 0);
}
}  // namespace board_sched

template <int N> __device__ __forceinline__ void wait_lgkm1(f16x8& a) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N));
}
template <int N> __device__ __forceinline__ void wait_lgkm2(f16x8& a, f16x8& b) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}
// the first NF fragments of the A ring and one B fragment
template <int N, int NF, int NA> __device__ __forceinline__ void wait_lgkm_all(f16x8 (&a)[NA], f16x8& b) {
    static_assert(NF >= 2 && NF <= 4 && NF <= NA, "2..4 A fragments per tap");
    if constexpr (NF == 2) asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a[0]), "+v"(a[1]), "+v"(b) : "n"(N));
    else if constexpr (NF == 3) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(b) : "n"(N));
    else asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b) : "n"(N));
}

__device__ __forceinline__ void swap16(f32x4& a, f32x4& b) {
    // per register, rows of 16 lanes: a's odd rows <-> b's even rows.  Inline asm: with hipcc (ROCm 7.2)
    // __builtin_amdgcn_permlane16_swap loses its second result here (the code that follows reads the first result for
    // both halves -- seen in the .s, and as wrong channels 4-7 / 12-15 of every row tile on the GPU); s_nop 1 = the
    // two wait states a VALU write of either operand needs before the swap reads it.
    float a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3], b0 = b[0], b1 = b[1], b2 = b[2], b3 = b[3];
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %4\n\tv_permlane16_swap_b32 %1, %5\n\tv_permlane16_swap_b32 %2, %6\n\t"
        "v_permlane16_swap_b32 %3, %7"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));
    a = f32x4{a0, a1, a2, a3};
    b = f32x4{b0, b1, b2, b3};
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
// Mish on two values with packed fp32 arithmetic (v_pk_mul / v_pk_add / v_pk_fma): x * (1 - 2 / (e^2 + 2e + 2)).
// Same values as activate(x, kMish): its x > 20 guard only fixes the code shape, e = inf gives rcp = 0 and the result x.
__device__ __forceinline__ f32x2 mish2(f32x2 x) {
    const f32x2 xl = x * 1.4426950408889634f;
    f32x2 e;
    e[0] = __builtin_amdgcn_exp2f(xl[0]);
    e[1] = __builtin_amdgcn_exp2f(xl[1]);
    const f32x2 t = __builtin_elementwise_fma(e, e + 2.f, f32x2{2.f, 2.f});
    f32x2 r;
    r[0] = __builtin_amdgcn_rcpf(t[0]);
    r[1] = __builtin_amdgcn_rcpf(t[1]);
    return x * __builtin_elementwise_fma(r, f32x2{-2.f, -2.f}, f32x2{1.f, 1.f});
}

// The accumulators live in the AGPR half of the register file for the whole kernel ("+a"): as compiler-allocated
// VGPRs hipcc shuffles the 192 accumulator registers around the K loop's back-edge and spills.  The statement is
// volatile, so the MFMA stream keeps the written order relative to the fragment reads and waits.  A and B come out of
// LDS reads retired by an s_waitcnt (no VALU-write hazard); consecutive MFMAs never share an accumulator.
__device__ __forceinline__ void mma_agpr(f32x4& acc, const f16x8& a, const f16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mma_vgpr(f32x4& acc, const f16x8& a, const f16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
// hipcc splits a 256-register budget evenly once a kernel names AGPRs: 128 accumulator registers (32 output tiles)
// sit in AGPRs, the remaining ones share the VGPR half with the fragments
template <int WMT, int I, int J> __device__ __forceinline__ void mma_tile(f32x4& acc, const f16x8& a, const f16x8& b) {
    if constexpr (WMT * J + I < 32) mma_agpr(acc, a, b);
    else mma_vgpr(acc, a, b);
}

// LDS-DMA with a scalar base: lane l moves 16 bytes from sbase + voff(l) to lds_dst + 16*l.  Inline asm because hipcc
// turns (uniform pointer + per-lane offset) into a per-lane 64-bit pointer held in two VGPRs per DMA piece.  M0 is
// compiler-reserved: saved and restored inside the statement; s_nop 4 covers an SGPR operand fresh from a VALU
// readfirstlane, s_nop 0 the M0 write (cdna_hip_programming.md 5.7).
__device__ __forceinline__ void glds16_s(uint32_t voff, const void* sbase, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}

// 16 bytes per lane from sbase + voff(l) into registers, as an asm statement: no compiler-inserted s_waitcnt follows it --
// the caller waits (s_waitcnt vmcnt) before it reads dst.
__device__ __forceinline__ void gload16_s(f16x8& dst, uint32_t voff, const void* sbase) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
}

// s_waitcnt vmcnt(N) that the ND asm-loaded registers depend on (the compiler may not read them before it)
template <int N, int ND, int NA> __device__ __forceinline__ void wait_vm_regs(f16x8 (&r)[NA]) {
    static_assert(ND == 0 || ND == 2 || ND == 4, "0, 2 or 4 register pieces");
    if constexpr (ND == 4) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : "n"(N) : "memory");
    else if constexpr (ND == 2) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(r[0]), "+v"(r[1]) : "n"(N) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// The main loop: returns with the accumulators (bias included) of this wave's (WMT x 12) output tiles.
// `full` = the wave's 12th column tile is in use (else its MFMAs are skipped; other unused tiles are computed on
// whatever the padded pixel slots point at and never stored).
template <int WMT, bool DBG = false, int CHAIN = 0>  // CHAIN: 1 = may find its first weight group in place, 2 = and hands over to the next layer
__device__ __forceinline__ void board_mainloop(const BoardParams& bp, unsigned char* smem, f32x4 (&acc)[WMT][kBoardNJ], int tile,
                                               int kt, int wave, int lane, int col0, bool full, int bs,
                                               unsigned long long* dbg = nullptr,
                                               const __attribute__((address_space(4))) BoardParams* chain = nullptr) {
    using Cfg = BoardCfg<WMT>;
    using namespace board_sched;
    constexpr int KO_T = Cfg::KO_T, NJ = Cfg::NJ, AI = Cfg::AI, NA = WMT;
    const ConvParams& p = bp.c;
    const int npos = bp.npos, b_bytes = npos * 64;
    const uint32_t a_ring = (uint32_t)(uintptr_t)smem;
    const uint32_t b_ring = a_ring + 2 * Cfg::A_BYTES;
    const int wave_m = wave & 3;
    const int kg = lane >> 4;
    const int ls16 = (bs + 2) * 16;  // halo row pitch in bytes of one k-group plane

    // ---- halo DMA: instruction q = wave + 8*i moves k-group plane q & 3 of position block q >> 2 (64 positions);
    // sources are 32-bit offsets from (in - kZeroPrefix): a halo cell reads the buffer's zero prefix, so adding the
    // chunk offset needs no test
    const unsigned char* gin0 = (const unsigned char*)p.in - kZeroPrefix;
    const unsigned char* gw = (const unsigned char*)p.w;
    const int nbinstr = npos / 64 * 4;
    const int nchunks = p.cin_s / kChunk;
    const int ngroups = nchunks * 3;
    const uint32_t lane16 = lane * 16;
    const size_t tap_stride = (size_t)nchunks * 4 * p.ko_pad * 16;  // bytes between taps
    const size_t chunk_stride = (size_t)4 * p.ko_pad * 16;          // bytes between chunks
    // weight piece q = wave + 8*i of a tap tile -> plane q / KO_PARTS, rows (q % KO_PARTS)*64 + lane
    auto issue_a = [&](int G, int dx, int i) {  // piece i of the weight tile (group G, tap-in-row dx) -> ring slot G & 1
        const int q = wave + 8 * i;
        if (q >= Cfg::A_INSTR) return;
        const int kgq = q / Cfg::KO_PARTS, part = q % Cfg::KO_PARTS;
        const int chunk = G / 3, row = G - chunk * 3;
        const unsigned char* base = gw + (size_t)(row * 3 + dx) * tap_stride + (size_t)chunk * chunk_stride +
                                    (size_t)((kgq * p.ko_pad + kt * KO_T + part * 64) * 16);
        glds16_s(lane16, base, a_ring + (G & 1) * Cfg::A_BYTES + dx * Cfg::A_TAP_BYTES + (kgq * KO_T + part * 64) * 16);
    };
    // the first weight group needs no table: it goes out first and lands while the tables are being read
    // (CHAIN: unless the previous layer of the run left it in slot 0 already)
    if (!CHAIN || !bp.w_ready) {
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int i = 0; i < AI; ++i) issue_a(0, dx, i);
    }

    // every table read goes out before the halo DMA (whose "memory" clobber keeps them above it): the halo sources are
    // waited for first, the pixel positions and the bias arrive while the DMA is being issued
    int src[4];
    int lp[NJ];
    if (bp.arith) {
        // what board_setup_kernel wrote for a one-sample tile, recomputed: position -> (halo row, column), pixel -> position
        // (x / n as (int)((x + 0.5) * rcp(n)): x < 512, n >= 4 -- the quotient is never within 0.02 of an integer boundary)
        const int w2 = bs + 2, npix = bs * bs;
        const float r_w2 = __builtin_amdgcn_rcpf((float)w2), r_bs = __builtin_amdgcn_rcpf((float)bs);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = wave + 8 * i, pos = (q >> 2) * 64 + lane;
            const int r = (int)(((float)pos + 0.5f) * r_w2), xc = pos - r * w2;
            const bool in = q < nbinstr && r >= 1 && r <= bs && xc >= 1 && xc <= bs;
            src[i] = in ? tile * p.g.slot_pix + (r - 1) * bs + (xc - 1) : -1;
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int px = (col0 + j) * 16 + (lane & 15);
            const int y = (int)(((float)px + 0.5f) * r_bs), x = px - y * bs;
            lp[j] = px < npix ? (y + 1) * w2 + x + 1 : w2 + 1;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = wave + 8 * i;
            src[i] = q < nbinstr ? bp.tab_src[(size_t)tile * npos + (q >> 2) * 64 + lane] : -1;
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) lp[j] = bp.tab_pix[(size_t)tile * kBoardPT + (col0 + j) * 16 + (lane & 15)].x;
    }
    f32x4 b4[WMT];
#pragma unroll
    for (int i = 0; i < WMT; ++i) b4[i] = *(const f32x4*)(p.bias + kt * KO_T + (wave_m * WMT + i) * 16 + 4 * kg);

    uint32_t boff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        boff[i] = (src[i] >= 0 ? (uint32_t)kZeroPrefix + (uint32_t)src[i] * (uint32_t)(p.cin_s * 2) : 0u) + ((wave + 8 * i) & 3) * 16;
    auto issue_b = [&](int chunk, int i) {
        const int q = wave + 8 * i;
        if (q >= nbinstr) return;
        glds16_s(boff[i], gin0 + chunk * (kChunk * 2), b_ring + (chunk & 1) * b_bytes + ((q & 3) * npos + (q >> 2) * 64) * 16);
    };
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_b(0, i);  // the first halo tile, as soon as its source rows are known

    // per-lane B fragment addresses of the current (halo slot, kernel row); moved by scalars from group to group
    uint32_t bb[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
        bb[j] = b_ring + (((uint32_t)lp[j] & 0xffffu) - 1) * 16 + (uint32_t)kg * npos * 16 - ls16;  // kernel row 0 (dy = -1) in slot 0
    // accumulators start at the bias: D rows 4*(lane>>4)+r of row tile i are 4 consecutive output channels
#pragma unroll
    for (int i = 0; i < WMT; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = b4[i];

    const uint32_t arow_off = (uint32_t)((kg * KO_T + wave_m * WMT * 16 + (lane & 15)) * 16);
    unsigned long long t_sync = 0;  // DBG: cycles spent in vmcnt(0) + s_barrier at the group boundaries
    if constexpr (DBG) { if (dbg) dbg[1] = __builtin_amdgcn_s_memtime(); }

    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const bool more_b = chunk + 1 < nchunks;
        static_for<3>([&](auto rowc) {
            constexpr int row = decltype(rowc)::value;
            const int G = chunk * 3 + row;
            // A(G) and B(chunk) were issued during earlier groups and nothing after them
            unsigned long long t0 = 0;
            if constexpr (DBG) t0 = __builtin_amdgcn_s_memtime();
            wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            if constexpr (DBG) {
                const unsigned long long t1 = __builtin_amdgcn_s_memtime();
                t_sync += t1 - t0;
                if (dbg && G == 0) dbg[2] = t1;
            }
            bool more_a = G + 1 < ngroups;
            int Gn = G + 1;
            if constexpr (CHAIN == 2 && row == 2) {
                // the run's next layer: its first weight group goes into slot 0 while this layer's last group (slot 1: the
                // host hands over only after an even number of groups) is being multiplied.  The pointer is read here, behind
                // an opaque point, so that it does not occupy two SGPRs through the K loop.
                if (!more_a) {
                    const __attribute__((address_space(4))) BoardParams* bq = chain;
                    asm volatile("" : "+s"(bq));
                    const unsigned char* wn = (const unsigned char*)bq->w_next;
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(wn)::"memory");  // a scalar load: retired before the counted LDS reads begin
                    if (wn) { gw = wn; Gn = 0; more_a = true; }
                }
            }
            const uint32_t abase = a_ring + (G & 1) * Cfg::A_BYTES + arow_off;
#if SAYURI_DROP_STREAM
            f16x8 afr[NA] = {}, bfr[3] = {};
#else
            f16x8 afr[NA], bfr[3];
#endif
            board_read_b<0>(bfr[0], bb[0]);
            static_for<WMT>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                board_read_a<i * 256>(afr[i], abase);
            });
            board_read_b<0>(bfr[1], bb[1]);
            static_for<kNB>([&](auto bc) {
                constexpr int b = decltype(bc)::value;
                constexpr int dx = b / NJ, j = b % NJ;
                // DMA of the next group / chunk, front-loaded: weights at blocks 0, 3, 6, ..., halo pieces at blocks 1
                // and 4 of rows 0 and 1 (the next chunk's slot is free from the start of this chunk)
                if constexpr (b % 3 == 0 && b / 3 < 3 * AI) {
                    if (more_a) issue_a(Gn, (b / 3) / AI, (b / 3) % AI);
                }
                if constexpr ((b == 1 || b == 4) && row < 2) {
                    if (more_b) issue_b(chunk + 1, row * 2 + (b == 4 ? 1 : 0));
                }
                if constexpr (b + 2 < kNB) {
                    constexpr int b2 = b + 2;
                    board_read_b<16 * (b2 / NJ)>(bfr[b2 % 3], bb[b2 % NJ]);
                }
                if constexpr (b == 0) wait_lgkm_all<young_b<WMT>(0), WMT>(afr, bfr[0]);
                else if constexpr (j > 0) wait_lgkm1<young_b<WMT>(b)>(bfr[b % 3]);
                static_for<WMT>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    // first block of a tap: A(dx, i) went out during the previous block, A(dx, i+1..) and B(b+2) after it
                    if constexpr (j == 0 && dx > 0) wait_lgkm2<WMT - i>(afr[i], bfr[b % 3]);
                    if constexpr (j == NJ - 1) {
                        if (full) mma_tile<WMT, i, j>(acc[i][j], afr[i], bfr[b % 3]);
                    } else {
                        mma_tile<WMT, i, j>(acc[i][j], afr[i], bfr[b % 3]);
                    }
                    if constexpr (j == NJ - 1 && dx < 2)
                        board_read_a<(dx + 1) * Cfg::A_TAP_BYTES + i * 256>(afr[i], abase);
                });
            });
            // next kernel row: one halo row further; after the third row the other halo slot, two rows back
            const int delta = row < 2 ? ls16 : ((chunk & 1) ? -b_bytes : b_bytes) - 2 * ls16;
#pragma unroll
            for (int j = 0; j < NJ; ++j) bb[j] += (uint32_t)delta;
        });
    }
    // the last MFMAs retire before the epilogue reads the accumulators (hipcc pads nothing after an asm statement)
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    if constexpr (DBG) { if (dbg) { dbg[3] = __builtin_amdgcn_s_memtime(); dbg[5] = t_sync; } }
}

template <int ACT> __device__ __forceinline__ f16x8 board_act8(const float (&v)[8]) {
    f16x8 h;
    if constexpr (ACT == kMish) {
#pragma unroll
        for (int q = 0; q < 8; q += 2) {
            const f32x2 y = mish2(f32x2{v[q], v[q + 1]});
            h[q] = (f16)y[0];
            h[q + 1] = (f16)y[1];
        }
    } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) h[q] = (f16)activate(v[q], ACT);
    }
    return h;
}

// One (column tile, row-tile pair) of the epilogue: swap, + residual, activation, 16-byte store.
template <int ACT>
__device__ __forceinline__ void board_store_pair(f32x4 a, f32x4 b, bool with_res, const f16x8& rr, unsigned char* __restrict__ gbase,
                                                 uint32_t voff, bool ok) {
    swap16(a, b);
    float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    if (with_res) {
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] += (float)rr[q];
    }
    const f16x8 h = board_act8<ACT>(v);
    if (ok) *(f16x8*)(gbase + voff) = h;  // uniform base + 32-bit lane offset: one VGPR of address, no 64-bit arithmetic per store
}

// Epilogue: optional residual, activation, fp16 NHWC store -- straight from the accumulators (bias is in).
// The residual rows are what bounds it when done naively: one load per (column tile, pair) and lane, consumed one
// column tile later, is a chain of twelve memory latencies per wave (36 k cycles per tile measured, unchanged by
// halving the VALU work or the number of active CUs).  So all residual rows of a wave are requested AT ONCE: the
// K loop's LDS rings are dead by now, each wave owns 20 KiB of them and has its residual pieces (1 KiB = 64 lanes x
// 16 bytes, lane-linear: the lane that reads a piece back is the lane that fetched it) delivered there by LDS-DMA,
// the few that do not fit go to registers; one wait covers all of them.
template <int WMT, int ACT>
__device__ __forceinline__ void board_epilogue(const BoardParams& bp, unsigned char* smem, f32x4 (&acc)[WMT][kBoardNJ], int tile, int kt,
                                               int wave, int lane, int col0, int nj) {
    using Cfg = BoardCfg<WMT>;
    constexpr int NJ = Cfg::NJ, NPAIR = WMT / 2;
    constexpr bool LONE = (WMT & 1) != 0;
    const ConvParams& p = bp.c;
    const int R = lane >> 4, px = lane & 15, wave_m = wave & 3;
    const f16* __restrict__ gres = (const f16*)p.res;
    f16* __restrict__ gout = (f16*)p.out;
    const int ko_w = kt * Cfg::KO_T + wave_m * WMT * 16;
    const int2* pix = bp.tab_pix + (size_t)tile * kBoardPT + col0 * 16 + px;
    // after the swap of pair pr this lane holds row tile 2*pr + (R & 1), channels (R >> 1)*8 .. +7 of it
    int cb[NPAIR > 0 ? NPAIR : 1];
#pragma unroll
    for (int pr = 0; pr < NPAIR; ++pr) cb[pr] = ko_w + (2 * pr + (R & 1)) * 16 + (R >> 1) * 8;

    if constexpr (!LONE) {
        // ---- even row-tile counts: residual through the dead LDS rings
        constexpr int kWaveLds = 20 * 1024;                     // 160 KiB / 8 waves (the launch asks for all of the LDS)
        constexpr int kPieces = NJ * NPAIR;                      // residual pieces of a wave
        constexpr int ND = kPieces > 20 ? kPieces - 20 : 0;      // pieces that go to registers instead (the first ones)
        static_assert(ND % (NPAIR > 0 ? NPAIR : 1) == 0, "whole column tiles");
        int orow[NJ];
        f16x8 rrd[ND > 0 ? ND : 1];
        const bool with_res = gres != nullptr;
        // store addresses: output row * row_bytes + channel bytes, all in 32 bits (an activation buffer is far below 4 GiB);
        // the row pitch sits in a VGPR the compiler cannot re-read from the argument block (it did: one scalar load and
        // one lgkmcnt(0) per column tile)
        int row_bytes = p.cout_s * 2;
        asm volatile("" : "+v"(row_bytes));
        uint32_t cb2[NPAIR > 0 ? NPAIR : 1];
        bool cok[NPAIR > 0 ? NPAIR : 1];
#pragma unroll
        for (int pr = 0; pr < NPAIR; ++pr) {
            cb2[pr] = (uint32_t)cb[pr] * 2u;
            cok[pr] = cb[pr] * 2 < row_bytes;
        }
        const uint32_t my_lds = (uint32_t)(uintptr_t)smem + wave * kWaveLds;
        // Column tiles [0, JH) are finished while the residual rows of [JH, nj) are still on their way.  Issue order: the LDS
        // pieces of the first half, the LDS pieces of the second half, last the pieces that go to registers (they belong to
        // the first tiles of the second half).  Only loads are outstanding at the first wait and they retire in issue
        // order, so "at most `late` outstanding" = the first half has landed; that wait touches no register (a wait that
        // names the asm-loaded registers inside a branch makes hipcc copy them BEFORE it); the second wait is vmcnt(0).
        constexpr int JH = NJ / 2, NRT = ND / NPAIR;  // register tiles: [JH, JH + NRT)
        auto load_orow = [&] {  // output activation row of this lane's pixel in each column tile (-1 = none)
            if (bp.arith) {
                const int npix = (bp.uniform_info >> 8) * (bp.uniform_info >> 8);
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int pxl = (col0 + j) * 16 + px;
                    orow[j] = (j < nj && pxl < npix) ? tile * p.g.slot_pix + pxl : -1;
                }
            } else {
#pragma unroll
                for (int j = 0; j < NJ; ++j) orow[j] = j < nj ? pix[j * 16].y : -1;
                // the table rows have arrived HERE: left pending, hipcc puts an s_waitcnt vmcnt(0) in front of every column
                // tile's first use -- behind the join with the computed path, so that every path waited for the previous
                // tile's stores to be acknowledged
#pragma unroll
                for (int j = 0; j < NJ; ++j) asm volatile("" : "+v"(orow[j]));
            }
        };
        auto lds_slot = [](int j, int pr) { return (j < JH ? j : j - NRT) * NPAIR + pr; };
        if (with_res) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // every wave is done with the rings
            load_orow();
            auto roff = [&](int j, int pr) -> uint32_t {  // byte offset of this lane's 8 residual channels
                return (orow[j] >= 0 && cb[pr] < p.cout_s) ? ((uint32_t)orow[j] * (uint32_t)p.cout_s + (uint32_t)cb[pr]) * 2u : 0u;
            };
            static_for<NJ>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                if constexpr (j < JH || j >= JH + NRT) {
                    if (j < nj) {
#pragma unroll
                        for (int pr = 0; pr < NPAIR; ++pr) glds16_s(roff(j, pr), gres, my_lds + lds_slot(j, pr) * 1024);
                    }
                }
            });
            // a register piece is only requested when its tile is in use: on a path that never reads the registers the
            // compiler would hand them out again while the load is still on its way to them
#pragma unroll
            for (int k = 0; k < ND; ++k)
                if (JH + k / NPAIR < nj) gload16_s(rrd[k], roff(JH + k / NPAIR, k % NPAIR), gres);
            // loads younger than the first half = the pieces of the column tiles [JH, nj)
            const int late = (nj > JH ? nj - JH : 0) * NPAIR;
            static_for<(NJ - JH) + 1>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                if (late == t * NPAIR) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(t * NPAIR) : "memory");
            });
        } else {
            load_orow();
        }
        static_for<NJ>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if constexpr (j == JH) {
                if (with_res) wait_vm_regs<0, ND>(rrd);  // the second half's rows (and the first half's stores)
            }
            if (j < nj) {  // wave-uniform
#pragma unroll
                for (int pr = 0; pr < NPAIR; ++pr) {
                    f16x8 rr = {0, 0, 0, 0, 0, 0, 0, 0};
                    if (with_res) {
                        if constexpr (j >= JH && j < JH + NRT) rr = rrd[(j - JH) * NPAIR + pr];
                        else rr = *(const f16x8*)(smem + wave * kWaveLds + lds_slot(j, pr) * 1024 + lane * 16);
                    }
                    board_store_pair<ACT>(acc[2 * pr][j], acc[2 * pr + 1][j], with_res, rr, (unsigned char*)gout,
                                          __umul24((unsigned)(orow[j] >= 0 ? orow[j] : 0), (unsigned)row_bytes) + cb2[pr], orow[j] >= 0 && cok[pr]);
                }
            }
        });
        return;
    }

    // ---- odd row-tile count (192-channel tiles): pairs + a lone row tile, residual rows one column tile ahead
    const int cl = ko_w + (WMT - 1) * 16 + 4 * R;  // lone row tile: accumulator layout, 4 channels per lane
    auto res8 = [&](int orow, int pr) -> f16x8 {
        if (gres && orow >= 0 && cb[pr] < p.cout_s) return *(const f16x8*)(gres + (size_t)orow * p.cout_s + cb[pr]);
        return f16x8{0, 0, 0, 0, 0, 0, 0, 0};
    };
    auto res4 = [&](int orow) -> f16x4 {
        if (gres && orow >= 0 && cl < p.cout_s) return *(const f16x4*)(gres + (size_t)orow * p.cout_s + cl);
        return f16x4{0, 0, 0, 0};
    };
    int orow[3];
    f16x8 rr[2][NPAIR > 0 ? NPAIR : 1];
    f16x4 rl[2];
    orow[0] = nj > 0 ? pix[0].y : -1;
    orow[1] = nj > 1 ? pix[16].y : -1;
#pragma unroll
    for (int pr = 0; pr < NPAIR; ++pr) rr[0][pr] = res8(orow[0], pr);
    rl[0] = res4(orow[0]);
    static_for<NJ>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        if (j < nj) {  // wave-uniform
            if constexpr (j + 2 < NJ) orow[(j + 2) % 3] = j + 2 < nj ? pix[(j + 2) * 16].y : -1;
            if constexpr (j + 1 < NJ) {
#pragma unroll
                for (int pr = 0; pr < NPAIR; ++pr) rr[(j + 1) & 1][pr] = res8(orow[(j + 1) % 3], pr);
                rl[(j + 1) & 1] = res4(orow[(j + 1) % 3]);
            }
            const int my = orow[j % 3];
#pragma unroll
            for (int pr = 0; pr < NPAIR; ++pr) {
                const bool ok = my >= 0 && cb[pr] < p.cout_s;
                board_store_pair<ACT>(acc[2 * pr][j], acc[2 * pr + 1][j], gres != nullptr, rr[j & 1][pr], (unsigned char*)gout,
                                      ((uint32_t)(ok ? my : 0) * (uint32_t)p.cout_s + (uint32_t)cb[pr]) * 2u, ok);
            }
            f32x4 v = acc[WMT - 1][j];
            if (gres) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] += (float)rl[j & 1][q];
            }
            f16x4 h;
#pragma unroll
            for (int q = 0; q < 4; ++q) h[q] = (f16)activate(v[q], ACT);
            if (my >= 0 && cl < p.cout_s) *(f16x4*)(gout + (size_t)my * p.cout_s + cl) = h;
        }
    });
}

__device__ __forceinline__ float max_raw(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// ---- the squeeze-and-excitation unit inside the convolution (reference SEUnit::Forward, se_unit.cc:70-128) -----------
// A block's last 3x3 convolution is followed by  pool -> FC(3C -> se, act) -> FC(se -> 2C) -> act(sigmoid(g) x + b + res).
// When the workgroup holds one whole sample and every channel (KO_T == padded C), x is sitting in its accumulators:
// pooling is a reduction over registers, the two FCs read their weights from L2 (196 + 128 KB at C = 256, the same
// bytes for every workgroup), the gate is applied to the accumulators and the ordinary epilogue (residual, activation,
// store) follows -- no se_pool / se_fc / se_scale launches, x never makes the round trip through HBM.
struct BoardSeParams {
    BoardParams b;
    FcDev squeeze, excite;  // transposed [in][out] fp32 weights (small_ops.h)
    int C;                  // real channel count of the block
    // fp16 images of the two FCs for LDS staging (build_se_images in engine.hip), or null: the FCs then read the fp32
    // weights above from L2.
    //   w1h: [2C rows][se] fp16 -- rows 0..C-1 multiply the channel means, rows C..2C-1 the channel maxima.  The
    //        reference's pooled vector is (mean, mean * (B-14)/10, max) (GlobalPooling<false>, se_unit.cc:9-40): the
    //        second third is a multiple of the first, so its weights are folded in, W' = W_mean + (B-14)/10 * W_scaled --
    //        ONE IMAGE PER BOARD SIZE, w1h + (B - 2) * w1_bytes -- which takes a third off the bytes to stage
    //   w2h: [se/4][2C][4] fp16, then excite bias [2C] fp32, then squeeze bias [se] fp32
    // both padded to whole 1 KiB DMA pieces (w1_bytes / w2_bytes)
    const void* w1h;
    const void* w2h;
    int w1_bytes, w2_bytes;
};

// rotate by `n` lanes inside each row of 16 lanes (DPP row_ror) -- four of them make an all-reduce over a row
template <int N> __device__ __forceinline__ float row_ror(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xf, 0xf, false));
}

// Partial sums of y = W^T x for a block of 512 threads: thread = (4 consecutive outputs, slice of the inputs), all of a
// thread's 16-byte weight loads are independent and go out eight at a time (these FCs are latency-bound: every
// workgroup of the launch reads the same few hundred KB from L2); red[slice][out] is folded by the caller.
// Needs out % 4 == 0 and out / 4 <= 512.
template <int U = 8>  // weight loads in flight per thread
__device__ __forceinline__ void se_fc4(const FcDev fc, const float* x, float* red, int tid) {
    const int quads = fc.out / 4, parts = 512 / quads;
    const int oq = tid % quads, part = tid / quads;
    if (part >= parts) return;
    const int per = (fc.in + parts - 1) / parts;
    const int i0 = part * per, i1 = min(fc.in, i0 + per);
    const float* w = fc.wt + oq * 4;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    for (int i = i0; i < i1; i += U) {
        f32x4 wv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) wv[u] = i + u < i1 ? *(const f32x4*)(w + (size_t)(i + u) * fc.out) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < U; ++u) a += (i + u < i1 ? x[i + u] : 0.f) * wv[u];
    }
    *(f32x4*)(red + part * fc.out + oq * 4) = a;
}

// LDS of the SE stage (byte offsets BEHIND the two weight images of the staged form; the K loop's rings are dead by then):
// [2][KO_T] sums, [2][KO_T] maxima (one row per wave column), pool[3 KO_T], red[2048], mid[512], gate[2 KO_T].  The stage is
// three steps -- pooling, the two FCs, the gate -- that meet in this layout; the persistent tower launch runs the first and the
// last as generated assembly on the K loop's own register assignment and the FCs as a compiled body of their own
// (conv_tower.h, tower_seam.py), the per-layer kernel below runs all three as compiled code.
template <int WMT> struct SeLds {
    static constexpr int KO_T = BoardCfg<WMT>::KO_T;
    static constexpr int psum = 0, pmax = psum + 2 * KO_T * 4, pool = pmax + 2 * KO_T * 4, red = pool + 3 * KO_T * 4,
                         mid = red + 2048 * 4, gate = mid + 512 * 4, end = gate + 2 * KO_T * 4;
};

// Step 1: pooling.  Leaves the per-wave-column partial sums and maxima of every channel in LDS and the two FC images on
// their way (staged form); returns behind the barrier that publishes both.
template <int WMT>
__device__ __forceinline__ void board_se_pool(const BoardSeParams& sp, unsigned char* smem, f32x4 (&acc)[WMT][kBoardNJ], int tile,
                                              int wave, int lane, int col0, int nj, int bs) {
    using Cfg = BoardCfg<WMT>;
    constexpr int NJ = Cfg::NJ, KO_T = Cfg::KO_T;
    const int q = lane >> 4, px = lane & 15, wave_m = wave & 3, wave_n = wave >> 2;
    const bool staged = sp.w1h != nullptr;
    const int w_bytes = staged ? sp.w1_bytes + sp.w2_bytes : 0;
    float* psum = (float*)(smem + w_bytes + SeLds<WMT>::psum);
    float* pmax = (float*)(smem + w_bytes + SeLds<WMT>::pmax);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every wave is done with the rings
    if (staged) {
        // both images on their way into LDS while the accumulators are being pooled: linear 1 KiB pieces, the squeeze
        // image of this tile's board size first (it is needed first)
        const int n1 = sp.w1_bytes >> 10, n2 = sp.w2_bytes >> 10;
        const unsigned char* g1 = (const unsigned char*)sp.w1h + (size_t)(bs - 2) * sp.w1_bytes;
        const unsigned char* g2 = (const unsigned char*)sp.w2h;
        const uint32_t l0 = (uint32_t)(uintptr_t)smem;
        for (int k = wave; k < n1; k += 8) glds16_s(lane * 16, g1 + k * 1024, l0 + k * 1024);
        for (int k = wave; k < n2; k += 8) glds16_s(lane * 16, g2 + k * 1024, l0 + sp.w1_bytes + k * 1024);
    }

    // ---- per lane over its column tiles (valid pixels only), then over the 16 pixel lanes of a row.
    // A one-sample tile has its unused pixel slots at the end: only the wave's LAST column tile can hold any, so only
    // that one is masked (the others are summed with packed adds).  One row tile at a time: eight live temporaries
    // beside the 192 accumulators instead of thirty-two (hipcc parked accumulators in scratch for the wider form --
    // 17 MB of spill stores per launch in the PMC pass)
    const bool last_valid = nj <= 0 ? false : (col0 + nj - 1) * 16 + px < bs * bs;
#pragma unroll
    for (int i = 0; i < WMT; ++i) {
        f32x4 s4 = {0.f, 0.f, 0.f, 0.f}, m4 = {-5000.f, -5000.f, -5000.f, -5000.f};
        // max_raw: v_max_f32 as is -- fmaxf() makes hipcc quiet both operands first (three v_max per maximum: 1 264 of them in this
        // stage); no NaN can come out of the MFMAs of finite weights and activations, and a NaN in would be one out either way.
        // The guards are wave-uniform and must stay branches: if-converted, every tile carried both forms and 8 selects.
        static_for<NJ>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if (j + 1 < nj) {  // a full tile
                asm volatile("");
                const f32x4 v = acc[i][j];
                s4 += v;
#pragma unroll
                for (int r = 0; r < 4; ++r) m4[r] = max_raw(m4[r], v[r]);
            }
        });
        static_for<NJ>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if (j + 1 == nj) {  // the wave's last tile: its unused pixel slots are masked
                asm volatile("");
                const f32x4 v = acc[i][j];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    s4[r] += last_valid ? v[r] : 0.f;
                    m4[r] = max_raw(m4[r], last_valid ? v[r] : -5000.f);
                }
            }
        });
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float a = s4[r], b = m4[r];
            a += row_ror<8>(a); b = max_raw(b, row_ror<8>(b));
            a += row_ror<4>(a); b = max_raw(b, row_ror<4>(b));
            a += row_ror<2>(a); b = max_raw(b, row_ror<2>(b));
            a += row_ror<1>(a); b = max_raw(b, row_ror<1>(b));
            s4[r] = a; m4[r] = b;
        }
        if (px == 0) {
            const int c0 = wave_m * WMT * 16 + i * 16 + 4 * q;
            *(f32x4*)(psum + wave_n * KO_T + c0) = s4;
            *(f32x4*)(pmax + wave_n * KO_T + c0) = m4;
        }
    }
    if (staged) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of the images have landed
    __syncthreads();
}

// Step 2: the two FCs, from the pooled partials to gate[c] = sigmoid(gamma_c), gate[KO_T + c] = beta_c in LDS (pad channels
// [C, KO_T): both 0 -- their weights and bias are 0, x stays 0).  Needs only LDS and the thread id: the tower launch runs it as a
// body of its own (tower_se_fc_kernel) between the assembly pooling and the assembly gate.
template <int WMT, int U = 8>  // U: se_fc4's loads in flight (the tower's FC body has 62 VGPRs: 4)
__device__ __forceinline__ void board_se_fc(const BoardSeParams& sp, unsigned char* smem, int tid, int bs, unsigned long long* dbg = nullptr) {
    using Cfg = BoardCfg<WMT>;
    constexpr int KO_T = Cfg::KO_T;
    const ConvParams& p = sp.b.c;
    const int C = sp.C, so = sp.squeeze.out;
    const bool staged = sp.w1h != nullptr;
    const int w_bytes = staged ? sp.w1_bytes + sp.w2_bytes : 0;
    float* psum = (float*)(smem + w_bytes + SeLds<WMT>::psum);
    float* pmax = (float*)(smem + w_bytes + SeLds<WMT>::pmax);
    float* pool = (float*)(smem + w_bytes + SeLds<WMT>::pool);
    float* red = (float*)(smem + w_bytes + SeLds<WMT>::red);  // [slices][outputs] of an FC: 512 threads x 4 floats
    float* mid = (float*)(smem + w_bytes + SeLds<WMT>::mid);
    float* gate = (float*)(smem + w_bytes + SeLds<WMT>::gate);
    if (dbg) dbg[2] = __builtin_amdgcn_s_memtime();  // pooled partials exchanged
    const float npix = (float)(bs * bs), bd = (float)bs - 14.f;
    for (int c = C + tid; c < KO_T; c += 512) { gate[c] = 0.f; gate[KO_T + c] = 0.f; }
    if (staged) {
        // ---- squeeze FC out of LDS: thread = (4 consecutive outputs, every parts-th row); a row's pooled value is folded
        // from the two wave columns' partials on the fly (mean rows: (s0 + s1) / npix, max rows: max(m0, m1))
        const int quads = so >> 2, parts = 512 / quads, oq = tid % quads, part = tid / quads;
        const float inv = 1.0f / npix;
        const unsigned char* w1 = smem + oq * 8;
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        if (part < parts) {
            for (int r = part; r < 2 * C; r += parts) {
                const f16x4 w = *(const f16x4*)(w1 + (size_t)r * so * 2);
                const float x = r < C ? (psum[r] + psum[KO_T + r]) * inv : fmaxf(pmax[r - C], pmax[KO_T + r - C]);
                a[0] += x * (float)w[0]; a[1] += x * (float)w[1]; a[2] += x * (float)w[2]; a[3] += x * (float)w[3];
            }
            *(f32x4*)(red + part * so + oq * 4) = a;
        }
        __syncthreads();
        const unsigned char* w2 = smem + sp.w1_bytes;
        const float* b2 = (const float*)(w2 + (size_t)so * 2 * C * 2);
        const float* b1 = b2 + 2 * C;
        for (int o = tid; o < so; o += 512) {
            float t = b1[o];
            for (int k = 0; k < parts; ++k) t += red[k * so + o];
            mid[o] = activate(t, p.act);
        }
        __syncthreads();
        if (dbg) dbg[3] = __builtin_amdgcn_s_memtime();  // squeeze FC done
        // ---- excite FC out of LDS: thread = one output, 4 inputs per 8-byte read
        for (int o = tid; o < 2 * C; o += 512) {
            float t = b2[o];
            for (int i = 0; i < so; i += 4) {
                const f16x4 w = *(const f16x4*)(w2 + ((size_t)(i >> 2) * 2 * C + o) * 8);
                const f32x4 m = *(const f32x4*)(mid + i);
                t += m[0] * (float)w[0] + m[1] * (float)w[1] + m[2] * (float)w[2] + m[3] * (float)w[3];
            }
            // gate[c] = sigmoid(gamma_c), gate[KO_T + c] = beta_c
            gate[o < C ? o : KO_T + (o - C)] = o < C ? 1.0f / (1.0f + fast_exp(-t)) : t;
        }
        __syncthreads();
        if (dbg) dbg[4] = __builtin_amdgcn_s_memtime();  // excite FC done, gate in LDS
    } else {
        for (int c = tid; c < C; c += 512) {
            const float mean = (psum[c] + psum[KO_T + c]) / npix;
            pool[c] = mean;
            pool[C + c] = mean * (bd / 10.f);
            pool[2 * C + c] = fmaxf(pmax[c], pmax[KO_T + c]);
        }
        __syncthreads();
        // ---- the two FCs (se_fc4 above): squeeze with the unit's activation, excite into the gate
        se_fc4<U>(sp.squeeze, pool, red, tid);
        __syncthreads();
        for (int o = tid; o < so; o += 512) {
            float a = sp.squeeze.b[o];
            const int parts = 512 / (so / 4);
            for (int k = 0; k < parts; ++k) a += red[k * so + o];
            mid[o] = activate(a, p.act);
        }
        __syncthreads();
        if (dbg) dbg[3] = __builtin_amdgcn_s_memtime();  // squeeze FC done
        se_fc4<U>(sp.excite, mid, red, tid);
        __syncthreads();
        {
            const int eo = sp.excite.out, parts = 512 / (eo / 4);
            for (int o = tid; o < eo; o += 512) {
                float a = sp.excite.b[o];
                for (int k = 0; k < parts; ++k) a += red[k * eo + o];
                // gate[c] = sigmoid(gamma_c), gate[KO_T + c] = beta_c
                gate[o < C ? o : KO_T + (o - C)] = o < C ? 1.0f / (1.0f + fast_exp(-a)) : a;
            }
        }
        __syncthreads();
        if (dbg) dbg[4] = __builtin_amdgcn_s_memtime();  // excite FC done, gate in LDS
    }
}

// Step 3: x <- sigmoid(gamma) x + beta on the accumulators (one fused multiply-add per value).
template <int WMT>
__device__ __forceinline__ void board_se_gate(const BoardSeParams& sp, unsigned char* smem, f32x4 (&acc)[WMT][kBoardNJ], int wave, int lane) {
    using Cfg = BoardCfg<WMT>;
    constexpr int NJ = Cfg::NJ, KO_T = Cfg::KO_T;
    const int q = lane >> 4, wave_m = wave & 3;
    const int w_bytes = sp.w1h != nullptr ? sp.w1_bytes + sp.w2_bytes : 0;
    const float* gate = (const float*)(smem + w_bytes + SeLds<WMT>::gate);
#pragma unroll
    for (int i = 0; i < WMT; ++i) {
        const int c0 = wave_m * WMT * 16 + i * 16 + 4 * q;
        const f32x4 g = *(const f32x4*)(gate + c0), be = *(const f32x4*)(gate + KO_T + c0);
        static_for<NJ>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = __builtin_fmaf(g[r], acc[i][j][r], be[r]);
        });
    }
    // the epilogue starts with a barrier before it reuses the LDS
}

template <int WMT>
__device__ __forceinline__ void board_se_stage(const BoardSeParams& sp, unsigned char* smem, f32x4 (&acc)[WMT][kBoardNJ], int tile,
                                               int wave, int lane, int col0, int nj, int bs, unsigned long long* dbg = nullptr) {
    board_se_pool<WMT>(sp, smem, acc, tile, wave, lane, col0, nj, bs);
    board_se_fc<WMT>(sp, smem, wave * 64 + lane, bs, dbg);
    board_se_gate<WMT>(sp, smem, acc, wave, lane);
}

template <int WMT>
__global__ __launch_bounds__(512, 2) void conv_board_se_kernel(const BoardSeParams sp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const BoardParams& bp = sp.b;
    const ConvParams& p = bp.c;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x;  // one output-channel tile: KO_T covers the layer
    const int info = bp.uniform_info >= 0 ? bp.uniform_info : __builtin_amdgcn_readfirstlane(bp.tab_cols[tile]);
    const int ncols = info & 0xff, bs = info >> 8;
    const int nj0 = (ncols + 1) >> 1;
    const int wave_n = wave >> 2;
    const int col0 = wave_n ? nj0 : 0;
    const int nj = wave_n ? ncols - nj0 : nj0;

    // timeline (SAYURI_BOARD_DBG=-n): [0] start, [1] K loop done, [2] pooled, [3] squeeze FC, [4] excite FC, [5] gate applied, [6] end
    unsigned long long* dbg = nullptr;
    if (bp.dbg && blockIdx.x < 4 && lane == 0) {
        dbg = bp.dbg + ((size_t)blockIdx.x * 8 + wave) * 8;
        dbg[0] = __builtin_amdgcn_s_memtime();
    }
    f32x4 acc[WMT][kBoardNJ];
    board_mainloop<WMT>(bp, smem, acc, tile, 0, wave, lane, col0, nj == kBoardNJ, bs);
    if (dbg) dbg[1] = __builtin_amdgcn_s_memtime();
    board_se_stage<WMT>(sp, smem, acc, tile, wave, lane, col0, nj, bs, dbg);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // gate fully read before the epilogue's residual pieces land in the same LDS
    if (dbg) dbg[5] = __builtin_amdgcn_s_memtime();

    switch (p.act) {
    case kMish: board_epilogue<WMT, kMish>(bp, smem, acc, tile, 0, wave, lane, col0, nj); break;
    case kIdentity: board_epilogue<WMT, kIdentity>(bp, smem, acc, tile, 0, wave, lane, col0, nj); break;
    case kReLU: board_epilogue<WMT, kReLU>(bp, smem, acc, tile, 0, wave, lane, col0, nj); break;
    case kSwish: board_epilogue<WMT, kSwish>(bp, smem, acc, tile, 0, wave, lane, col0, nj); break;
    case kELU: board_epilogue<WMT, kELU>(bp, smem, acc, tile, 0, wave, lane, col0, nj); break;
    case kSELU: board_epilogue<WMT, kSELU>(bp, smem, acc, tile, 0, wave, lane, col0, nj); break;
    case kGELU: board_epilogue<WMT, kGELU>(bp, smem, acc, tile, 0, wave, lane, col0, nj); break;
    default: board_epilogue<WMT, kHardSwish>(bp, smem, acc, tile, 0, wave, lane, col0, nj); break;
    }
    if (dbg) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        dbg[6] = __builtin_amdgcn_s_memtime();
    }
}

template <int WMT, bool DBG = false>
__global__ __launch_bounds__(512, 2) void conv_board_kernel(const BoardParams bp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* dbg = nullptr;
    if constexpr (DBG) {
        if (bp.dbg && blockIdx.x < 4 && (threadIdx.x & 63) == 0) {
            dbg = bp.dbg + ((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 8;
            dbg[0] = __builtin_amdgcn_s_memtime();
        }
    }
    const ConvParams& p = bp.c;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // (ConvParams::npos, the generic kernels' halo bound, is this launch's FIRST tile here: a launch may cover the tile range
    // [npos, npos + num_pix_tiles) of the batch -- Engine::conv_se splits an SE layer of a mixed batch by board size)
    const int tile = p.npos + blockIdx.x % p.num_pix_tiles;
    const int kt = blockIdx.x / p.num_pix_tiles;
    // column tiles: wave column 0 (waves 0-3) takes the first ceil(n/2), wave column 1 (waves 4-7, the SIMD partners
    // of 0-3) the rest
    const int info = bp.uniform_info >= 0 ? bp.uniform_info : __builtin_amdgcn_readfirstlane(bp.tab_cols[tile]);
    const int ncols = info & 0xff, bs = info >> 8;
    const int nj0 = (ncols + 1) >> 1;
    const int wave_n = wave >> 2;
    const int col0 = wave_n ? nj0 : 0;
    const int nj = wave_n ? ncols - nj0 : nj0;

    f32x4 acc[WMT][kBoardNJ];
    board_mainloop<WMT, DBG>(bp, smem, acc, tile, kt, wave, lane, col0, nj == kBoardNJ, bs, dbg);

    switch (p.act) {
    case kMish: board_epilogue<WMT, kMish>(bp, smem, acc, tile, kt, wave, lane, col0, nj); break;
    case kIdentity: board_epilogue<WMT, kIdentity>(bp, smem, acc, tile, kt, wave, lane, col0, nj); break;
    case kReLU: board_epilogue<WMT, kReLU>(bp, smem, acc, tile, kt, wave, lane, col0, nj); break;
    case kSwish: board_epilogue<WMT, kSwish>(bp, smem, acc, tile, kt, wave, lane, col0, nj); break;
    case kELU: board_epilogue<WMT, kELU>(bp, smem, acc, tile, kt, wave, lane, col0, nj); break;
    case kSELU: board_epilogue<WMT, kSELU>(bp, smem, acc, tile, kt, wave, lane, col0, nj); break;
    case kGELU: board_epilogue<WMT, kGELU>(bp, smem, acc, tile, kt, wave, lane, col0, nj); break;
    default: board_epilogue<WMT, kHardSwish>(bp, smem, acc, tile, kt, wave, lane, col0, nj); break;
    }
    if constexpr (DBG) {
        if (dbg) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            dbg[4] = __builtin_amdgcn_s_memtime();
        }
    }
}

}  // namespace sayuri
