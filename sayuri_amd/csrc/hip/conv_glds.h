// conv_glds.h -- the tuned fp16 3x3 convolution (implicit GEMM, same LDS images as
// conv_mfma.h) built around LDS-DMA and a hand-counted LDS fragment pipeline.
//
// Structure (one workgroup = 8 waves = KO_T output channels x PT pixels):
//   * every byte reaches LDS by `global_load_lds_dwordx4` (no VGPR staging, no ds_write pass);
//   * K is walked as (32-channel chunk c) x (kernel row r): group G = 3c + r holds the three
//     weight tiles of taps 3r..3r+2 (48 KiB at KO_T = 256) in ring slot G & 1; the halo tile
//     B(c) of a chunk is double-buffered and re-read by all nine taps;
//   * ONE raw s_barrier per group (24 per 256-channel layer), s_waitcnt vmcnt(0) just before
//     it; the DMA of group G+1 / chunk c+1 is issued in thirds between the MFMA blocks of
//     group G, so loads are in flight across the whole group;
//   * fragment reads are inline-asm ds_read_b128 with hand-counted s_waitcnt lgkmcnt(N): while an
//     LDS-DMA is pending hipcc degrades every LDS wait to lgkmcnt(0) and re-exposes the LDS
//     latency before each MFMA group (cdna_hip_programming.md 5.7);
//   * the epilogue stages fp32 accumulators through LDS and writes whole NHWC rows (512 B per
//     pixel) instead of 8-byte fragments at a 512-byte stride; residual rows are read the
//     same way.  Results are bit-identical to conv_mfma.h's epilogue.
//   * per-tile index tables (halo source rows, pixel -> LDS position / output row) depend only on
//     the batch geometry, so they are built once per uploaded batch by tile_setup_kernel and
//     shared by every layer.
#pragma once
#include <type_traits>
#include <utility>

#include "common.h"
#include "conv_mfma.h"

namespace sayuri {

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

__device__ __forceinline__ void glds16(const void* g, void* lds_wave_base) {
    // 64 lanes x 16 B: lane l reads its own global address, lands at lds_wave_base + 16*l
    __builtin_amdgcn_global_load_lds((glb_void*)g, (lds_void*)(uint32_t)(uintptr_t)lds_wave_base, 16, 0, 0);
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// LDS fragment read that hipcc does not count (see header).
template <int OFF> __device__ __forceinline__ void ds_read16(f16x8& dst, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
// Wait until at most N younger LDS reads are outstanding.  The fragments named here count as
// (re)defined by the wait, so no consumer can be scheduled above it.
template <int N, int NB> __device__ __forceinline__ void wait_frags(f16x8& a, f16x8 (&b)[NB]) {
    static_assert(NB >= 1 && NB <= 6, "unsupported fragment count");
    if constexpr (NB == 1) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b[0]) : "n"(N));
    else if constexpr (NB == 2) asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a), "+v"(b[0]), "+v"(b[1]) : "n"(N));
    else if constexpr (NB == 3)
        asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]) : "n"(N));
    else if constexpr (NB == 4)
        asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(a), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) : "n"(N));
    else if constexpr (NB == 5)
        asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(a), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]) : "n"(N));
    else
        asm volatile("s_waitcnt lgkmcnt(%7)"
                     : "+v"(a), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5])
                     : "n"(N));
}

// compile-time loop: f(integral_constant<int, 0>{}), ..., f(integral_constant<int, N-1>{})
template <typename F, int... Q> __device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Q...>) {
    (f(std::integral_constant<int, Q>{}), ...);
}
template <int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

template <int WMT_, int WNT_> struct GldsCfg {
    static constexpr int WMT = WMT_, WNT = WNT_, WAVM = 2, WAVN = 4, NWAVE = WAVM * WAVN;
    static constexpr int KO_T = WAVM * WMT * 16;
    static constexpr int PT = WAVN * WNT * 16;
    static constexpr int NT = 64 * NWAVE;
    static constexpr int NPOS_CAP = ((PT + PT / 4 + 128 + 63) / 64) * 64;  // halo positions, multiple of 64
    static constexpr int A_INSTR = KO_T * 64 / 1024;                      // 1 KiB DMA instructions per tap tile
    static constexpr int AI = A_INSTR / NWAVE;                            // per wave per tap
    static constexpr int B_INSTR = NPOS_CAP / 64 * 4;
    static constexpr int BI = (B_INSTR + NWAVE - 1) / NWAVE;              // per wave per chunk
    static constexpr int NPOS = BI * NWAVE / 4 * 64;                      // positions incl. dummy DMA slots
    static constexpr int A_TAP_BYTES = KO_T * 64;
    static constexpr int A_BYTES = 3 * A_TAP_BYTES;
    static constexpr int B_BYTES = NPOS * 64;
    static constexpr int STAGE_RS = KO_T * 4 + 16;                        // epilogue staging row (fp32 + pad)
    static constexpr int STAGE_BYTES = (PT / 2) * STAGE_RS;
    static constexpr int RING_BYTES = 2 * A_BYTES + 2 * B_BYTES;
    static_assert(A_INSTR % NWAVE == 0, "KO_T must be a multiple of 128");
    static_assert(BI <= 6, "halo DMA is spread over the six DMA slots of a group");
    static_assert(STAGE_BYTES <= RING_BYTES, "the epilogue staging tile re-uses the DMA rings");
    static constexpr int ROWID_OFF = RING_BYTES;                          // int rowid[PT] behind the rings
    static constexpr size_t lds_bytes() { return RING_BYTES + PT * 4; }
};

struct GldsParams {
    ConvParams c;
    const void* zeros;    // >= 64 bytes of zeros in global memory (source of halo / dummy rows)
    const int* tab_src;   // [tile][NPOS]   activation row feeding each halo position, -1 = zero
    const int2* tab_pix;  // [tile][PT]     x = lpos | lstr << 16, y = output activation row (-1 = none)
};

// Per-tile index tables (shared by all layers of one forward).  One workgroup per pixel tile.
// Subregion s of a tile = the board rows of one sample it covers plus a one-cell zero halo.
template <int PT, int NPOS>
__global__ __launch_bounds__(256) void tile_setup_kernel(BatchGeom g, int* __restrict__ tab_src,
                                                         int2* __restrict__ tab_pix) {
    __shared__ int hdr[8 + 8 * kMaxSub];
    const int tile = blockIdx.x, tid = threadIdx.x;
    const int g0 = tile * PT, total = g.total_pix;
    if (tid == 0) {
        const int g1 = min(g0 + PT, total);
        int lo = 0, hi = g.n_samples;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (g.sample_off[mid] <= g0) lo = mid; else hi = mid;
        }
        int n = lo, base = 0, cnt = 0;
        while (n < g.n_samples && cnt < kMaxSub) {
            const int off = g.sample_off[n];
            if (off >= g1) break;
            const int bs = g.bsz[n];
            const int a = max(g0, off) - off, b = min(g1, g.sample_off[n + 1]) - off;
            const int ylo = a / bs, yhi = (b - 1) / bs, rows = yhi - ylo + 3;
            int* sb = hdr + 8 + 8 * cnt;
            sb[0] = base; sb[1] = ylo; sb[2] = bs; sb[3] = n;
            sb[4] = off + a; sb[5] = off + b; sb[6] = rows; sb[7] = off;
            base += rows * (bs + 2);
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
            const int bs = sb[2], w2 = bs + 2, rel = pos - sb[0];
            if (rel >= 0 && rel < sb[6] * w2) {
                const int r = rel / w2, xc = rel - r * w2;
                const int y = sb[1] - 1 + r, x = xc - 1;
                if (y >= 0 && y < bs && x >= 0 && x < bs) src = sb[3] * g.slot_pix + y * bs + x;
                break;
            }
        }
        tab_src[(size_t)tile * NPOS + pos] = src;
    }
    for (int i = tid; i < PT; i += blockDim.x) {
        const int gi = g0 + i;
        int lstr = hdr[8 + 2] + 2, lpos = lstr + 1, orow = -1;  // an interior cell: every tap in range
        if (gi < total) {
            for (int s = 0; s < nsub; ++s) {
                const int* sb = hdr + 8 + 8 * s;
                if (gi >= sb[4] && gi < sb[5]) {
                    const int bs = sb[2], pp = gi - sb[7];
                    const int y = pp / bs, x = pp - y * bs;
                    lstr = bs + 2;
                    lpos = sb[0] + (y - sb[1] + 1) * (bs + 2) + x + 1;
                    orow = sb[3] * g.slot_pix + pp;
                    break;
                }
            }
        }
        tab_pix[(size_t)tile * PT + i] = make_int2(lpos | (lstr << 16), orow);
    }
}

// Cooperative second half of the epilogue: whole rows out of the fp32 staging tile.
// EpiRows holds, per lane, the output row ids and residual rows of one phase; they are fetched for
// BOTH phases before the first staging pass, so one memory latency covers the whole epilogue.
template <int KO_T, int PT, int NW = 8> struct EpiRows {
    static constexpr int HALF = PT / 2;
    static constexpr int LPR = KO_T / 8;          // lanes per row (8 channels = 16 B of fp16 each)
    static constexpr int RPI = 64 / LPR;          // rows per wave instruction
    static constexpr int NR = HALF / (NW * RPI);  // rows per lane per phase
    static constexpr int STEP = NW * RPI;         // row distance between a lane's consecutive rows
    static_assert(HALF % (NW * RPI) == 0, "rows must split evenly over the waves");
    int grow[NR];
    f16x8 rr[NR];
};

template <int KO_T, int PT, int NW>
__device__ __forceinline__ void epi_fetch(EpiRows<KO_T, PT, NW>& e, const GldsParams& gp, const int* rowid, int phase,
                                          int kt, int wave, int lane) {
    using E = EpiRows<KO_T, PT, NW>;
    const ConvParams& p = gp.c;
    const f16* __restrict__ gres = (const f16*)p.res;
    const int ko = kt * KO_T + (lane % E::LPR) * 8;
    const bool ko_ok = ko < p.cout_s;  // cout_s is a multiple of 32, so 8-channel groups never straddle it
    const int r0 = wave * E::RPI + lane / E::LPR;
#pragma unroll
    for (int k = 0; k < E::NR; ++k) e.grow[k] = ko_ok ? rowid[phase * E::HALF + r0 + k * E::STEP] : -1;
    if (gres) {
#pragma unroll
        for (int k = 0; k < E::NR; ++k)
            e.rr[k] = e.grow[k] >= 0 ? *(const f16x8*)(gres + (size_t)e.grow[k] * p.cout_s + ko)
                                     : f16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
}

template <int ACT, int KO_T, int PT, int NW>
__device__ __forceinline__ void epi_store(const EpiRows<KO_T, PT, NW>& e, const GldsParams& gp, const unsigned char* stage,
                                          int kt, int wave, int lane) {
    using E = EpiRows<KO_T, PT, NW>;
    constexpr int RS = KO_T * 4 + 16;
    const ConvParams& p = gp.c;
    f16* __restrict__ gout = (f16*)p.out;
    const int col = (lane % E::LPR) * 8;
    const int ko = kt * KO_T + col;
    const int r0 = wave * E::RPI + lane / E::LPR;
#pragma unroll
    for (int k = 0; k < E::NR; ++k) {
        const unsigned char* src = stage + (r0 + k * E::STEP) * RS + col * 4;
        const f32x4 v0 = *(const f32x4*)src, v1 = *(const f32x4*)(src + 16);
        float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        if (p.res) {
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] += (float)e.rr[k][q];
        }
        f16x8 h;
#pragma unroll
        for (int q = 0; q < 8; ++q) h[q] = (f16)activate(v[q], ACT);
        if (e.grow[k] >= 0) *(f16x8*)(gout + (size_t)e.grow[k] * p.cout_s + ko) = h;
    }
}

template <int WMT, int WNT>
__global__ __launch_bounds__(512, 2) void conv_glds_kernel(const GldsParams gp) {
    using Cfg = GldsCfg<WMT, WNT>;
    constexpr int KO_T = Cfg::KO_T, PT = Cfg::PT, WAVN = Cfg::WAVN, NWAVE = Cfg::NWAVE;
    constexpr int AI = Cfg::AI, BI = Cfg::BI, NPOS = Cfg::NPOS;
    const ConvParams& p = gp.c;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Aring = smem;
    unsigned char* Bring = smem + 2 * Cfg::A_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave / WAVN, wave_n = wave % WAVN;
    const int tile = blockIdx.x % p.num_pix_tiles;
    const int kt = blockIdx.x / p.num_pix_tiles;

    // ---- DMA roles.  Halo instruction q = wave + 8*i covers k-group plane q & 3 of position
    // block q >> 2 (64 positions); this lane's source row for each of them is fixed.
    const unsigned char* gin = (const unsigned char*)p.in;
    const unsigned char* gw = (const unsigned char*)p.w;
    const unsigned char* bsrc[BI];
    int bdst[BI];
#pragma unroll
    for (int i = 0; i < BI; ++i) {
        const int q = wave + NWAVE * i, kgq = q & 3, blk = q >> 2;
        const int src = gp.tab_src[(size_t)tile * NPOS + blk * 64 + lane];
        bsrc[i] = src >= 0 ? gin + ((size_t)src * p.cin_s + kgq * 8) * 2 : (const unsigned char*)gp.zeros;
        bdst[i] = (kgq * NPOS + blk * 64) * 16;
    }
    int* rowid = (int*)(smem + Cfg::ROWID_OFF);
    if (tid < PT) rowid[tid] = gp.tab_pix[(size_t)tile * PT + tid].y;
    int lpos[WNT], lstr[WNT];
#pragma unroll
    for (int j = 0; j < WNT; ++j) {
        const int2 pk = gp.tab_pix[(size_t)tile * PT + (wave_n * WNT + j) * 16 + (lane & 15)];
        lpos[j] = pk.x & 0xffff;
        lstr[j] = pk.x >> 16;
    }
    const int nchunks = p.cin_s / kChunk;
    const int ngroups = nchunks * 3;
    // weight piece q = wave + 8*i of a tap tile -> plane q / (KO_T/64), rows (q % (KO_T/64))*64 + lane
    size_t aoff[AI];
    int adst[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int q = wave + NWAVE * i;
        const int kgq = q / (KO_T / 64), part = q % (KO_T / 64);
        aoff[i] = (((size_t)kgq * p.ko_pad + (size_t)kt * KO_T + part * 64 + lane) * 8) * 2;
        adst[i] = (kgq * KO_T + part * 64) * 16;
    }
    const size_t tap_stride = (size_t)nchunks * 4 * p.ko_pad * 16;  // bytes between taps
    const size_t chunk_stride = (size_t)4 * p.ko_pad * 16;          // bytes between chunks
    auto issue_a = [&](int G, int dx) {  // weight tile of (group G, tap-in-row dx) -> ring slot G & 1
        const int chunk = G / 3, row = G - chunk * 3;
        const unsigned char* base = gw + (size_t)(row * 3 + dx) * tap_stride + (size_t)chunk * chunk_stride;
        unsigned char* slot = Aring + (G & 1) * Cfg::A_BYTES + dx * Cfg::A_TAP_BYTES;
#pragma unroll
        for (int i = 0; i < AI; ++i) glds16(base + aoff[i], slot + adst[i]);
    };
    auto issue_b1 = [&](int chunk, int i) {
        unsigned char* slot = Bring + (chunk & 1) * Cfg::B_BYTES;
        const unsigned char* s = bsrc[i];
        if (s != (const unsigned char*)gp.zeros) s += (size_t)chunk * kChunk * 2;
        glds16(s, slot + bdst[i]);
    };

    f32x4 acc[WMT][WNT];
#pragma unroll
    for (int i = 0; i < WMT; ++i)
#pragma unroll
        for (int j = 0; j < WNT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int i = 0; i < BI; ++i) issue_b1(0, i);
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) issue_a(0, dx);

    const int kg = lane >> 4;
    const int arow_off = (kg * KO_T + wave_m * WMT * 16 + (lane & 15)) * 16;

    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const bool more_b = chunk + 1 < nchunks;
        const unsigned char* Bc = Bring + (chunk & 1) * Cfg::B_BYTES + (size_t)kg * NPOS * 16;
#pragma unroll
        for (int row = 0; row < 3; ++row) {
            const int G = chunk * 3 + row;
            // A(G) and B(chunk) were issued during the previous group and nothing after them.
            wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            const bool more_a = G + 1 < ngroups;
            const int dy = row - 1;
            // ---- software-pipelined fragment stream (hand-counted LDS queue) ----
            // MFMA block q = (tap dx = q / WMT, row tile i = q % WMT) consumes A(q) and B(dx).
            // Reads run D-1 blocks ahead; B(dx+1) is read at block WMT*dx + IB, before that
            // block's A read, so it is older than every A fragment of tap dx+1.
            constexpr int D = 4, NQ = 3 * WMT, IB = WMT - 4;
            const uint32_t abase = (uint32_t)(uintptr_t)(Aring + (G & 1) * Cfg::A_BYTES + arow_off);
            uint32_t bbase[WNT];
#pragma unroll
            for (int j = 0; j < WNT; ++j)
                bbase[j] = (uint32_t)(uintptr_t)Bc + (uint32_t)((lpos[j] + dy * lstr[j] - 1) * 16);
            f16x8 bfr[2][WNT];
            f16x8 afr[D];
#pragma unroll
            for (int j = 0; j < WNT; ++j) ds_read16<0>(bfr[0][j], bbase[j]);
            ds_read16<0>(afr[0], abase);
            ds_read16<256>(afr[1], abase);
            ds_read16<512>(afr[2], abase);
            static_for<NQ>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                constexpr int dx = q / WMT, i = q % WMT;
                // DMA of the next group, front-loaded: block 0: tap 0 + halo piece 0, block WMT/2:
                // tap 1, block WMT: tap 2 + halo piece 1, block 3*WMT/2: halo piece 2.
                constexpr int SP = WMT / 2;  // DMA slot spacing
                constexpr int NSLOT = BI > 3 ? 6 : 4;
                if constexpr (q % SP == 0 && q / SP < NSLOT) {
                    constexpr int slot = q / SP;
                    // halo pieces: three or fewer go out at slots 0, 2, 3; more than three one per slot
                    constexpr int bpiece = BI > 3 ? slot : (slot == 0 ? 0 : slot == 2 ? 1 : slot == 3 ? 2 : -1);
                    constexpr int atap = slot < 3 ? slot : -1;
                    if constexpr (bpiece >= 0 && bpiece < BI) {
                        if (row == 0 && more_b) issue_b1(chunk + 1, bpiece);
                    }
                    if constexpr (atap >= 0) {
                        if (more_a) issue_a(G + 1, atap);
                    }
                }
                if constexpr (i == IB && dx < 2) {
#pragma unroll
                    for (int j = 0; j < WNT; ++j) ds_read16<16 * (dx + 1)>(bfr[(dx + 1) & 1][j], bbase[j]);
                }
                if constexpr (q + D - 1 < NQ) {
                    constexpr int q2 = q + D - 1;
                    ds_read16<(q2 / WMT) * Cfg::A_TAP_BYTES + (q2 % WMT) * 256>(afr[q2 % D], abase);
                }
                // reads issued after A(q): the next D-1 A fragments that exist, plus B(dx+1)
                // when it went out after A(q) (A(q) is issued at block q-(D-1)).
                constexpr int gb = WMT * dx + IB;
                constexpr int young = (NQ - 1 - q < D - 1 ? NQ - 1 - q : D - 1) +
                                      ((dx < 2 && gb <= q && gb > q - (D - 1)) ? WNT : 0);
                wait_frags<young>(afr[q % D], bfr[dx & 1]);
#pragma unroll
                for (int j = 0; j < WNT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(afr[q % D], bfr[dx & 1][j], acc[i][j], 0, 0, 0);
            });
        }
    }

    // ---- epilogue: fp32 accumulators (+ bias) -> LDS [pixel][channel], two phases of PT/2
    // pixels (wave columns 0-1, then 2-3), then whole rows: + residual, activation, fp16 store.
    // Barriers here are raw s_barrier + lgkmcnt(0): a __syncthreads() would also drain vmcnt,
    // i.e. wait for the previous phase's global stores to be acknowledged.
    constexpr int RS = Cfg::STAGE_RS;
    unsigned char* stage = smem;
    auto lds_barrier = [] {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    lds_barrier();  // rowid visible, rings no longer read
    EpiRows<KO_T, PT, NWAVE> rows0, rows1;
    epi_fetch(rows0, gp, rowid, 0, kt, wave, lane);
    epi_fetch(rows1, gp, rowid, 1, kt, wave, lane);
#pragma unroll
    for (int phase = 0; phase < 2; ++phase) {
        if (phase) lds_barrier();  // staging tile of the previous phase fully read
        if ((wave_n / (WAVN / 2)) == phase) {
#pragma unroll
            for (int i = 0; i < WMT; ++i) {
                const int kol = wave_m * WMT * 16 + i * 16 + 4 * (lane >> 4);  // channel inside the tile
                const f32x4 bias = *(const f32x4*)(p.bias + kt * KO_T + kol);
#pragma unroll
                for (int j = 0; j < WNT; ++j) {
                    const int rl = ((wave_n % (WAVN / 2)) * WNT + j) * 16 + (lane & 15);
                    *(f32x4*)(stage + rl * RS + kol * 4) = acc[i][j] + bias;
                }
            }
        }
        lds_barrier();
        const EpiRows<KO_T, PT, NWAVE>& rows = phase ? rows1 : rows0;
        switch (p.act) {
        case kMish: epi_store<kMish, KO_T, PT, NWAVE>(rows, gp, stage, kt, wave, lane); break;
        case kIdentity: epi_store<kIdentity, KO_T, PT, NWAVE>(rows, gp, stage, kt, wave, lane); break;
        case kReLU: epi_store<kReLU, KO_T, PT, NWAVE>(rows, gp, stage, kt, wave, lane); break;
        case kSwish: epi_store<kSwish, KO_T, PT, NWAVE>(rows, gp, stage, kt, wave, lane); break;
        case kELU: epi_store<kELU, KO_T, PT, NWAVE>(rows, gp, stage, kt, wave, lane); break;
        case kSELU: epi_store<kSELU, KO_T, PT, NWAVE>(rows, gp, stage, kt, wave, lane); break;
        case kGELU: epi_store<kGELU, KO_T, PT, NWAVE>(rows, gp, stage, kt, wave, lane); break;
        default: epi_store<kHardSwish, KO_T, PT, NWAVE>(rows, gp, stage, kt, wave, lane); break;
        }
    }
}

}  // namespace sayuri
