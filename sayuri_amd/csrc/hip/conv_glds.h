// conv_glds.h -- the tuned fp16 3x3 convolution: same implicit GEMM and LDS images as
// conv_mfma.h, but every byte reaches LDS by LDS-DMA (`global_load_lds_dwordx4`), staged
// through an NS-deep ring with counted `s_waitcnt vmcnt(N)` and ONE raw `s_barrier` per
// k-step, so weight and activation loads stay in flight across barriers
// (cdna_hip_programming.md section 5 "Pipelining across barriers", T3+T4).
//
//   step s = (chunk c, tap t), s = 9c + t.        A(s): weight tile [4][KO_T][8] halfs (16 B rows)
//   ring slot s % NS holds A(s).                  B(c): halo tile   [4][NPOS][8] halfs, 2 slots
//
//   prologue : issue B(0), A(0) .. A(NS-2)
//   step s   : wait vmcnt(younger loads)      -> this wave's share of A(s) (and B(c)) has landed
//              s_barrier                      -> everyone's share has landed; slot (s-1)%NS is free
//              if t == 0: issue B(c+1)        -> into the B slot last read during chunk c-1
//              issue A(s+NS-1)                -> into slot (s-1)%NS
//              ds_read fragments of A(s), B(c) shifted by tap t; MFMA
//
// Loads complete in issue order, so "at most N younger loads outstanding" == "A(s) landed".
// Every wave issues the same static number of DMA instructions per step (AI) and per chunk
// (BI), which makes N a compile-time constant per tap (the tap loop is unrolled).
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

template <int N> __device__ __forceinline__ void wait_lgkmcnt_raw() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}
// LDS fragment read that hipcc does not see as an LDS access.  While an LDS-DMA is in flight
// the compiler's waitcnt pass degrades every LDS wait to lgkmcnt(0) ("pending flat"), which
// exposes the full LDS latency before each MFMA group; these reads are counted by hand
// (cdna_hip_programming.md 5.7: loads + waits owned by the kernel author).
template <int OFF> __device__ __forceinline__ void ds_read16(f16x8& dst, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
// wait until at most N younger LDS reads are outstanding; the fragments named here count as
// (re)defined by the wait, so no consumer can be scheduled above it.
template <int N, int NB> __device__ __forceinline__ void wait_frags(f16x8& a, f16x8 (&b)[NB]) {
    if constexpr (NB == 2) asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a), "+v"(b[0]), "+v"(b[1]) : "n"(N));
    else if constexpr (NB == 3)
        asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]) : "n"(N));
    else
        asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(a), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) : "n"(N));
}
template <int N> __device__ __forceinline__ void wait_frag1(f16x8& a) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N));
}

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(integral_constant<int, N-1>{})
template <typename F, int... Q> __device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Q...>) {
    (f(std::integral_constant<int, Q>{}), ...);
}
template <int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

template <int WMT_, int WNT_, int NS_> struct GldsCfg {
    static constexpr int WMT = WMT_, WNT = WNT_, NS = NS_, WAVM = 2, WAVN = 4, NWAVE = 8;
    static constexpr int KO_T = WAVM * WMT * 16;
    static constexpr int PT = WAVN * WNT * 16;
    static constexpr int NT = 512;
    static constexpr int NPOS_CAP = ((PT + PT / 4 + 128 + 63) / 64) * 64;  // multiple of 64
    static constexpr int A_INSTR = KO_T * 64 / 1024;                      // 1 KiB DMA instrs per A tile
    static constexpr int AI = A_INSTR / NWAVE;                            // per wave
    static constexpr int B_INSTR = NPOS_CAP / 64 * 4;
    static constexpr int BI = (B_INSTR + NWAVE - 1) / NWAVE;
    static constexpr int NPOS_ALLOC = BI * NWAVE / 4 * 64;                // positions incl. dummy slots
    static constexpr int A_BYTES = KO_T * 64;
    static constexpr int B_BYTES = NPOS_ALLOC * 64;
    static_assert(A_INSTR % NWAVE == 0, "KO_T must be a multiple of 128");
    static constexpr size_t lds_bytes() {
        return kHdrBytes + NPOS_ALLOC * 4 + (size_t)NS * A_BYTES + 2 * (size_t)B_BYTES;
    }
};

struct GldsParams {
    ConvParams c;
    const void* zeros;  // >= 64 bytes of zeros in global memory (source of halo / dummy rows)
};

template <int ACT, int WMT, int WNT>
__device__ __forceinline__ void glds_epilogue(const ConvParams& p, f32x4 (&acc)[WMT][WNT], const int (&orow)[WNT],
                                              int ko_base, int lane) {
    f16* __restrict__ gout = (f16*)p.out;
    const f16* __restrict__ gres = (const f16*)p.res;
#pragma unroll
    for (int i = 0; i < WMT; ++i) {
        const int ko = ko_base + i * 16 + 4 * (lane >> 4);
        if (ko >= p.cout_s) continue;
        const f32x4 bias = *(const f32x4*)(p.bias + ko);
#pragma unroll
        for (int j = 0; j < WNT; ++j) {
            if (orow[j] < 0) continue;
            const size_t o = (size_t)orow[j] * p.cout_s + ko;
            f32x4 v = acc[i][j] + bias;
            if (gres) {
                const f16x4 r = *(const f16x4*)(gres + o);
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] += (float)r[q];
            }
            f16x4 h;
#pragma unroll
            for (int q = 0; q < 4; ++q) h[q] = (f16)activate(v[q], ACT);
            *(f16x4*)(gout + o) = h;
        }
    }
}

template <int WMT, int WNT, int NS>
__global__ __launch_bounds__(512) void conv_glds_kernel(const GldsParams gp) {
    using Cfg = GldsCfg<WMT, WNT, NS>;
    constexpr int KO_T = Cfg::KO_T, PT = Cfg::PT, NT = Cfg::NT, WAVN = Cfg::WAVN;
    constexpr int AI = Cfg::AI, BI = Cfg::BI, NPOS = Cfg::NPOS_ALLOC;
    const ConvParams& p = gp.c;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* hdr = (int*)smem;
    int* srctab = (int*)(smem + kHdrBytes);
    unsigned char* Aring = smem + kHdrBytes + NPOS * 4;
    unsigned char* Bring = Aring + NS * Cfg::A_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave / WAVN, wave_n = wave % WAVN;
    const int tile = blockIdx.x % p.num_pix_tiles;
    const int kt = blockIdx.x / p.num_pix_tiles;
    const int g0 = tile * PT;
    const int total = p.g.total_pix;

    if (tid == 0) {
        const int g1 = min(g0 + PT, total);
        int lo = 0, hi = p.g.n_samples;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (p.g.sample_off[mid] <= g0) lo = mid; else hi = mid;
        }
        int n = lo, base = 0, cnt = 0;
        while (n < p.g.n_samples && cnt < kMaxSub) {
            const int off = p.g.sample_off[n];
            if (off >= g1) break;
            const int bs = p.g.bsz[n];
            const int a = max(g0, off) - off, b = min(g1, p.g.sample_off[n + 1]) - off;
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

    for (int pos = tid; pos < NPOS; pos += NT) {
        int src = -1;
        for (int s = 0; s < nsub; ++s) {
            const int* sb = hdr + 8 + 8 * s;
            const int bs = sb[2], w2 = bs + 2, rel = pos - sb[0];
            if (rel >= 0 && rel < sb[6] * w2) {
                const int r = rel / w2, xc = rel - r * w2;
                const int y = sb[1] - 1 + r, x = xc - 1;
                if (y >= 0 && y < bs && x >= 0 && x < bs) src = sb[3] * p.g.slot_pix + y * bs + x;
                break;
            }
        }
        srctab[pos] = src;
    }

    int lpos[WNT], lstr[WNT], orow[WNT];
#pragma unroll
    for (int j = 0; j < WNT; ++j) {
        const int gi = g0 + (wave_n * WNT + j) * 16 + (lane & 15);
        lstr[j] = hdr[8 + 2] + 2;
        lpos[j] = lstr[j] + 1;
        orow[j] = -1;
        if (gi < total) {
            for (int s = 0; s < nsub; ++s) {
                const int* sb = hdr + 8 + 8 * s;
                if (gi >= sb[4] && gi < sb[5]) {
                    const int bs = sb[2], pp = gi - sb[7];
                    const int y = pp / bs, x = pp - y * bs;
                    lstr[j] = bs + 2;
                    lpos[j] = sb[0] + (y - sb[1] + 1) * (bs + 2) + x + 1;
                    orow[j] = sb[3] * p.g.slot_pix + pp;
                    break;
                }
            }
        }
    }
    __syncthreads();

    // ---- DMA roles.  B instruction q = wave + 8*i covers k-group plane q & 3 of position block
    // q >> 2 (64 positions); this lane's source row for each of them is fixed for the kernel.
    const unsigned char* gin = (const unsigned char*)p.in;
    const unsigned char* gw = (const unsigned char*)p.w;
    const unsigned char* bsrc[BI];
    int bdst[BI];
#pragma unroll
    for (int i = 0; i < BI; ++i) {
        const int q = wave + 8 * i, kg = q & 3, blk = q >> 2;
        const int pos = blk * 64 + lane;
        const int src = srctab[pos];
        bsrc[i] = src >= 0 ? gin + ((size_t)src * p.cin_s + kg * 8) * 2 : (const unsigned char*)gp.zeros;
        bdst[i] = (kg * NPOS + blk * 64) * 16;  // wave-uniform byte offset inside a B slot
    }
    const int nchunks = p.cin_s / kChunk;
    const int nsteps = nchunks * 9;
    // A tile of step (chunk, tap): planes of KO_T rows at ((tap*nchunks+chunk)*4 + kg)*ko_pad + kt*KO_T
    auto issue_a = [&](int step) {
        const int chunk = step / 9, tap = step - chunk * 9;
        const size_t ws = (size_t)tap * nchunks + chunk;
        unsigned char* slot = Aring + (step % NS) * Cfg::A_BYTES;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const int q = wave + 8 * i;                 // 1 KiB piece index inside the tile
            const int kg = q / (KO_T / 64), part = q % (KO_T / 64);
            const unsigned char* src = gw + (((ws * 4 + kg) * p.ko_pad + (size_t)kt * KO_T + part * 64 + lane) * 8) * 2;
            glds16(src, slot + (kg * KO_T + part * 64) * 16);
        }
    };
    auto issue_b = [&](int chunk) {
        unsigned char* slot = Bring + (chunk & 1) * Cfg::B_BYTES;
        const size_t coff = (size_t)chunk * kChunk * 2;
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            const unsigned char* s = bsrc[i];
            if (s != (const unsigned char*)gp.zeros) s += coff;
            glds16(s, slot + bdst[i]);
        }
    };

    f32x4 acc[WMT][WNT];
#pragma unroll
    for (int i = 0; i < WMT; ++i)
#pragma unroll
        for (int j = 0; j < WNT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    issue_b(0);
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) issue_a(s);  // nsteps >= 9 > NS-1

    const int kg = lane >> 4;
    const int arow_off = (kg * KO_T + wave_m * WMT * 16 + (lane & 15)) * 16;

    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const bool more_b = chunk + 1 < nchunks;
        const unsigned char* Bc = Bring + (chunk & 1) * Cfg::B_BYTES + (size_t)kg * NPOS * 16;
        const bool last = chunk + 1 == nchunks;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int step = chunk * 9 + tap;
            // loads younger than A(step): A(step+1 .. step+NS-2), plus B(chunk+1) while tap in [1, NS-2]
            constexpr int kYoungA = (NS - 2) * AI;
            if (!last) {
                if (tap >= 1 && tap <= NS - 2) wait_vmcnt<kYoungA + BI>();
                else wait_vmcnt<kYoungA>();
            } else {
                // the ring drains: only A(step+1 .. min(step+NS-2, nsteps-1)) are younger
                const int young = 8 - tap < NS - 2 ? 8 - tap : NS - 2;
                switch (young) {
                case 0: wait_vmcnt<0>(); break;
                case 1: wait_vmcnt<AI>(); break;
                case 2: wait_vmcnt<2 * AI>(); break;
                default: wait_vmcnt<kYoungA>(); break;
                }
            }
            __builtin_amdgcn_s_barrier();
            if (tap == 0 && more_b) issue_b(chunk + 1);
            if (step + NS - 1 < nsteps) issue_a(step + NS - 1);

            const unsigned char* Ac = Aring + (step % NS) * Cfg::A_BYTES + arow_off;
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
            f16x8 bf[WNT];
#pragma unroll
            for (int j = 0; j < WNT; ++j) bf[j] = *(const f16x8*)(Bc + (lpos[j] + dy * lstr[j] + dx) * 16);
#pragma unroll
            for (int i = 0; i < WMT; ++i) {
                const f16x8 af = *(const f16x8*)(Ac + i * 256);
#pragma unroll
                for (int j = 0; j < WNT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf[j], acc[i][j], 0, 0, 0);
            }
        }
    }

    const int ko_base = kt * KO_T + wave_m * WMT * 16;
    switch (p.act) {
    case kMish: glds_epilogue<kMish, WMT, WNT>(p, acc, orow, ko_base, lane); break;
    case kIdentity: glds_epilogue<kIdentity, WMT, WNT>(p, acc, orow, ko_base, lane); break;
    case kReLU: glds_epilogue<kReLU, WMT, WNT>(p, acc, orow, ko_base, lane); break;
    case kSwish: glds_epilogue<kSwish, WMT, WNT>(p, acc, orow, ko_base, lane); break;
    case kELU: glds_epilogue<kELU, WMT, WNT>(p, acc, orow, ko_base, lane); break;
    case kSELU: glds_epilogue<kSELU, WMT, WNT>(p, acc, orow, ko_base, lane); break;
    case kGELU: glds_epilogue<kGELU, WMT, WNT>(p, acc, orow, ko_base, lane); break;
    default: glds_epilogue<kHardSwish, WMT, WNT>(p, acc, orow, ko_base, lane); break;
    }
}

}  // namespace sayuri

// ======================================================================================
// Row-grouped variant: ONE barrier per kernel ROW (3 taps) instead of per tap.
//   group G = (chunk c, row r), G = 3c + r; A(G) = the three weight tiles of taps 3r..3r+2
//   (48 KiB at KO_T = 256) in ring slot G & 1; B(c) as before.
//   step G : wait vmcnt -> barrier -> [per tap dx: issue 1/3 of A(G+1) (+ 1/3 of B(c+1) when
//            r == 0), then that tap's fragments + MFMAs]
// 24 barriers per 256-channel layer instead of 72; the DMA issue is spread between the MFMA
// blocks so it never stalls a whole step.
namespace sayuri {

template <int WMT_, int WNT_> struct Glds3Cfg {
    static constexpr int WMT = WMT_, WNT = WNT_, WAVM = 2, WAVN = 4, NWAVE = 8;
    static constexpr int KO_T = WAVM * WMT * 16;
    static constexpr int PT = WAVN * WNT * 16;
    static constexpr int NT = 512;
    static constexpr int NPOS_CAP = ((PT + PT / 4 + 128 + 63) / 64) * 64;
    static constexpr int A_INSTR = KO_T * 64 / 1024;  // per tap
    static constexpr int AI = A_INSTR / NWAVE;        // per wave per tap
    static constexpr int B_INSTR = NPOS_CAP / 64 * 4;
    static constexpr int BI = (B_INSTR + NWAVE - 1) / NWAVE;
    static constexpr int NPOS_ALLOC = BI * NWAVE / 4 * 64;
    static constexpr int A_TAP_BYTES = KO_T * 64;
    static constexpr int A_BYTES = 3 * A_TAP_BYTES;
    static constexpr int B_BYTES = NPOS_ALLOC * 64;
    static_assert(A_INSTR % NWAVE == 0, "KO_T must be a multiple of 128");
    static_assert(BI <= 3, "B issue is spread over the three taps of a row");
    static constexpr size_t lds_bytes() { return kHdrBytes + NPOS_ALLOC * 4 + 2 * (size_t)A_BYTES + 2 * (size_t)B_BYTES; }
};

// ABL: timing-only ablation mask (never used by the engine proper): 1 = no weight DMA after the
// first group, 2 = no halo DMA after the first chunk, 4 = no MFMA, 8 = no epilogue.
template <int WMT, int WNT, int ABL = 0>
__global__ __launch_bounds__(512) void conv_glds3_kernel(const GldsParams gp) {
    using Cfg = Glds3Cfg<WMT, WNT>;
    constexpr int KO_T = Cfg::KO_T, PT = Cfg::PT, NT = Cfg::NT, WAVN = Cfg::WAVN;
    constexpr int AI = Cfg::AI, BI = Cfg::BI, NPOS = Cfg::NPOS_ALLOC;
    const ConvParams& p = gp.c;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* hdr = (int*)smem;
    int* srctab = (int*)(smem + kHdrBytes);
    unsigned char* Aring = smem + kHdrBytes + NPOS * 4;
    unsigned char* Bring = Aring + 2 * Cfg::A_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave / WAVN, wave_n = wave % WAVN;
    const int tile = blockIdx.x % p.num_pix_tiles;
    const int kt = blockIdx.x / p.num_pix_tiles;
    const int g0 = tile * PT;
    const int total = p.g.total_pix;

    if (tid == 0) {
        const int g1 = min(g0 + PT, total);
        int n;
        if (p.npos > 0) {  // uniform boards: npos carries bs*bs, no search needed
            n = g0 / p.npos;
        } else {
            int lo = 0, hi = p.g.n_samples;
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (p.g.sample_off[mid] <= g0) lo = mid; else hi = mid;
            }
            n = lo;
        }
        int base = 0, cnt = 0;
        while (n < p.g.n_samples && cnt < kMaxSub) {
            const int off = p.g.sample_off[n];
            if (off >= g1) break;
            const int bs = p.g.bsz[n];
            const int a = max(g0, off) - off, b = min(g1, p.g.sample_off[n + 1]) - off;
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

    for (int pos = tid; pos < NPOS; pos += NT) {
        int src = -1;
        for (int s = 0; s < nsub; ++s) {
            const int* sb = hdr + 8 + 8 * s;
            const int bs = sb[2], w2 = bs + 2, rel = pos - sb[0];
            if (rel >= 0 && rel < sb[6] * w2) {
                const int r = rel / w2, xc = rel - r * w2;
                const int y = sb[1] - 1 + r, x = xc - 1;
                if (y >= 0 && y < bs && x >= 0 && x < bs) src = sb[3] * p.g.slot_pix + y * bs + x;
                break;
            }
        }
        srctab[pos] = src;
    }

    int lpos[WNT], lstr[WNT], orow[WNT];
#pragma unroll
    for (int j = 0; j < WNT; ++j) {
        const int gi = g0 + (wave_n * WNT + j) * 16 + (lane & 15);
        lstr[j] = hdr[8 + 2] + 2;
        lpos[j] = lstr[j] + 1;
        orow[j] = -1;
        if (gi < total) {
            for (int s = 0; s < nsub; ++s) {
                const int* sb = hdr + 8 + 8 * s;
                if (gi >= sb[4] && gi < sb[5]) {
                    const int bs = sb[2], pp = gi - sb[7];
                    const int y = pp / bs, x = pp - y * bs;
                    lstr[j] = bs + 2;
                    lpos[j] = sb[0] + (y - sb[1] + 1) * (bs + 2) + x + 1;
                    orow[j] = sb[3] * p.g.slot_pix + pp;
                    break;
                }
            }
        }
    }
    __syncthreads();

    const unsigned char* gin = (const unsigned char*)p.in;
    const unsigned char* gw = (const unsigned char*)p.w;
    const unsigned char* bsrc[BI];
    int bdst[BI];
#pragma unroll
    for (int i = 0; i < BI; ++i) {
        const int q = wave + 8 * i, kgq = q & 3, blk = q >> 2;
        const int src = srctab[blk * 64 + lane];
        bsrc[i] = src >= 0 ? gin + ((size_t)src * p.cin_s + kgq * 8) * 2 : (const unsigned char*)gp.zeros;
        bdst[i] = (kgq * NPOS + blk * 64) * 16;
    }
    const int nchunks = p.cin_s / kChunk;
    const int ngroups = nchunks * 3;
    // per-lane part of the weight address: piece q = wave + 8*i -> plane q/(KO_T/64), rows (q%(KO_T/64))*64+lane
    size_t aoff[AI];
    int adst[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int q = wave + 8 * i;
        const int kgq = q / (KO_T / 64), part = q % (KO_T / 64);
        aoff[i] = (((size_t)kgq * p.ko_pad + (size_t)kt * KO_T + part * 64 + lane) * 8) * 2;
        adst[i] = (kgq * KO_T + part * 64) * 16;
    }
    const size_t tap_stride = (size_t)nchunks * 4 * p.ko_pad * 16;   // bytes between taps
    const size_t chunk_stride = (size_t)4 * p.ko_pad * 16;           // bytes between chunks
    // issue the weight tile of (group G, tap-in-row dx) into ring slot G & 1
    auto issue_a = [&](int G, int dx) {
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
            // A(G) was issued during the previous step and nothing after it: drain everything.
            wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            const bool more_a = G + 1 < ngroups;
            const int dy = row - 1;
            // ---- software-pipelined fragment stream (hand-counted LDS queue) ----
            // group q = (tap dx = q / WMT, row tile i = q % WMT) consumes A(q) and B(dx).
            // Reads run D-1 groups ahead of the MFMAs; B(dx+1) is read at group WMT*dx + IB,
            // before that group's A read, so it is older than every A fragment of tap dx+1.
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
                if constexpr (i == 0) {  // this tap's share of the DMA traffic
                    if constexpr (!(ABL & 2)) {
                        if (row == 0 && more_b && dx < BI) issue_b1(chunk + 1, dx);
                    }
                    if constexpr (!(ABL & 1)) {
                        if (more_a) issue_a(G + 1, dx);
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
                // when it was issued after A(q) (A(q) goes out at group q-(D-1)).
                constexpr int gb = WMT * dx + IB;
                constexpr int young = (NQ - 1 - q < D - 1 ? NQ - 1 - q : D - 1) +
                                      ((dx < 2 && gb <= q && gb > q - (D - 1)) ? WNT : 0);
                wait_frags<young>(afr[q % D], bfr[dx & 1]);
                if constexpr (!(ABL & 4)) {
#pragma unroll
                    for (int j = 0; j < WNT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(afr[q % D], bfr[dx & 1][j], acc[i][j], 0, 0, 0);
                } else {
                    acc[i][0][0] += (float)afr[q % D][0] + (float)bfr[dx & 1][0][0];
                }
            });
        }
    }

    if constexpr (ABL & 8) {
        if (acc[0][0][0] == 12345.678f) ((float*)p.out)[tid] = acc[1][1][1];
        return;
    }
    const int ko_base = kt * KO_T + wave_m * WMT * 16;
    switch (p.act) {
    case kMish: glds_epilogue<kMish, WMT, WNT>(p, acc, orow, ko_base, lane); break;
    case kIdentity: glds_epilogue<kIdentity, WMT, WNT>(p, acc, orow, ko_base, lane); break;
    case kReLU: glds_epilogue<kReLU, WMT, WNT>(p, acc, orow, ko_base, lane); break;
    case kSwish: glds_epilogue<kSwish, WMT, WNT>(p, acc, orow, ko_base, lane); break;
    case kELU: glds_epilogue<kELU, WMT, WNT>(p, acc, orow, ko_base, lane); break;
    case kSELU: glds_epilogue<kSELU, WMT, WNT>(p, acc, orow, ko_base, lane); break;
    case kGELU: glds_epilogue<kGELU, WMT, WNT>(p, acc, orow, ko_base, lane); break;
    default: glds_epilogue<kHardSwish, WMT, WNT>(p, acc, orow, ko_base, lane); break;
    }
}

}  // namespace sayuri
