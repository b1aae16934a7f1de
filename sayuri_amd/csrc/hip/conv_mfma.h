// Implicit-GEMM 3x3 / 1x1 convolution on the gfx950 matrix cores, fused bias + residual +
// activation epilogue.  Replaces, for one layer, the reference's transform_in ->
// gemm_strided_batched -> transform_out triple (src/neural/cuda/cuda_layers.cc:455-619,
// cuda_kernels.cu:521-667) and its im2col/1x1 variants (cuda_kernels.cu:182-239,
// cuda_layers.cc:586-604), plus add_spatial (cuda_kernels.cu:37-79).
//
// Why direct and not Winograd on this chip: F(4x4,3x3) cuts multiplies 3.6x but moves
// 2.5x more bytes through V/M scratch (566 MB per 256->256 layer at batch 256 vs 95 MB
// direct).  At 2.5 PFLOP/s : 8 TB/s the un-fused Winograd layer is HBM-bound at ~113 us,
// the direct layer is MFMA-bound at ~44 us and keeps fp16 error at the level of a plain
// dot product.  See DESIGN.md "Kernel 1".
//
// GEMM view:  D[ko][pix] = sum_{tap,c} W[ko][tap][c] * X[pix shifted by tap][c]
//   M = output channels (A operand rows, from LDS image of the weights)
//   N = valid board pixels of the whole batch, compact (no off-board work, no tile padding
//       except the last tile), B operand columns
//   K = taps * cin, walked as (32-channel chunk) x (tap): the input halo tile of a chunk is
//       staged in LDS ONCE and re-read by all 9 taps through per-lane shifted addresses.
//
// Workgroup = 8 waves (2 along M x 4 along N), tile KO_T x PT = (2*WMT*16) x (4*WNT*16).
// LDS images are "k-group planes": [k/8][row][8 elems], so that every ds_read_b128 of an
// MFMA fragment (lane -> row lane&15, k-group lane>>4) touches 16 distinct 16-byte slots.
#pragma once
#include "common.h"

namespace sayuri {

constexpr int kChunk = 32;     // channels per K chunk
constexpr int kMaxSub = 34;    // max samples a pixel tile may touch (a 128-pixel tile over 2x2 boards: 33)
constexpr int kHdrBytes = 1280;  // 8 + 8 * kMaxSub ints

struct ConvParams {
    const void* in;     // [slot][pix][cin_s]
    const void* w;      // device image [tap][chunk][4][ko_pad][8]
    const float* bias;  // [ko_pad]
    const void* res;    // optional residual, same geometry as out
    void* out;          // [slot][pix][cout_s]
    BatchGeom g;
    int cin_s;          // input channel stride == padded cin (multiple of 32)
    int cout_s;         // output channel stride (multiple of 32)
    int ko_pad;         // weight rows (multiple of KO_T)
    int taps;           // 1 or 9
    int act;
    int npos;           // LDS halo positions per chunk (multiple of 16, host-computed bound)
    int num_pix_tiles;
};

template <typename T, int WMT_, int WNT_> struct ConvCfg {
    static constexpr int WMT = WMT_, WNT = WNT_, WAVM = 2, WAVN = 4;
    static constexpr int KO_T = WAVM * WMT * 16;
    static constexpr int PT = WAVN * WNT * 16;
    static constexpr int NT = 64 * WAVM * WAVN;
    // halo positions a pixel tile may need: a tile over several small boards pays a one-cell frame per board
    // (worst mixes of 9/13/19 boards measured: 304 @ PT 128, 400 @ 192, 480 @ 256)
    // The 128-pixel tiles (WNT = 2) are the fallback for ANY geometry the reference accepts (boards from 2x2, types.h:23): a tile
    // over 2x2 boards touches 33 samples with a 4x4 halo each -- 4 positions per pixel (round 6: the fuzz drew a batch with two
    // dozen 2x2 / 3x3 boards in a row on the 96-channel network and the engine refused it).
    static constexpr int NPOS_CAP = WNT_ == 2 ? 4 * PT + 32 : PT + PT / 2 + 160;
    static size_t lds_bytes(int npos) {
        return kHdrBytes + ((npos * 4 + 15) & ~15) + 2 * (size_t)KO_T * kChunk * sizeof(T) +
               2 * (size_t)npos * kChunk * sizeof(T);
    }
};

template <typename T> struct Frag;
template <> struct Frag<f16> { typedef f16x8 type; };
template <> struct Frag<float> { struct type { f32x4 lo, hi; }; };

template <typename T> __device__ __forceinline__ typename Frag<T>::type lds_frag(const T* p);
template <> __device__ __forceinline__ f16x8 lds_frag<f16>(const f16* p) { return *(const f16x8*)p; }
template <> __device__ __forceinline__ Frag<float>::type lds_frag<float>(const float* p) {
    Frag<float>::type f;
    f.lo = *(const f32x4*)p;
    f.hi = *(const f32x4*)(p + 4);
    return f;
}

__device__ __forceinline__ f32x4 mma(const f16x8& a, const f16x8& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
// fp32 mode: lane group g = lane>>4 owns channels 8g..8g+7 of the chunk; step s feeds channel
// 8g+s into v_mfma_f32_16x16x4_f32 (k slot = g).  Any channel<->(step,slot) assignment is a
// valid contraction order as long as A and B agree.
__device__ __forceinline__ f32x4 mma(const Frag<float>::type& a, const Frag<float>::type& b, f32x4 c) {
#pragma unroll
    for (int s = 0; s < 4; ++s) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo[s], b.lo[s], c, 0, 0, 0);
#pragma unroll
    for (int s = 0; s < 4; ++s) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi[s], b.hi[s], c, 0, 0, 0);
    return c;
}

template <typename T, int WMT, int WNT>
__global__ __launch_bounds__(512) void conv_mfma_kernel(const ConvParams p) {
    using Cfg = ConvCfg<T, WMT, WNT>;
    constexpr int KO_T = Cfg::KO_T, PT = Cfg::PT, NT = Cfg::NT, WAVN = Cfg::WAVN;
    constexpr int EPP = ElemTraits<T>::kPieceElems;  // elements per 16-byte piece
    constexpr int PPP = kChunk / EPP;                // pieces per position (per row of 32 ch)
    constexpr int A_PIECES = KO_T * PPP;
    constexpr int AI = (A_PIECES + NT - 1) / NT;
    constexpr int BI = (Cfg::NPOS_CAP * PPP + NT - 1) / NT;
    constexpr int PLANE_PIECES = KO_T * 8 / EPP;  // 16-byte pieces per k-group plane of A

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* hdr = (int*)smem;  // hdr[0] = #subregions; subregion s at hdr[8 + 8*s ...]
    int* srctab = (int*)(smem + kHdrBytes);
    const int npos = p.npos;
    T* Abuf = (T*)(smem + kHdrBytes + ((npos * 4 + 15) & ~15));
    T* Bbuf = Abuf + 2 * KO_T * kChunk;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_m = wave / WAVN, wave_n = wave % WAVN;
    const int tile = blockIdx.x % p.num_pix_tiles;
    const int kt = blockIdx.x / p.num_pix_tiles;
    const int g0 = tile * PT;
    const int total = p.g.total_pix;

    // ---- subregions: for every sample this pixel tile touches, the board rows it covers plus
    // a one-cell zero halo; region s holds (rows+2) x (bs+2) positions starting at base_s.
    if (tid == 0) {
        const int g1 = min(g0 + PT, total);
        int lo = 0, hi = p.g.n_samples;  // last n with sample_off[n] <= g0
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

    for (int pos = tid; pos < npos; pos += NT) {
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

    // ---- per-lane pixel columns (B operand / output columns)
    int lpos[WNT], lstr[WNT], orow[WNT];
#pragma unroll
    for (int j = 0; j < WNT; ++j) {
        const int gi = g0 + (wave_n * WNT + j) * 16 + (lane & 15);
        lstr[j] = hdr[8 + 2] + 2;
        lpos[j] = lstr[j] + 1;  // an interior cell of region 0: every tap stays in range
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

    const int nchunks = p.cin_s / kChunk;
    const int taps = p.taps;
    const int nsteps = nchunks * taps;
    const T* __restrict__ gin = (const T*)p.in;
    const T* __restrict__ gw = (const T*)p.w;

    uint4 areg[AI], breg[BI];

    auto load_a = [&](int chunk, int tap) {
        const size_t ws = (size_t)tap * nchunks + chunk;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const int a = tid + i * NT;
            if (a < A_PIECES) {
                const int kg = a / PLANE_PIECES, within = a - kg * PLANE_PIECES;
                const T* src = gw + ((ws * 4 + kg) * p.ko_pad + (size_t)kt * KO_T) * 8 + within * EPP;
                areg[i] = *(const uint4*)src;
            }
        }
    };
    auto store_a = [&](int buf) {
        T* dst = Abuf + buf * KO_T * kChunk;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const int a = tid + i * NT;
            if (a < A_PIECES) *(uint4*)(dst + a * EPP) = areg[i];
        }
    };
    auto load_b = [&](int chunk) {
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            const int idx = tid + i * NT;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (idx < npos * PPP) {
                const int pos = idx / PPP, piece = idx - pos * PPP;
                const int src = srctab[pos];
                if (src >= 0) v = *(const uint4*)(gin + (size_t)src * p.cin_s + chunk * kChunk + piece * EPP);
            }
            breg[i] = v;
        }
    };
    auto store_b = [&](int buf) {
        T* dst = Bbuf + buf * npos * kChunk;
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            const int idx = tid + i * NT;
            if (idx < npos * PPP) {
                const int pos = idx / PPP, piece = idx - pos * PPP;
                const int el = piece * EPP, kg = el >> 3, within = el & 7;
                *(uint4*)(dst + ((size_t)kg * npos + pos) * 8 + within) = breg[i];
            }
        }
    };

    f32x4 acc[WMT][WNT];
#pragma unroll
    for (int i = 0; i < WMT; ++i)
#pragma unroll
        for (int j = 0; j < WNT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    load_a(0, 0);
    load_b(0);
    store_a(0);
    store_b(0);
    __syncthreads();

    const int kg = lane >> 4;
    const int arow0 = wave_m * WMT * 16 + (lane & 15);

    int chunk = 0, tap = 0;
    for (int step = 0; step < nsteps; ++step) {
        int nchunk = chunk, ntap = tap + 1;
        if (ntap == taps) { ntap = 0; nchunk = chunk + 1; }
        const bool has_next = step + 1 < nsteps;
        const bool fetch_b = (tap == 0) && (chunk + 1 < nchunks);
        if (has_next) load_a(nchunk, ntap);
        if (fetch_b) load_b(chunk + 1);

        const T* Ac = Abuf + (step & 1) * KO_T * kChunk + (kg * KO_T + arow0) * 8;
        const T* Bc = Bbuf + (chunk & 1) * npos * kChunk + (size_t)kg * npos * 8;
        int dy = 0, dx = 0;
        if (taps == 9) { dy = tap / 3 - 1; dx = tap - (tap / 3) * 3 - 1; }

        typename Frag<T>::type bf[WNT];
#pragma unroll
        for (int j = 0; j < WNT; ++j) bf[j] = lds_frag<T>(Bc + (lpos[j] + dy * lstr[j] + dx) * 8);
#pragma unroll
        for (int i = 0; i < WMT; ++i) {
            const typename Frag<T>::type af = lds_frag<T>(Ac + i * 16 * 8);
#pragma unroll
            for (int j = 0; j < WNT; ++j) acc[i][j] = mma(af, bf[j], acc[i][j]);
        }

        if (has_next) store_a((step + 1) & 1);
        if (fetch_b) store_b((chunk + 1) & 1);
        __syncthreads();
        chunk = nchunk;
        tap = ntap;
    }

    // ---- epilogue: D rows 4*(lane>>4)+r are 4 consecutive output channels of pixel lane&15
    T* __restrict__ gout = (T*)p.out;
    const T* __restrict__ gres = (const T*)p.res;
    const int act = p.act;
#pragma unroll
    for (int i = 0; i < WMT; ++i) {
        const int ko = kt * KO_T + (wave_m * WMT + i) * 16 + 4 * (lane >> 4);
        if (ko >= p.cout_s) continue;
        const f32x4 bias = *(const f32x4*)(p.bias + ko);
#pragma unroll
        for (int j = 0; j < WNT; ++j) {
            if (orow[j] < 0) continue;
            const size_t o = (size_t)orow[j] * p.cout_s + ko;
            f32x4 v = acc[i][j] + bias;
            if (gres) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += to_float(gres[o + r]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = activate(v[r], act);
            if constexpr (sizeof(T) == 2) {
                f16x4 h;
#pragma unroll
                for (int r = 0; r < 4; ++r) h[r] = (f16)v[r];
                *(f16x4*)(gout + o) = h;
            } else {
                *(f32x4*)(gout + o) = v;
            }
        }
    }
}

}  // namespace sayuri
