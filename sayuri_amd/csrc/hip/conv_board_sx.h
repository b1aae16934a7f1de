// conv_board_sx.h -- a block's last 3x3 convolution WITH its squeeze-and-excitation unit when the layer's channels are SPLIT over
// several workgroups (384 channels = three 128-channel tiles per board tile; reference SEUnit::Forward, se_unit.cc:70-128,
// cuda_layers.cc:957-1039).
//
// conv_board.h's fused form needs the whole layer in one workgroup (KO_T = C).  Here a board tile's kts sibling workgroups each
// hold 128 channels of x in their accumulators, and what the unit needs across channels is small:
//   pool        per (sample, channel): local to the workgroup that holds the channel
//   squeeze FC  mid = act(b1 + W1 pooled): a SUM OVER CHANNELS -> each sibling computes the partial sum over ITS channels
//               (se values per sample) and the siblings EXCHANGE the partials through global memory: <= 4 x 96 floats each
//   excite FC   gamma / beta of a workgroup's own channels need all of mid and only the own rows of W2: local again
//   gate        x <- sigmoid(gamma) x + beta on the accumulators, then the ordinary epilogue (residual, activation, store)
// so x never makes the round trip through HBM and the unit's three launches (se_pool / se_fc / se_scale: 58 us per unit on
// configs[4]'s batch, 7 % of the forward) are gone.  What the stage costs instead: ~12 us per workgroup (profiles/r06_split_se_ab.txt).
//
// The exchange follows cdna_hip_programming.md section 6 Guideline 16, form R2 ("the data is the flag"): every partial is ONE
// 8-byte granule {tag = the launch's epoch, value} written by one relaxed agent-scope store (sc1, write-through) and polled by
// relaxed agent-scope loads until the tag matches -- no fence, no separate flag, nothing to zero between launches (the epoch
// grows by one per SE layer launch and a slot's old tag never equals a later epoch).  The sum over the siblings is taken in
// channel-tile order 0, 1, 2 by every sibling, so all of them compute the same mid.
//
// A tile may hold several samples of one size (two 13x13, four 9x9): the pooling is per SAMPLE and must not depend on where in
// the tile the sample sits (a request's result is a function of the request, batch_forward_pipe.cc:15-33).  So it is not done on
// the accumulator layout: the tile's accumulators go through LDS as [pixel slot][channel] fp32 (64 channels at a time), and
// wave e adds the e-th eighth of every sample's pixels IN PIXEL ORDER (lane = channel); the eight eighths are folded in
// order.  The order of every sum is a function of the sample's own size alone.
//
// Progress: a workgroup waits for its siblings inside the launch, so the siblings must get to run.  Block index =
// 8 kts g + 8 kt + (tile & 7) for tile = 8 g + (tile & 7): the siblings of a tile are 8 apart, i.e. on the same XCD's queue
// (workgroups go to XCDs round-robin) and next to each other in it; with workgroups dispatched in index order, per XCD or
// globally, a resident workgroup waits for at most the few siblings at the dispatch frontier, and every complete tile runs
// to its end and frees its CUs.  The wait is bounded all the same: after ~0.3 s a workgroup gives up, says so in a
// host-visible word (the engine fails the forward loudly) and leaves.
#pragma once
#include "conv_board.h"

namespace sayuri {

constexpr int kSxMaxSub = 4;     // samples per tile this kernel pools (boards of 9x9 and larger)
constexpr int kSxSlots = 128;    // granules per (tile, channel tile, sample): se <= 128
typedef __attribute__((address_space(1))) unsigned long long sx_gu64;

struct BoardSxParams {
    BoardParams b;        // c.res / c.act are the UNIT's residual and activation; c.npos = first tile, c.num_pix_tiles = tiles of the launch
    const void* w1t;      // squeeze images: [kt][board size 2..][2 x 128 rows][se] fp16, mean rows (scaled mean folded in) then max rows 
    const void* w2t;      // excite images:  [kt][se / 4][2 x 128][4] fp16, then excite bias of the own 2 x 128 outputs, then squeeze bias [se] (fp32)
    int w1_bytes, w2_bytes;  // bytes of one image (multiples of 1 KiB)
    int nsizes;           // board sizes per channel tile in w1t (board - 1)
    int se, kts;
    unsigned long long* xchg;  // granules [tile][kt][kSxMaxSub][kSxSlots]
    unsigned epoch;       // tag of this launch's granules (never 0)
    unsigned* err;        // host-visible: set to epoch when a wait for the siblings ran out
    int dbg_stall;        // SAYURI_DEBUG_SX_STALL=1 (tests): channel tile 1 publishes under a wrong tag -- its siblings' wait must run out
};

// LDS map of the stage (the K loop's rings are dead): [0, kStage) the accumulators of 64 channels as [slot][68] fp32 -- later
// the two weight images -- then the small arrays.
struct SxLds {
    static constexpr int kPitch = 68;                                  // floats per pixel slot (64 + 4: the 16 pixel lanes of a store hit different banks)
    static constexpr int stage = 0, stage_bytes = kBoardPT * kPitch * 4;   // 104 448
    static constexpr int psum = stage + stage_bytes;                   // [sample][eighth][128] partial sums
    static constexpr int pmax = psum + kSxMaxSub * 8 * 128 * 4;        // ... and maxima
    // (the eighths are dead once folded: the squeeze FC's row-slice partials lie over them)
    static constexpr int red = psum;                                   // [sample][parts][se], parts * se <= 2048
    static constexpr int pool = red + kSxMaxSub * 2048 * 4;            // [sample][256]: mean of the 128 own channels, then their maxima
    static constexpr int part = pool + kSxMaxSub * 256 * 4;            // [kt <= 4][sample][128]: the siblings' partial sums (own included)
    static constexpr int mid = part + 4 * kSxMaxSub * 128 * 4;         // [sample][128]
    static constexpr int gate = mid + kSxMaxSub * 128 * 4;             // [sample][256]: sigmoid(gamma) of the own channels, then beta
    static constexpr int end = gate + kSxMaxSub * 256 * 4;
};
static_assert(SxLds::end <= 160 * 1024, "the SE stage must fit the LDS");

// LDS traffic of this wave retired, then the workgroup barrier -- one statement, so that nothing can be scheduled between the two
// (no vmcnt: the weight loads and the LDS-DMA in flight stay in flight across it)
__device__ __forceinline__ void sx_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int WMT>
__global__ __launch_bounds__(512, 2) void conv_board_sx_kernel(const BoardSxParams sp) {
    static_assert(WMT == 2, "128-channel tiles");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using Cfg = BoardCfg<WMT>;
    constexpr int NJ = Cfg::NJ;
    const BoardParams& bp = sp.b;
    const ConvParams& p = bp.c;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kts = sp.kts;
    // siblings 8 apart (see the head of the file)
    const int grp = blockIdx.x / (8 * kts), rem = blockIdx.x - grp * 8 * kts;
    const int kt = rem >> 3, tl = grp * 8 + (rem & 7);
    if (tl >= p.num_pix_tiles) return;  // the last group's empty places: nobody waits for them
    const int tile = p.npos + tl;
    const int info = bp.uniform_info >= 0 ? bp.uniform_info : __builtin_amdgcn_readfirstlane(bp.tab_cols[tile]);
    const int ncols = info & 0xff, bs = info >> 8;
    const int nj0 = (ncols + 1) >> 1;
    const int wave_m = wave & 3, wave_n = wave >> 2;
    const int col0 = wave_n ? nj0 : 0;
    const int nj = wave_n ? ncols - nj0 : nj0;
    const int npix = bs * bs;
    int nsub = (ncols * 16) / npix;  // samples of the tile (npix >= 81 > 16: the padding of the last column tile never reaches a whole sample)
    nsub = nsub < 1 ? 1 : (nsub > kSxMaxSub ? kSxMaxSub : nsub);

    // timeline (measuring runs, SAYURI_SX_DBG=n: the n-th SE layer of the forward): wave 0 of the launch's blocks 0, 8, 16 (the
    // three siblings of its first tile) and 1: [0] start, [1] K loop done, [2] pooled, [3] partials published, [4] siblings'
    // partials in, [5] gate in LDS, [6] gate applied, [7] end
    unsigned long long* dbg = nullptr;
    if (bp.dbg && tid == 0 && (blockIdx.x == 0 || blockIdx.x == 8 || blockIdx.x == 16 || blockIdx.x == 1)) {
        dbg = bp.dbg + (size_t)(blockIdx.x == 1 ? 3 : blockIdx.x >> 3) * 64;
        dbg[0] = __builtin_amdgcn_s_memtime();
    }
    f32x4 acc[WMT][kBoardNJ];
    board_mainloop<WMT>(bp, smem, acc, tile, kt, wave, lane, col0, nj == kBoardNJ, bs);
    if (dbg) dbg[1] = __builtin_amdgcn_s_memtime();

    float* stage = (float*)(smem + SxLds::stage);
    float* psum = (float*)(smem + SxLds::psum);
    float* pmax = (float*)(smem + SxLds::pmax);
    float* pool = (float*)(smem + SxLds::pool);
    float* red = (float*)(smem + SxLds::red);
    float* part = (float*)(smem + SxLds::part);
    float* mid = (float*)(smem + SxLds::mid);
    float* gate = (float*)(smem + SxLds::gate);
    const int q = lane >> 4, px = lane & 15;
    const int se = sp.se;

    const int quads = se >> 2, parts = min(32, 512 / quads);  // squeeze FC: thread = (4 consecutive outputs, every parts-th own row); parts * se <= 2048
    const int oq = tid % quads, pt = tid / quads;

    // ---- 1. pooling, 64 channels at a time through LDS
    sx_barrier();  // every wave is done with the rings
    const int per = (npix + 7) >> 3;  // pixels per eighth of a sample
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if ((wave_m >> 1) == half) {
            static_for<NJ>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                if (j < nj) {  // wave-uniform
                    const int slot = (col0 + j) * 16 + px;
#pragma unroll
                    for (int i = 0; i < WMT; ++i)
                        *(f32x4*)(stage + slot * SxLds::kPitch + (wave_m & 1) * 32 + i * 16 + 4 * q) = acc[i][j];
                }
            });
        }
        sx_barrier();
        {
            // wave = eighth e of every sample, lane = channel c of the half: the pixels of the eighth IN ORDER (eight reads in
            // flight, the additions in sequence)
            const int c = lane, e = wave;
            const int p0 = e * per, p1 = min(npix, p0 + per);
            for (int s = 0; s < nsub; ++s) {
                const float* src = stage + (s * npix + p0) * SxLds::kPitch + c;
                float a = 0.f, m = -5000.f;
                int pp = p0;
                for (; pp + 8 <= p1; pp += 8) {
                    float v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = src[u * SxLds::kPitch];
                    src += 8 * SxLds::kPitch;
#pragma unroll
                    for (int u = 0; u < 8; ++u) { a += v[u]; m = max_raw(m, v[u]); }
                }
                for (; pp < p1; ++pp) {
                    const float v = *src;
                    src += SxLds::kPitch;
                    a += v;
                    m = max_raw(m, v);
                }
                psum[(s * 8 + e) * 128 + half * 64 + c] = a;
                pmax[(s * 8 + e) * 128 + half * 64 + c] = m;
            }
        }
        sx_barrier();
    }

    // ---- 2. the two images of this channel tile into the (now free) stage area; the eighths folded in order meanwhile
    {
        const int n1 = sp.w1_bytes >> 10, n2 = sp.w2_bytes >> 10;
        const unsigned char* g1 = (const unsigned char*)sp.w1t + ((size_t)kt * sp.nsizes + (bs - 2)) * sp.w1_bytes;
        const unsigned char* g2 = (const unsigned char*)sp.w2t + (size_t)kt * sp.w2_bytes;
        const uint32_t l0 = (uint32_t)(uintptr_t)smem;
        for (int k = wave; k < n1; k += 8) glds16_s(lane * 16, g1 + k * 1024, l0 + k * 1024);
        for (int k = wave; k < n2; k += 8) glds16_s(lane * 16, g2 + k * 1024, l0 + sp.w1_bytes + k * 1024);
    }
    {
        const float inv = 1.0f / (float)npix;
        for (int k = tid; k < nsub * 128; k += 512) {
            const int s = k >> 7, c = k & 127;
            const float* a = psum + s * 8 * 128 + c;
            const float* m = pmax + s * 8 * 128 + c;
            float t = a[0], x = m[0];
#pragma unroll
            for (int e = 1; e < 8; ++e) { t += a[e * 128]; x = max_raw(x, m[e * 128]); }
            pool[s * 256 + c] = t * inv;
            pool[s * 256 + 128 + c] = x;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of the two images have landed
    sx_barrier();

    if (dbg) dbg[2] = __builtin_amdgcn_s_memtime();
    // ---- 3. squeeze FC over the own 2 x 128 rows: thread = (4 consecutive outputs, every parts-th row), all samples at once
    {
        if (pt < parts) {
            const unsigned char* w1 = smem + oq * 8;
            f32x4 a[kSxMaxSub];
#pragma unroll
            for (int s = 0; s < kSxMaxSub; ++s) a[s] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
            for (int r = pt; r < 256; r += parts) {
                const f16x4 w = *(const f16x4*)(w1 + (size_t)r * se * 2);
                const f32x4 wf = {(float)w[0], (float)w[1], (float)w[2], (float)w[3]};
#pragma unroll
                for (int s = 0; s < kSxMaxSub; ++s)
                    if (s < nsub) a[s] += pool[s * 256 + r] * wf;
            }
#pragma unroll
            for (int s = 0; s < kSxMaxSub; ++s)
                if (s < nsub) *(f32x4*)(red + (s * parts + pt) * se + oq * 4) = a[s];
        }
    }
    sx_barrier();
    // the row slices folded in order; the result is published: one granule {epoch, value} per (sample, output)
    sx_gu64* xg = (sx_gu64*)sp.xchg + ((size_t)tile * kts) * (kSxMaxSub * kSxSlots);
    for (int k = tid; k < nsub * se; k += 512) {
        const int s = k / se, o = k - s * se;
        float t = 0.f;
        for (int pt = 0; pt < parts; ++pt) t += red[(s * parts + pt) * se + o];
        part[(kt * kSxMaxSub + s) * 128 + o] = t;
        const unsigned tag = (sp.dbg_stall && kt == 1) ? (sp.epoch ^ 0x80000000u) : sp.epoch;
        __hip_atomic_store(xg + ((size_t)kt * kSxMaxSub + s) * kSxSlots + o, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(t),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (dbg) dbg[3] = __builtin_amdgcn_s_memtime();
    // ---- 4. the siblings' partials: every thread polls its own granules until they carry this launch's tag
    {
        const int want = (kts - 1) * nsub * se;
        bool gave_up = false;
        for (int k = tid; k < want; k += 512) {
            const int sib = k / (nsub * se), r2 = k - sib * nsub * se;
            const int s = r2 / se, o = r2 - s * se;
            const int okt = sib + (sib >= kt ? 1 : 0);
            sx_gu64* g = xg + ((size_t)okt * kSxMaxSub + s) * kSxSlots + o;
            unsigned long long v = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned spins = 0;
            while ((unsigned)(v >> 32) != sp.epoch) {
                if (++spins > (1u << 18)) { gave_up = true; break; }
                __builtin_amdgcn_s_sleep(8);
                v = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            part[(okt * kSxMaxSub + s) * 128 + o] = __uint_as_float((unsigned)v);
        }
        if (gave_up) __hip_atomic_store((__attribute__((address_space(1))) unsigned*)sp.err, sp.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    sx_barrier();
    if (dbg) dbg[4] = __builtin_amdgcn_s_memtime();
    // mid = act(b1 + p_0 + p_1 + ...): the same order in every sibling
    const unsigned char* w2 = smem + sp.w1_bytes;
    const float* b2 = (const float*)(w2 + (size_t)se * 256 * 2);  // excite bias of the own outputs
    const float* b1 = b2 + 256;                                    // squeeze bias
    for (int k = tid; k < nsub * se; k += 512) {
        const int s = k / se, o = k - s * se;
        float t = b1[o];
        for (int okt = 0; okt < kts; ++okt) t += part[(okt * kSxMaxSub + s) * 128 + o];
        mid[s * 128 + o] = activate(t, p.act);
    }
    sx_barrier();
    // ---- 5. excite FC, the own 2 x 128 outputs: thread = (output, sample parity), 4 inputs per 8-byte read
    {
        const int o = tid & 255;
        for (int s = tid >> 8; s < nsub; s += 2) {
            float t = b2[o];
            for (int i = 0; i < se; i += 4) {
                const f16x4 w = *(const f16x4*)(w2 + ((size_t)(i >> 2) * 256 + o) * 8);
                const f32x4 m = *(const f32x4*)(mid + s * 128 + i);
                t += m[0] * (float)w[0] + m[1] * (float)w[1] + m[2] * (float)w[2] + m[3] * (float)w[3];
            }
            gate[s * 256 + o] = o < 128 ? 1.0f / (1.0f + fast_exp(-t)) : t;
        }
    }
    sx_barrier();
    if (dbg) dbg[5] = __builtin_amdgcn_s_memtime();
    // ---- 6. the gate on the accumulators: x <- sigmoid(gamma) x + beta, gamma / beta of the pixel's sample
#pragma unroll
    for (int i = 0; i < WMT; ++i) {
        const int c0 = wave_m * WMT * 16 + i * 16 + 4 * q;
        if (nsub == 1) {  // (wave-uniform) one sample: one gate for every column tile
            const f32x4 g = *(const f32x4*)(gate + c0), be = *(const f32x4*)(gate + 128 + c0);
            static_for<NJ>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = __builtin_fmaf(g[r], acc[i][j][r], be[r]);
            });
        } else {
            static_for<NJ>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                if (j < nj) {
                    const int slot = (col0 + j) * 16 + px;
                    int s = (slot >= npix ? 1 : 0) + (slot >= 2 * npix ? 1 : 0) + (slot >= 3 * npix ? 1 : 0);
                    s = s < nsub ? s : nsub - 1;  // unused pixel slots behind the last sample: computed, never stored
                    const f32x4 g = *(const f32x4*)(gate + s * 256 + c0), be = *(const f32x4*)(gate + s * 256 + 128 + c0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][r] = __builtin_fmaf(g[r], acc[i][j][r], be[r]);
                }
            });
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // gate fully read before the epilogue's residual pieces land in the same LDS
    if (dbg) dbg[6] = __builtin_amdgcn_s_memtime();

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
    if (dbg) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        dbg[7] = __builtin_amdgcn_s_memtime();
    }
}

}  // namespace sayuri
