// conv_tower.h -- a RUN of consecutive board convolutions as ONE persistent launch.
//
// With one workgroup per board (conv_board.h) layer L+1 of a tile reads only what the SAME workgroup wrote in layer L
// (its halo is the tile's own zero frame, its residual rows are the tile's own rows): the dependency between two
// layers is workgroup-local.  So a workgroup can walk its tile through every 3x3 convolution of the tower alone --
// no grid barrier, no kernel boundary (the reference launches ~170 kernels per forward, cuda_forward_pipe.cc:713-981;
// this backend ran 41 + 2).  What a boundary cost: the drain of the slowest workgroup, ~1.3 us of dispatch, the cold
// start of the prologue (kernel arguments, the first weight group and the first halo tile requested from an empty
// pipeline by every CU at once).
//
// How the layer loop is built.  A C++ loop around the body does not survive hipcc: everything that is invariant from
// layer to layer is hoisted and kept alive across K loops whose 128 + 128 registers are all spoken for (150-540 vector
// spills in round 2, some of them of fragment registers of the hand-counted LDS pipeline -- wrong results, not only
// slow ones).  So the kernel below is compiled as a SINGLE-layer kernel whose by-value argument is one TowerLayer,
// and the loop is closed in the assembly hipcc emits for it (tower_seam.py, run by sayuri_amd/_build.py):
//   entry:   the real kernel argument is a pointer to a device array of TowerLayer; the seam parks that pointer, the
//            workgroup id and the wave id in SGPRs above the compiler's allocation and points s[0:1] (where the
//            compiled body expects its kernarg segment) at element 0;
//   s_endpgm -> seam:  s_waitcnt vmcnt(0) (this wave's stores have reached L2) + s_barrier (so have the other seven
//            waves'), then, unless the element says `last`, s[0:1] += sizeof(TowerLayer), the ABI's entry registers
//            (s2 = workgroup id, v0 = thread id, exec) are rebuilt and the wave branches back to the first compiled
//            instruction.  The compiled body never sees a loop, so it cannot hoist anything across one.
// Coherence: producer and consumer are the same CU; its vector L1 is write-through and shared by the workgroup's waves
// (LLVM AMDGPU memory model, workgroup scope, non-tgsplit mode), ordering comes from vmcnt(0) + s_barrier.
#pragma once
#include "conv_board.h"

namespace sayuri {

constexpr int kTowerStride = 320;  // bytes per TowerLayer in the device table (tower_seam.py: STRIDE)

// One convolution of the run: an element of the device table the launch walks.
struct alignas(16) TowerLayer {
    const TowerLayer* self;  // offset 0: the element's own address -- s[0:1] pointing AT an element is a valid kernarg
                             // segment for the compiled body, whose only argument is a pointer to the element
    int last;                // offset 8 (tower_seam.py: LAST_OFFSET): 1 = the run ends with this layer
    int has_se;              // the squeeze-and-excitation unit follows inside the kernel (sp.squeeze / sp.excite / sp.C valid)
    BoardSeParams sp;
    char pad[kTowerStride - 16 - sizeof(BoardSeParams)];
};
static_assert(sizeof(TowerLayer) == kTowerStride, "tower_seam.py steps the table by kTowerStride bytes");
static_assert(offsetof(TowerLayer, self) == 0 && offsetof(TowerLayer, last) == 8, "tower_seam.py: SELF at 0, LAST at 8");

// The table lives in device memory and is never written while a launch runs: the body reads it through the constant
// address space (scalar loads).  Such loads are invariant to the compiler, which would otherwise fetch every field up
// front and keep ~60 SGPRs alive across the K loop (spilled to VGPR lanes, and accumulators to scratch for those);
// what the SE stage and the epilogue need is therefore read through a pointer the compiler cannot see through,
// AFTER the main loop.
typedef const __attribute__((address_space(4))) TowerLayer* TowerLayerCP;
__device__ __forceinline__ TowerLayerCP tower_launder(TowerLayerCP p) {
    asm volatile("" : "+s"(p));
    return p;
}

// ---- the SE unit inside the run ----------------------------------------------------------------------------------
// Compiled in one piece with the convolution (rounds 2-3: a second body, conv_tower_kernel<WMT, true>) hipcc re-assigned the
// 192 accumulator registers after the K loop -- ~60 of the AGPR values copied to VGPRs, six tiles parked in scratch, the
// pooling reduction three times the instructions it needs -- and the unit cost +17 us per layer.  Now there is ONE convolution
// body.  Right behind its K loop sits a HOOK: an asm statement that names the table element and the thread id as its only
// operands and clobbers the registers the K loop's transients lived in (kTowerFreeVgpr.., kTowerFreeSgprs).  tower_seam.py
// replaces the hook by
//     has_se ?  pooling (generated assembly, on the accumulators where the K loop left them: the script reads the register of
//               every output tile off the MFMA stream)  ->  the two FCs (tower_se_fc_kernel below: a compiled body of its own
//               that only sees LDS, entered like a subroutine with its VGPRs renamed into the clobbered range)  ->  the gate
//               (generated assembly, in place)  :  nothing
// and the compiled epilogue follows, for which the accumulators simply have other values.  No scratch, no accumulator moves.
constexpr int kTowerFreeVgpr = 66;   // v[66:127] are the hook's: the K loop keeps its accumulator tiles below (the build checks)
constexpr int kTowerFreeSgprs = 64;  // s[0:63] are the hook's
#define SAYURI_TOWER_CLOBBER_V                                                                                                            \
    "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84",   \
        "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102",  \
        "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118",   \
        "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127"
#define SAYURI_TOWER_CLOBBER_S                                                                                                            \
    "s0", "s1", "s2", "s3", "s4", "s5", "s6", "s7", "s8", "s9", "s10", "s11", "s12", "s13", "s14", "s15", "s16", "s17", "s18", "s19",     \
        "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35", "s36", "s37",     \
        "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55",     \
        "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63"

// One asm statement per accumulator tile: "; TOWER_ACC <side> <row tile> <column tile> <register>".
template <int WMT, int SIDE, int K> __device__ __forceinline__ void tower_anchor_tile(f32x4& t) {
    constexpr int i = K % WMT, j = K / WMT;
    if constexpr (K < 32) asm volatile("; TOWER_ACC %1 %2 %3 %0" : "+a"(t) : "n"(SIDE), "n"(i), "n"(j));
    else asm volatile("; TOWER_ACC %1 %2 %3 %0" : "+v"(t) : "n"(SIDE), "n"(i), "n"(j));
}
template <int WMT, int SIDE> __device__ __forceinline__ void tower_anchor(f32x4 (&acc)[WMT][kBoardNJ]) {
    static_for<WMT * kBoardNJ>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        tower_anchor_tile<WMT, SIDE, k>(acc[k % WMT][k / WMT]);
    });
}

// The compiled convolution body (tile = workgroup, one channel tile): K loop, hook, epilogue.
template <int WMT>
__global__ __launch_bounds__(512, 2) void conv_tower_kernel(const TowerLayer* layer) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const TowerLayerCP L = (TowerLayerCP)layer;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x;
    const int ui = L->sp.b.uniform_info;
    const int info = ui >= 0 ? ui : __builtin_amdgcn_readfirstlane(L->sp.b.tab_cols[tile]);
    const int ncols = info & 0xff, bs = info >> 8;
    const int nj0 = (ncols + 1) >> 1;
    const int wave_n = wave >> 2;
    const int col0 = wave_n ? nj0 : 0;
    const int nj = wave_n ? ncols - nj0 : nj0;

    f32x4 acc[WMT][kBoardNJ];
    {
        const BoardParams& bp = *(const BoardParams*)&L->sp.b;
        board_mainloop<WMT, false, 2>(bp, smem, acc, tile, 0, wave, lane, col0, nj == kBoardNJ, bs, nullptr,
                                      (const __attribute__((address_space(4))) BoardParams*)&L->sp.b);
    }
    // The hook (see above; the operand list is what tower_seam.py parses).  Everything the epilogue needs is derived again from
    // (table element, thread id, workgroup id) behind it: nothing but those three stays alive across the K loop.
    // The anchors in front of it name every accumulator tile as a read-write operand of its K-loop register class and print its
    // register: the tiles are in THOSE registers at the hook and whatever copy the compiler made earlier is dead.  tower_seam.py
    // fails the build if any instruction sits between the first anchor and the hook.
    TowerLayerCP L2 = L;
    int tid2 = tid;
    tower_anchor<WMT, 0>(acc);
    asm volatile("; TOWER_SE_HOOK elem=%0 tid=%1 wmt=%2 ui=%3 cols=%4 w1h=%5 w2h=%6 w1b=%7 w2b=%8 psum=%9 pmax=%10 gate=%11 kot=%12 "
                 "res=%13 out=%14 couts=%15 slotpix=%16 act=%17 arith=%18 mish=%19 roword=%20 relu=%21 identity=%22"
                 : "+s"(L2), "+v"(tid2)
                 : "n"(WMT), "n"(offsetof(TowerLayer, sp.b.uniform_info)), "n"(offsetof(TowerLayer, sp.b.tab_cols)),
                   "n"(offsetof(TowerLayer, sp.w1h)), "n"(offsetof(TowerLayer, sp.w2h)), "n"(offsetof(TowerLayer, sp.w1_bytes)),
                   "n"(offsetof(TowerLayer, sp.w2_bytes)), "n"(SeLds<WMT>::psum), "n"(SeLds<WMT>::pmax), "n"(SeLds<WMT>::gate),
                   "n"(BoardCfg<WMT>::KO_T), "n"(offsetof(TowerLayer, sp.b.c.res)), "n"(offsetof(TowerLayer, sp.b.c.out)),
                   "n"(offsetof(TowerLayer, sp.b.c.cout_s)), "n"(offsetof(TowerLayer, sp.b.c.g.slot_pix)), "n"(offsetof(TowerLayer, sp.b.c.act)),
                   "n"(offsetof(TowerLayer, sp.b.arith)), "n"((int)kMish), "n"(offsetof(TowerLayer, sp.b.row_order)), "n"((int)kReLU), "n"((int)kIdentity)
                 : "memory", "vcc", "scc", SAYURI_TOWER_CLOBBER_V, SAYURI_TOWER_CLOBBER_S);
    const BoardSeParams& sp = *(const BoardSeParams*)&L2->sp;
    const BoardParams& bp = sp.b;
    const int lane2 = tid2 & 63, wave2 = __builtin_amdgcn_readfirstlane(tid2 >> 6);
    const int ui2 = bp.uniform_info;
    const int info2 = ui2 >= 0 ? ui2 : __builtin_amdgcn_readfirstlane(bp.tab_cols[tile]);
    const int ncols2 = info2 & 0xff, nj02 = (ncols2 + 1) >> 1;
    const int col02 = (wave2 >> 2) ? nj02 : 0, nj2 = (wave2 >> 2) ? ncols2 - nj02 : nj02;
    switch (bp.c.act) {
    case kMish: board_epilogue<WMT, kMish>(bp, smem, acc, tile, 0, wave2, lane2, col02, nj2); break;
    case kIdentity: board_epilogue<WMT, kIdentity>(bp, smem, acc, tile, 0, wave2, lane2, col02, nj2); break;
    case kReLU: board_epilogue<WMT, kReLU>(bp, smem, acc, tile, 0, wave2, lane2, col02, nj2); break;
    case kSwish: board_epilogue<WMT, kSwish>(bp, smem, acc, tile, 0, wave2, lane2, col02, nj2); break;
    case kELU: board_epilogue<WMT, kELU>(bp, smem, acc, tile, 0, wave2, lane2, col02, nj2); break;
    case kSELU: board_epilogue<WMT, kSELU>(bp, smem, acc, tile, 0, wave2, lane2, col02, nj2); break;
    case kGELU: board_epilogue<WMT, kGELU>(bp, smem, acc, tile, 0, wave2, lane2, col02, nj2); break;
    default: board_epilogue<WMT, kHardSwish>(bp, smem, acc, tile, 0, wave2, lane2, col02, nj2); break;
    }
}

// The FCs of the SE unit as a body of their own: pooled partials in LDS -> gate in LDS (board_se_fc, conv_board.h).  Entered
// from the hook like a subroutine (s[0:1] = the element, s2 = workgroup id, v0 = thread id), every s_endpgm returns there.
// It may use neither AGPRs nor scratch nor more than 128 - kTowerFreeVgpr VGPRs / kTowerFreeSgprs SGPRs: tower_seam.py checks.
template <int WMT>
__global__ __launch_bounds__(512) void tower_se_fc_kernel(const TowerLayer* layer) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, tile = blockIdx.x;
    const BoardSeParams& sp = layer->sp;
    const int ui = sp.b.uniform_info;
    const int info = ui >= 0 ? ui : __builtin_amdgcn_readfirstlane(sp.b.tab_cols[tile]);
    board_se_fc<WMT, 4>(sp, smem, tid, info >> 8);
}

}  // namespace sayuri
