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

// The compiled bodies (tile = workgroup, one channel tile): SE = false is conv_board_kernel, SE = true
// conv_board_se_kernel.  Two bodies, not one with a flag: behind a run-time `if` the SE stage makes hipcc park 22
// accumulator tiles in scratch (748 bytes per lane); the seam picks the body by the element's has_se.  The launch
// enters through the SE = false kernel (its descriptor carries the resources of both).
template <int WMT, bool SE>
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
        board_mainloop<WMT, false, SE ? 1 : 2>(bp, smem, acc, tile, 0, wave, lane, col0, nj == kBoardNJ, bs, nullptr,
                                         (const __attribute__((address_space(4))) BoardParams*)&L->sp.b);
    }
    // everything the SE stage and the epilogue need is derived again from (table element, thread id, workgroup id)
    // behind an opaque point: nothing but those three stays alive across the K loop
    const TowerLayerCP L2 = tower_launder(L);
    int tid2 = tid;
    asm volatile("" : "+v"(tid2));
    const BoardSeParams& sp = *(const BoardSeParams*)&L2->sp;
    const BoardParams& bp = sp.b;
    const int lane2 = tid2 & 63, wave2 = __builtin_amdgcn_readfirstlane(tid2 >> 6);
    const int ui2 = bp.uniform_info;
    const int info2 = ui2 >= 0 ? ui2 : __builtin_amdgcn_readfirstlane(bp.tab_cols[tile]);
    const int ncols2 = info2 & 0xff, bs2 = info2 >> 8, nj02 = (ncols2 + 1) >> 1;
    const int col02 = (wave2 >> 2) ? nj02 : 0, nj2 = (wave2 >> 2) ? ncols2 - nj02 : nj02;
    if constexpr (SE) {
        board_se_stage<WMT>(sp, smem, acc, tile, wave2, lane2, col02, nj2, bs2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // gate fully read before the epilogue's residual pieces land in the same LDS
    }
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

}  // namespace sayuri
