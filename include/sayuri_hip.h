/* include/sayuri_hip.h -- C-ABI of the MI355X (gfx950) forward-pipe engine.
 *
 * This is the drop-in boundary for Sayuri's NN hot path.  In the reference the path sits
 * behind `class NetworkForwardPipe` (reference src/neural/network_basic.h:132-161) and, for
 * GPU backends, `BatchForwardPipe::BatchForward(int gpu, const std::vector<InputData>&)`
 * (reference src/neural/batch_forward_pipe.h:27-28), implemented for CUDA by
 * `CudaForwardPipe::NNGraph` (reference src/neural/cuda/cuda_forward_pipe.cc:133-1090).
 * A `HipForwardPipe` (sayuri_amd/csrc/host/hip_forward_pipe.h, or the stub shown in
 * INTEGRATION.md inside the reference tree) owns one `sayuri_hip_ctx` per GPU and calls the
 * entry points below; no HIP or torch type crosses this header.
 *
 * Conventions: plain C, pointers + sizes, return 0 on success / -1 on failure with the
 * message in sayuri_hip_last_error() (thread-local); no exception crosses the ABI
 * (the C++ pipe turns -1 into std::runtime_error like reference cuda_common.cc:46-62).
 * One ctx per GPU; a ctx is driven by ONE caller thread at a time (the pump thread).
 */
#ifndef SAYURI_HIP_H
#define SAYURI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sayuri_hip_ctx sayuri_hip_ctx;

/* reference src/neural/activation.h:8-17 (enum class Activation) */
enum {
    SAYURI_ACT_IDENTITY = 0, SAYURI_ACT_RELU = 1, SAYURI_ACT_ELU = 2, SAYURI_ACT_SELU = 3,
    SAYURI_ACT_GELU = 4, SAYURI_ACT_MISH = 5, SAYURI_ACT_SWISH = 6, SAYURI_ACT_HARDSWISH = 7
};

/* reference src/neural/description.h:92-93 (BlockBasic::Type) */
enum {
    SAYURI_BLOCK_RESIDUAL = 1, SAYURI_BLOCK_BOTTLENECK = 2, SAYURI_BLOCK_NESTED_BOTTLENECK = 3,
    SAYURI_BLOCK_MIXER = 4
};

/* One tower block: mirrors the shape fields of reference description.h:90-137 (BlockBasic). */
typedef struct {
    int32_t type;                 /* SAYURI_BLOCK_* */
    int32_t apply_se;             /* BlockBasic::apply_se */
    int32_t se_size;              /* BlockBasic::se_size */
    int32_t bottleneck_channels;  /* BlockBasic::bottleneck_channels (0 if n/a) */
    int32_t feedforward_channels; /* BlockBasic::feedforward_channels (0 if n/a) */
    int32_t dw_filter;            /* dw_conv.GetFilter() for mixer blocks (0 if n/a) */
} sayuri_hip_blockdesc;

/* The architecture part of reference description.h:164-215 (DNNWeights). */
typedef struct {
    int32_t version;                  /* DNNWeights::version (1..5) */
    int32_t input_channels;           /* 43 (v3+) or 38 */
    int32_t residual_channels;
    int32_t residual_blocks;
    int32_t policy_head_channels;
    int32_t value_head_channels;
    int32_t probabilities_channels;   /* 5 (v3+) or 1 */
    int32_t pass_probability_outputs; /* 5 (v3+) or 1 */
    int32_t ownership_channels;       /* 1 */
    int32_t value_misc_outputs;       /* 15 (v3+) or 5 */
    int32_t default_act;              /* SAYURI_ACT_* */
    int32_t policy_head_type;         /* 0 = kNormal, 1 = kRepLK */
    int32_t policy_dw_filter;         /* p_dw_conv.GetFilter() when RepLK, else 0 */
    const sayuri_hip_blockdesc* blocks; /* [residual_blocks] */
} sayuri_hip_netdesc;

/* Layer ids for sayuri_hip_load_tensor: the named members of DNNWeights / BlockBasic. */
enum {
    SAYURI_L_INPUT_CONV = 0, SAYURI_L_P_HD_CONV = 1, SAYURI_L_P_DW_CONV = 2, SAYURI_L_P_PT_CONV = 3,
    SAYURI_L_P_INTER_FC = 4, SAYURI_L_PROB_CONV = 5, SAYURI_L_PASS_FC = 6, SAYURI_L_V_HD_CONV = 7,
    SAYURI_L_V_INTER_FC = 8, SAYURI_L_V_OWNERSHIP = 9, SAYURI_L_V_MISC = 10,
    SAYURI_L_BLOCK_BASE = 16 /* block b, slot s -> 16 + 16*b + s */
};
enum { /* slots inside a block */
    SAYURI_S_CONV1 = 0, SAYURI_S_CONV2 = 1, SAYURI_S_CONV3 = 2, SAYURI_S_CONV4 = 3,
    SAYURI_S_PRE_BTL = 4, SAYURI_S_POST_BTL = 5, SAYURI_S_DW_CONV = 6, SAYURI_S_SQUEEZE = 7,
    SAYURI_S_EXCITE = 8
};
#define SAYURI_L_BLOCK(b, slot) (SAYURI_L_BLOCK_BASE + 16 * (b) + (slot))
enum { SAYURI_T_WEIGHTS = 0, SAYURI_T_BIASES = 1 };

/* Number of visible gfx950 devices (replaces reference src/utils/probe_gpu.h:10-27 GetGpuCount). */
int sayuri_hip_device_count(void);

/* Build the per-GPU graph: replaces NNGraph::ConstructGraph (cuda_forward_pipe.cc:133-613).
 * `board` is the NN board size (ForwardPipeOption::board_size), `max_batch` the largest batch
 * BatchForward will be handed, `use_fp16` mirrors option "fp16" (1: fp16 storage + MFMA with
 * fp32 accumulation; 0: fp32 storage + fp32 MFMA, the strict-parity mode). NULL on failure. */
sayuri_hip_ctx* sayuri_hip_create(int device, const sayuri_hip_netdesc* desc, int max_batch,
                                  int board, int use_fp16);

/* Hand one host tensor of DNNWeights to the device (replaces the LoadWeights calls of
 * cuda_layers.cc:621-716).  Tensors are the BN-folded fp32 arrays the reference loader
 * produces: conv weights [K][C][k][k] (depthwise [C][1][k][k]), fc weights [out][in],
 * biases [K].  `n` must equal the element count implied by the netdesc. */
int sayuri_hip_load_tensor(sayuri_hip_ctx* ctx, int layer_id, int kind, const float* host, size_t n);

/* One batch through the net: replaces NNGraph::BatchForward (cuda_forward_pipe.cc:684-1018)
 * up to, not including, FillOutputs.
 *   planes      [n][input_channels][board*board]  each sample already re-padded top-left into
 *               the NN grid as BatchForwardPipe::SendQueryAndWait does (batch_forward_pipe.cc:15-33)
 *   board_sizes [n]  the sample's own board size (InputData::board_size), NULL = all `board`
 *   prob [n][probabilities_channels][board*board], pass [n][pass_probability_outputs],
 *   misc [n][value_misc_outputs], own [n][board*board]   raw (pre-softmax/tanh) outputs in the
 *               NN grid; off-board entries of smaller samples are 0.
 * Blocking; host pointers. */
int sayuri_hip_forward(sayuri_hip_ctx* ctx, int n, const float* planes, const int* board_sizes,
                       float* prob, float* pass, float* misc, float* own);

/* The three phases of sayuri_hip_forward, for callers that keep inputs resident in HBM
 * (bench.py times `run` only) or overlap copies with compute. upload/download block. */
int sayuri_hip_upload(sayuri_hip_ctx* ctx, int n, const float* planes, const int* board_sizes);
int sayuri_hip_run(sayuri_hip_ctx* ctx);   /* asynchronous on the ctx stream */
int sayuri_hip_sync(sayuri_hip_ctx* ctx);
int sayuri_hip_download(sayuri_hip_ctx* ctx, float* prob, float* pass, float* misc, float* own);

/* Pipelined form for the pump thread: enqueue H2D + the whole graph + D2H on the ctx stream and
 * return at once; `ticket` (0 or 1, two batches may be in flight) is waited for / polled later.
 * All host buffers must be page-locked and stay valid until the wait.  `pass` and `misc` are written by the heads kernel
 * itself when the device can address them -- memory from sayuri_hip_host_alloc (or any other page-locked memory that is
 * MAPPED into the device's address space); a buffer that is page-locked but not mapped is detected once per pointer
 * (hipHostGetDevicePointer) and served through copies.  Packed records (sayuri_hip_submit_packed) in such memory are likewise
 * read by the first kernel where they lie, with no copy: they must stay UNCHANGED until the wait, not merely valid. */
int sayuri_hip_submit(sayuri_hip_ctx* ctx, int n, const float* planes, const int* board_sizes, float* prob,
                      float* pass, float* misc, float* own, int* ticket);
/* The same two entry points for PACKED planes (sayuri_amd/csrc/host/packed_planes.h; SURVEY.md section 8 row f1, the
 * encoder on the critical path -- reference src/neural/encoder.cc:101-368 fills 43 fp32 planes per evaluation).
 *   records  [n][binary_planes*12 + 8] 32-bit words: bits[binary_planes][12] (bit y*bs+x of a 0/1 plane, in the
 *            SAMPLE's own cell order -- no re-padding into the NN grid), then 8 floats (the value of each broadcast
 *            plane: rule, wave, komi/20, -komi/20, N/361, 1 for 43-plane nets)
 *   binary_planes = input_channels - 6 (37) for v3+ nets, 34 for the 38-plane v1/v2 encoder
 * 1.8 KB per sample instead of 62 KB; the first kernel expands the bits into the fp16 activations, the network sees the
 * same values as through sayuri_hip_forward and returns bit-identical outputs. */
int sayuri_hip_forward_packed(sayuri_hip_ctx* ctx, int n, const unsigned* records, int binary_planes, const int* board_sizes,
                              float* prob, float* pass, float* misc, float* own);
int sayuri_hip_submit_packed(sayuri_hip_ctx* ctx, int n, const unsigned* records, int binary_planes, const int* board_sizes,
                             float* prob, float* pass, float* misc, float* own, int* ticket);
int sayuri_hip_wait(sayuri_hip_ctx* ctx, int ticket);
int sayuri_hip_query(sayuri_hip_ctx* ctx, int ticket); /* 1 = finished, 0 = still running, -1 = error */

/* Time `iters` back-to-back runs of the uploaded batch with HIP events on the ctx stream
 * (torch.cuda.Event cannot see this stream). total_ms = wall of all iters on the device. */
int sayuri_hip_time_runs(sayuri_hip_ctx* ctx, int iters, float* total_ms);

/* Per-kernel-class device time of ONE run of the uploaded batch (events around every launch).
 * Fills up to `cap` rows; returns the row count, -1 on error. */
typedef struct {
    char name[48];
    int32_t launches;
    float total_ms;
    double flops;  /* algorithmic FLOPs (2*MAC of the direct convolution) of those launches */
    double bytes;  /* algorithmic HBM bytes (inputs + outputs + weights once) of those launches */
} sayuri_hip_kernel_stat;
int sayuri_hip_profile_run(sayuri_hip_ctx* ctx, sayuri_hip_kernel_stat* rows, int cap);

/* Per-launch timing INSIDE the timed region: after sayuri_hip_mark_kernel(ctx, "conv3x3_tower"),
 * every launch of that kernel class during sayuri_hip_time_runs is bracketed by an
 * un-synchronised HIP event pair on the ctx stream; sayuri_hip_timed_stat returns their sum
 * (launch count, total ms, algorithmic FLOPs/bytes).  NULL / "" disables marking. */
int sayuri_hip_mark_kernel(sayuri_hip_ctx* ctx, const char* name);
int sayuri_hip_timed_stat(sayuri_hip_ctx* ctx, sayuri_hip_kernel_stat* row);

/* Page-locked host memory for the staging buffers handed to upload/download/forward
 * (replaces the cudaHostAlloc staging of reference cuda_common.cc:312-403); the host side
 * never includes a HIP header. */
void* sayuri_hip_host_alloc(size_t bytes);
void sayuri_hip_host_free(void* p);

/* Bytes of device memory held by the ctx. */
size_t sayuri_hip_device_bytes(const sayuri_hip_ctx* ctx);

/* How many chains the last forward of the ctx was run as (measurement / tests).  A batch whose layers are more than one
 * round of workgroups on a network without the persistent launch (configs[4]: 40b x 384) is cut into groups of board tiles,
 * each a chain of per-layer launches on a stream of its own -- the reference has one stream per GPU and one launch per
 * kernel (cuda_forward_pipe.cc:713-981).  SAYURI_CHAINS=1 turns it off, =N forces N. */
int sayuri_hip_last_chains(const sayuri_hip_ctx* ctx);

/* 1 when the ctx runs its residual tower as ONE persistent launch (the code object of csrc/hip/conv_tower.h is in the library
 * and loaded), 0 when it falls back to one launch per layer -- a build whose assembly post-processor (tower_seam.py, validated
 * on ROCm 7.2's hipcc) rejected the compiler's output, SAYURI_TOWER=0, or the fp32 engine.  bench.py prints it on its line.
 * The reference always launches per layer (cuda_forward_pipe.cc:713-981). */
int sayuri_hip_tower_state(const sayuri_hip_ctx* ctx);

/* Release everything (replaces NNGraph::DestroyGraph, cuda_forward_pipe.cc:1092-1130). */
void sayuri_hip_destroy(sayuri_hip_ctx* ctx);

const char* sayuri_hip_last_error(void);

/* ---- layer-level taps for the parity tests (not used by the pipe) ---------------------- */
/* One convolution layer on host NCHW fp32 tensors, through the same device kernels:
 * x [n][cin][bs*bs] with per-sample board sizes (compact, stride bs*bs per channel),
 * w [cout][cin][k][k], bias [cout] or NULL, res like the output or NULL,
 * y [n][cout][bs*bs].  k = 1 or 3 (MFMA implicit GEMM) or depthwise (cin == 1 in w). */
int sayuri_hip_test_conv(int device, int use_fp16, int n, const int* board_sizes, int max_board,
                         int cin, int cout, int k, int depthwise, int act, int post_residual,
                         const float* x, const float* w, const float* bias, const float* res,
                         float* y);
/* Kernel family the calling thread's last sayuri_hip_test_conv ran: 0 generic implicit GEMM (conv_mfma.h),
 * 1 LDS-DMA tiles across samples (conv_glds.h), 2 one workgroup per board (conv_board.h), 3 depthwise. */
int sayuri_hip_test_last_conv_kind(void);
/* One squeeze-and-excitation unit through the se_pool / se_fc / se_scale kernels (reference SEUnit::Forward,
 * src/neural/blas/se_unit.cc:70-128): x, res (or NULL), y are [n][channels][bs*bs] like sayuri_hip_test_conv's tensors,
 * w1 [se_size][3*channels], b1 [se_size], w2 [2*channels][se_size], b2 [2*channels];
 * gate (or NULL) receives [n][2*channels] = sigmoid(gamma) | beta, i.e. GlobalPooling<false> + both FullyConnects. */
int sayuri_hip_test_se_unit(int device, int use_fp16, int n, const int* board_sizes, int max_board, int channels,
                            int se_size, int act, const float* x, const float* res, const float* w1, const float* b1,
                            const float* w2, const float* b2, float* y, float* gate);
/* Everything after the two head convolutions in one launch (head_tail_kernel; reference blas_forward_pipe.cc:496-580):
 * pconv [n][policy_channels][bs*bs], vconv [n][value_channels][bs*bs] (activated head convolutions),
 * weights12 = { p_inter w [Cp][3Cp], b; pass_fc w [pass_outs][Cp], b; v_inter w [3Cv][3Cv], b; v_misc w [misc_outs][3Cv], b;
 * prob_conv w [prob_channels][Cp], b; ownership w [Cv], b[1] };  outputs in the NN grid as sayuri_hip_forward's. */
int sayuri_hip_test_head_tail(int device, int use_fp16, int n, const int* board_sizes, int max_board, int policy_channels,
                              int value_channels, int prob_channels, int pass_outs, int misc_outs, int act,
                              const float* pconv, const float* vconv, const float* const* weights12, float* prob,
                              float* pass, float* misc, float* own);

/* The two fused kernels of the fp16 engine at kernel level (round 3):
 * sayuri_hip_test_conv_se   a channels -> channels 3x3 convolution (w [C][C][3][3], bias [C]) with the SE unit that follows
 *   it INSIDE the kernel (conv_board_se_kernel; via_tower != 0: the same stage inside a one-layer run of the persistent
 *   tower kernel): y = act(sigmoid(gamma) * conv(x) + beta + res), reference SEUnit::Forward on the convolution's output
 *   (se_unit.cc:70-128).  Tensors as in sayuri_hip_test_conv / sayuri_hip_test_se_unit.
 * sayuri_hip_test_head_board   both heads of a sample in one workgroup (head_board_kernel): trunk [n][channels][bs*bs],
 *   p_w [Cp][channels], p_b [Cp], v_w [Cv][channels], v_b [Cv] (the two 1x1 head convolutions, reference
 *   blas_forward_pipe.cc:449-495), weights12 and outputs as in sayuri_hip_test_head_tail.
 * Both return 1 (not an error) when the fused kernel does not apply to the arguments: the engine then runs the separate
 * kernels (several samples per tile / channel counts without a kernel variant). */
int sayuri_hip_test_conv_se(int device, int n, const int* board_sizes, int max_board, int channels, int se_size, int act,
                            int via_tower, const float* x, const float* w, const float* bias, const float* res, const float* w1,
                            const float* b1, const float* w2, const float* b2, float* y);
int sayuri_hip_test_head_board(int device, int n, const int* board_sizes, int max_board, int channels, int policy_channels,
                               int value_channels, int prob_channels, int pass_outs, int misc_outs, int act, const float* trunk,
                               const float* p_w, const float* p_b, const float* v_w, const float* v_b,
                               const float* const* weights12, float* prob, float* pass, float* misc, float* own);

#ifdef __cplusplus
}
#endif
#endif
