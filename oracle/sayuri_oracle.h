/* oracle/sayuri_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the reference's NN hot path: the weight-file loader and the
 * batch-1 CPU forward pipe.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library; the product path (sayuri_amd/csrc) never links or calls it.
 *
 * Pinned against the reference itself: tests/golden/ holds outputs of the reference's own
 * BlasForwardPipe / DNNLoader (compiled unmodified into oracle/_ref by oracle/Makefile and
 * driven by tests/golden/make_golden.py), and tests/test_oracle.py checks this restatement
 * against them.  The reference ships no tests or golden vectors of its own.
 */
#ifndef SAYURI_ORACLE_H
#define SAYURI_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct so_net so_net;

/* reference src/neural/loader.cc:26-121 (FromFile/Parse); winograd mirrors option "winograd" */
so_net* so_load(const char* path, int winograd, char* err, int errlen);
void so_free(so_net* net);

/* info[12]: version, input_channels, residual_blocks, residual_channels, policy_head_channels,
 * value_head_channels, probabilities_channels, pass_probability_outputs, ownership_channels,
 * value_misc_outputs, default_act, policy_head_type (0 normal, 1 RepLK) */
int so_info(const so_net* net, int* info);
/* binfo[5]: type (1 residual, 2 bottleneck, 3 nested bottleneck, 4 mixer), apply_se, se_size,
 * bottleneck_channels, feedforward_channels */
int so_block_info(const so_net* net, int idx, int* binfo);
/* post-fold tensors by name, e.g. "input_conv.w", "tower.3.conv1.u", "v_misc.b" */
long so_get_tensor(const so_net* net, const char* name, float* dst, long cap);

/* reference src/neural/blas/blas_forward_pipe.cc:314-619.  planes = [C_in][bs*bs].
 * out = prob[bs*bs], own[bs*bs], pass, wdl[3], stm, score, q_err, score_err, (float)offset */
int so_forward(const so_net* net, int board_size, float komi, int offset, const float* planes,
               float* out);
/* the same evaluation before FillOutputs selects a policy plane:
 * prob [prob_ch][bs*bs], pass [pass_outs], misc [misc_outs], own [own_ch][bs*bs] */
/* generator aid: max |x| of the residual stream after every block (slot 0 = after the input convolution) of the next
 * so_forward / so_forward_raw calls on this thread; NULL switches it off */
void so_set_trunk_probe(float* absmax, int cap);
int so_forward_raw(const so_net* net, int board_size, const float* planes, float* prob,
                   float* pass, float* misc, float* own);

/* reference src/neural/network.cc:361-429 (TransformResult + ActivatePolicy), symmetry = identity.
 * in: the `out` vector of so_forward; post: prob softmax over bs*bs+1 (pass last) at `temp`,
 * own tanh, then wdl softmax[3], wdl_winrate, stm_winrate, final_score, q_error, score_error */
int so_postprocess(int board_size, float temp, const float* raw, float* post);

#ifdef __cplusplus
}
#endif
#endif
