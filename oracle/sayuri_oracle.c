/* oracle/sayuri_oracle.c -- TEST INFRASTRUCTURE ONLY (see sayuri_oracle.h).
 *
 * Plain-C restatement of the reference's CPU path for one NN evaluation.  Every
 * function cites the reference file:line it follows (paths relative to
 * /root/reference/).  It is deliberately simple and scalar: it is the checker, not
 * the product, and nothing under sayuri_amd/ may call it.
 *
 * Pinning: tests/test_oracle.py compares this file against golden vectors produced by
 * the *reference's own* BlasForwardPipe and DNNLoader (oracle/_ref, built from the
 * unmodified reference sources by oracle/Makefile; generator tests/golden/make_golden.py).
 */
#include "sayuri_oracle.h"

#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>

/* ------------------------------------------------------------------ model description */

enum { ACT_IDENTITY = 0, ACT_RELU, ACT_ELU, ACT_SELU, ACT_GELU, ACT_MISH, ACT_SWISH, ACT_HARDSWISH };
enum { BLK_RESIDUAL = 1, BLK_BOTTLENECK = 2, BLK_NESTED = 3, BLK_MIXER = 4 };

typedef struct {
    int cin, cout, k;
    float *w, *b, *u; /* w [cout][cin][k][k], b [cout], u [36][cin][cout] for k == 3 */
    long nw;
} so_conv;

typedef struct {
    int cin, cout;
    float *w, *b; /* w [cout][cin] */
} so_fc;

enum { CV_1 = 0, CV_2, CV_3, CV_4, CV_PRE, CV_POST, CV_DW, CV_COUNT };

typedef struct {
    int type, se, se_size, inner, ffn;
    so_conv conv[CV_COUNT];
    so_fc squeeze, excite;
} so_block;

struct so_net {
    int version, in_ch, nblocks, channels, pol_ch, val_ch, prob_ch, pass_outs, own_ch, misc_outs;
    int act, pol_type, winograd;
    so_conv input, p_hd, p_dw, p_pt, prob, v_hd, v_own;
    so_fc p_inter, pass_fc, v_inter, v_misc;
    so_block* tower;
};

/* ------------------------------------------------------------------ activations
 * reference src/neural/activation.h:41-81 */
static float activate(float x, int act) {
    switch (act) {
    case ACT_RELU: return x > 0.f ? x : 0.f;
    case ACT_ELU: return x > 0.f ? x : (expf(x) - 1);
    case ACT_SELU: return x > 0.f ? (1.05070098f * x) : (1.05070098f * 1.67326324f * (expf(x) - 1.0f));
    case ACT_GELU: return 0.5f * x * (1.0f + tanhf(0.7978845608028654f * (x + 0.044715 * x * x * x)));
    case ACT_MISH: return x * tanhf(logf(1.0f + expf(x)));
    case ACT_SWISH: return x / (1.0f + expf(-x));
    case ACT_HARDSWISH: return x >= 3.f ? x : x <= -3.f ? 0.f : (x * (x + 3.0f) / 6.0f);
    default: return x;
    }
}

static int act_from_string(const char* s) { /* activation.h:19-39 */
    char buf[32];
    size_t i;
    for (i = 0; s[i] && i + 1 < sizeof(buf); ++i) buf[i] = (s[i] >= 'A' && s[i] <= 'Z') ? s[i] + 32 : s[i];
    buf[i] = 0;
    if (!strcmp(buf, "identity")) return ACT_IDENTITY;
    if (!strcmp(buf, "relu")) return ACT_RELU;
    if (!strcmp(buf, "elu")) return ACT_ELU;
    if (!strcmp(buf, "selu")) return ACT_SELU;
    if (!strcmp(buf, "gelu")) return ACT_GELU;
    if (!strcmp(buf, "mish")) return ACT_MISH;
    if (!strcmp(buf, "swish")) return ACT_SWISH;
    if (!strcmp(buf, "hardswish")) return ACT_HARDSWISH;
    return -1;
}

/* ------------------------------------------------------------------ Winograd F(4x4,3x3)
 * reference src/neural/winograd_helper.{h,cc}.  kSqrt2 there is a `double` initialised
 * from a float literal (winograd_helper.h:10), so its value is (double)(float)sqrt(2). */
#define WINO_M 4
#define WINO_ALPHA 6
#define WINO_TILE 36
static const double kSqrt2 = (double)1.4142135623730951f;

static int wino_wtiles(int bs) { return bs / WINO_M + (bs % WINO_M != 0); } /* winograd_helper.cc:3-5 */
static int wino_p(int bs) { int t = wino_wtiles(bs); return t * t; }       /* :7-10 */

/* winograd_helper.cc:13-83: U = G f G^T stored [36][cin][cout] */
static float* wino_transform_f(const float* f, int outputs, int channels) {
    const float G[18] = {1.0f, 0.0f, 0.0f,
                         (float)(-2.0f / 3.0f), (float)(-kSqrt2 / 3.0f), (float)(-1.0f / 3.0f),
                         (float)(-2.0f / 3.0f), (float)(kSqrt2 / 3.0f), (float)(-1.0f / 3.0f),
                         (float)(1.0f / 6.0f), (float)(kSqrt2 / 6.0f), (float)(1.0f / 3.0f),
                         (float)(1.0f / 6.0f), (float)(-kSqrt2 / 6.0f), (float)(1.0f / 3.0f),
                         0.0f, 0.0f, 1.0f};
    float* U = (float*)malloc(sizeof(float) * WINO_TILE * outputs * channels);
    float temp[18];
    for (int c = 0; c < channels; ++c) {
        for (int o = 0; o < outputs; ++o) {
            for (int i = 0; i < WINO_ALPHA; ++i)
                for (int j = 0; j < 3; ++j) {
                    float acc = 0.0f;
                    for (int k = 0; k < 3; ++k) acc += G[i * 3 + k] * f[o * channels * 9 + c * 9 + k * 3 + j];
                    temp[i * 3 + j] = acc;
                }
            for (int xi = 0; xi < WINO_ALPHA; ++xi)
                for (int nu = 0; nu < WINO_ALPHA; ++nu) {
                    float acc = 0.0f;
                    for (int k = 0; k < 3; ++k) acc += temp[xi * 3 + k] * G[nu * 3 + k];
                    U[(xi * WINO_ALPHA + nu) * outputs * channels + c * outputs + o] = acc;
                }
        }
    }
    return U;
}

/* winograd_convolution3.cc:44-70: one column of B^T d */
static void mul_bt(float* o, const float i0, const float i1, const float i2, const float i3,
                   const float i4, const float i5) {
    double i3m1 = i1 * -kSqrt2 + i3 * (kSqrt2 / 2.0f);
    float i4m2 = i2 * -2.0f + i4 * 1.0f;
    o[0] = i0 + i2 * (-5.0f / 2.0f) + i4;
    o[1] = (float)(i3m1 + i4m2);
    o[2] = (float)(-i3m1 + i4m2);
    double i3m1_2 = i3 * (kSqrt2) + i1 * (-kSqrt2 / 2.0f);
    float i4m2_2 = i2 * (-1.0f / 2.0f) + i4;
    o[3] = (float)(i3m1_2 + i4m2_2);
    o[4] = (float)(-i3m1_2 + i4m2_2);
    o[5] = i1 + i3 * (-5.0f / 2.0f) + i5;
}

/* winograd_convolution3.cc:12-152: V[36][C][P] */
static void wino_transform_in(int bs, const float* in, float* V, int C) {
    const int WT = wino_wtiles(bs), P = WT * WT, Wpad = 2 + WINO_M * WT;
    float* pad = (float*)malloc(sizeof(float) * Wpad * Wpad);
    for (int ch = 0; ch < C; ++ch) {
        memset(pad, 0, sizeof(float) * Wpad * Wpad);
        for (int y = 0; y < bs; ++y)
            for (int x = 0; x < bs; ++x) pad[(y + 1) * Wpad + x + 1] = in[ch * bs * bs + y * bs + x];
        for (int by = 0; by < WT; ++by)
            for (int bx = 0; bx < WT; ++bx) {
                const int yin = WINO_M * by, xin = WINO_M * bx;
                float T1[6][6], col[6], r[6];
                for (int xx = 0; xx < 6; ++xx) { /* T1[:, xx] = B^T * d[:, xx] */
                    mul_bt(col, pad[(yin + 0) * Wpad + xin + xx], pad[(yin + 1) * Wpad + xin + xx],
                           pad[(yin + 2) * Wpad + xin + xx], pad[(yin + 3) * Wpad + xin + xx],
                           pad[(yin + 4) * Wpad + xin + xx], pad[(yin + 5) * Wpad + xin + xx]);
                    for (int i = 0; i < 6; ++i) T1[i][xx] = col[i];
                }
                for (int xx = 0; xx < 6; ++xx) { /* row xx of (B^T d) times B */
                    mul_bt(r, T1[xx][0], T1[xx][1], T1[xx][2], T1[xx][3], T1[xx][4], T1[xx][5]);
                    for (int j = 0; j < 6; ++j) V[(xx * 6 + j) * C * P + ch * P + by * WT + bx] = r[j];
                }
            }
    }
    free(pad);
}

/* sgemm.cc:44-62 sgemm_tn as used by winograd_convolution3.cc:154-185:
 * M[b][k][p] = sum_c U[b][c][k] * V[b][c][p] */
static void wino_sgemm(int bs, const float* U, const float* V, float* M, int C, int K) {
    const int P = wino_p(bs);
    for (int b = 0; b < WINO_TILE; ++b) {
        const float* A = U + (long)b * K * C;
        const float* B = V + (long)b * C * P;
        float* Cm = M + (long)b * K * P;
        for (long i = 0; i < (long)K * P; ++i) Cm[i] = 0.f;
        for (int i = 0; i < K; ++i)
            for (int k = 0; k < C; ++k) {
                const float a = A[k * K + i];
                for (int j = 0; j < P; ++j) Cm[i * P + j] += a * B[k * P + j];
            }
    }
}

/* winograd_convolution3.cc:203-221 */
static void mul_at(float* o, const float i0, const float i1, const float i2, const float i3,
                   const float i4, const float i5) {
    float t1p2 = (i1 + i2) * (1.0f / 2.0f);
    double t1m2 = (i1 - i2) * (kSqrt2 / 4.0f);
    float t3p4 = i3 + i4;
    double t3m4 = (i3 - i4) * (kSqrt2);
    o[0] = i0 + t1p2 + t1p2 + t3p4;
    o[1] = (float)(t1m2 + t1m2 + t3m4);
    o[2] = t1p2 + t3p4 + t3p4;
    o[3] = (float)(t1m2 + t3m4 + t3m4 + i5);
}

/* winograd_convolution3.cc:187-278 */
static void wino_transform_out(int bs, const float* M, float* Y, int K) {
    const int WT = wino_wtiles(bs), P = WT * WT;
    for (int k = 0; k < K; ++k)
        for (int bx = 0; bx < WT; ++bx)
            for (int by = 0; by < WT; ++by) {
                const int x = WINO_M * bx, y = WINO_M * by, b = by * WT + bx;
                float m[6][6], t[4][6], o[4][4], c4[4];
                for (int xi = 0; xi < 6; ++xi)
                    for (int nu = 0; nu < 6; ++nu) m[xi][nu] = M[(long)(xi * 6 + nu) * K * P + k * P + b];
                for (int j = 0; j < 6; ++j) {
                    mul_at(c4, m[0][j], m[1][j], m[2][j], m[3][j], m[4][j], m[5][j]);
                    for (int i = 0; i < 4; ++i) t[i][j] = c4[i];
                }
                for (int i = 0; i < 4; ++i) mul_at(o[i], t[i][0], t[i][1], t[i][2], t[i][3], t[i][4], t[i][5]);
                for (int i = 0; i < 4; ++i)
                    for (int j = 0; j < 4; ++j)
                        if (y + i < bs && x + j < bs) Y[k * bs * bs + (y + i) * bs + x + j] = o[i][j];
            }
}

/* ------------------------------------------------------------------ direct convolutions */

/* convolution.h:41-125 (Im2col + sgemm_nn): out[K][S] = W[K][C*9] * col[C*9][S] */
static void conv3_im2col(int bs, int C, int K, const float* in, const float* w, float* out) {
    const int S = bs * bs;
    float* col = (float*)calloc((size_t)C * 9 * S, sizeof(float));
    for (int c = 0; c < C; ++c)
        for (int kr = 0; kr < 3; ++kr)
            for (int kc = 0; kc < 3; ++kc) {
                float* dst = col + ((long)(c * 9 + kr * 3 + kc)) * S;
                for (int y = 0; y < bs; ++y) {
                    const int iy = y + kr - 1;
                    if (iy < 0 || iy >= bs) continue;
                    for (int x = 0; x < bs; ++x) {
                        const int ix = x + kc - 1;
                        if (ix >= 0 && ix < bs) dst[y * bs + x] = in[c * S + iy * bs + ix];
                    }
                }
            }
    const int KD = C * 9;
    for (long i = 0; i < (long)K * S; ++i) out[i] = 0.f;
    for (int i = 0; i < K; ++i)
        for (int k = 0; k < KD; ++k) { /* sgemm.cc:3-21 sgemm_nn */
            const float a = w[(long)i * KD + k];
            for (int j = 0; j < S; ++j) out[(long)i * S + j] += a * col[(long)k * S + j];
        }
    free(col);
}

/* convolution.cc:3-25 (Convolution1 -> sgemm_nn) */
static void conv1(int bs, int C, int K, const float* in, const float* w, float* out) {
    const int S = bs * bs;
    for (long i = 0; i < (long)K * S; ++i) out[i] = 0.f;
    for (int i = 0; i < K; ++i)
        for (int k = 0; k < C; ++k) {
            const float a = w[(long)i * C + k];
            for (int j = 0; j < S; ++j) out[(long)i * S + j] += a * in[(long)k * S + j];
        }
}

/* convolution.cc:27-62 */
static void conv_depthwise(int bs, int fs, int C, const float* in, const float* w, float* out) {
    const int S = bs * bs, pad = fs / 2;
    for (int c = 0; c < C; ++c)
        for (int row = 0; row < bs; ++row)
            for (int col = 0; col < bs; ++col) {
                float val = 0.0f;
                for (int kr = 0; kr < fs; ++kr)
                    for (int kc = 0; kc < fs; ++kc) {
                        const int ir = -pad + kr + row, ic = -pad + kc + col;
                        if (ir >= 0 && ir < bs && ic >= 0 && ic < bs)
                            val += in[c * S + ir * bs + ic] * w[c * fs * fs + kr * fs + kc];
                    }
                out[c * S + row * bs + col] = val;
            }
}

/* blas_forward_pipe.cc:18-44 (Convolution3Forward) */
static void conv3(const so_net* n, int bs, const so_conv* cv, const float* in, float* out) {
    if (n->winograd) {
        const int P = wino_p(bs);
        float* V = (float*)malloc(sizeof(float) * WINO_TILE * cv->cin * P);
        float* M = (float*)malloc(sizeof(float) * WINO_TILE * cv->cout * P);
        wino_transform_in(bs, in, V, cv->cin);
        wino_sgemm(bs, cv->u, V, M, cv->cin, cv->cout);
        wino_transform_out(bs, M, out, cv->cout);
        free(V);
        free(M);
    } else {
        conv3_im2col(bs, cv->cin, cv->cout, in, cv->w, out);
    }
}

/* ------------------------------------------------------------------ elementwise / pooling / fc */

/* biases.cc:14-47 AddSpatialBiases: act(x + b[c] + res) */
static void add_spatial(int bs, int C, float* x, const float* bias, const float* res, int act) {
    const int S = bs * bs;
    for (int c = 0; c < C; ++c) {
        const float b = bias ? bias[c] : 0.0f;
        for (int i = 0; i < S; ++i) {
            float v = x[c * S + i] + b;
            if (res) v += res[c * S + i];
            x[c * S + i] = activate(v, act);
        }
    }
}

/* biases.cc:49-77 AddSpatialBiasesPost: act(x + b[c]) + res */
static void add_spatial_post(int bs, int C, float* x, const float* bias, int act, const float* res) {
    const int S = bs * bs;
    for (int c = 0; c < C; ++c) {
        const float b = bias ? bias[c] : 0.0f;
        for (int i = 0; i < S; ++i) {
            float v = activate(x[c * S + i] + b, act);
            if (res) v += res[c * S + i];
            x[c * S + i] = v;
        }
    }
}

/* se_unit.cc:9-68 GlobalPooling<false/true>; se_unit.h:19-22 constants */
static void global_pool(int bs, int C, const float* x, float* out, int value_head) {
    const int S = bs * bs;
    const float b_diff = (float)bs - 14.0f;
    const float c0 = b_diff / 10.f, c1 = b_diff * b_diff / 100.f - 0.1f;
    for (int c = 0; c < C; ++c) {
        float sum = 0.0f, mx = -5000.0f;
        for (int i = 0; i < S; ++i) {
            const float v = x[c * S + i];
            sum += v;
            if (v > mx) mx = v;
        }
        const float mean = sum / (float)S;
        out[c] = mean;
        out[c + C] = mean * c0;
        out[c + 2 * C] = value_head ? mean * c1 : mx;
    }
}

/* fullyconnect.cc:7-19 + sgemm.cc:23-42 (sgemm_nt) + biases.cc:79-89 */
static void fully_connect(const so_fc* fc, const float* in, float* out, int act) {
    for (int o = 0; o < fc->cout; ++o) {
        float sum = 0;
        for (int k = 0; k < fc->cin; ++k) sum += in[k] * fc->w[(long)o * fc->cin + k];
        out[o] = activate(fc->b[o] + sum, act);
    }
}

/* se_unit.cc:70-128 */
static void se_unit(int bs, int C, const so_block* b, float* x, const float* res, int act) {
    const int S = bs * bs;
    float* pool = (float*)malloc(sizeof(float) * 3 * C);
    float* mid = (float*)malloc(sizeof(float) * b->se_size);
    global_pool(bs, C, x, pool, 0);
    fully_connect(&b->squeeze, pool, mid, act);
    fully_connect(&b->excite, mid, pool, ACT_IDENTITY);
    for (int c = 0; c < C; ++c) {
        const float gamma = 1.0f / (1.0f + expf(-pool[c]));
        const float beta = pool[C + c];
        for (int i = 0; i < S; ++i) {
            float v = gamma * x[c * S + i] + beta;
            if (res) v += res[c * S + i];
            x[c * S + i] = activate(v, act);
        }
    }
    free(pool);
    free(mid);
}

/* ---- layer-level taps for the kernel tests (tests/test_gpu_smallops.py): the static restatements above, callable.
 * They are the same code the whole-network goldens pin (tests/test_oracle.py). */
void so_tap_global_pool(int bs, int C, const float* x, float* out, int value_head) { global_pool(bs, C, x, out, value_head); }
void so_tap_fully_connect(int cin, int cout, const float* w, const float* b, const float* in, float* out, int act) {
    so_fc fc;
    memset(&fc, 0, sizeof(fc));
    fc.cin = cin; fc.cout = cout; fc.w = (float*)w; fc.b = (float*)b;
    fully_connect(&fc, in, out, act);
}
/* SEUnit::Forward on x [C][bs*bs] in place (se_unit.cc:70-128); w1 [se][3C], w2 [2C][se] */
void so_tap_se_unit(int bs, int C, int se, const float* w1, const float* b1, const float* w2, const float* b2, float* x,
                    const float* res, int act) {
    so_block b;
    memset(&b, 0, sizeof(b));
    b.se = 1; b.se_size = se;
    b.squeeze.cin = 3 * C; b.squeeze.cout = se; b.squeeze.w = (float*)w1; b.squeeze.b = (float*)b1;
    b.excite.cin = se; b.excite.cout = 2 * C; b.excite.w = (float*)w2; b.excite.b = (float*)b2;
    se_unit(bs, C, &b, x, res, act);
}
/* everything after the two head convolutions (blas_forward_pipe.cc:496-580): pconv [PC][S] and vconv [VC][S] are the
 * activated head convolutions; outputs prob [prob_ch][S], pass [pass_outs], own [S], misc [misc_outs] */
void so_tap_head_tail(int bs, int PC, int VC, int prob_ch, int pass_outs, int misc_outs, int act, float* pconv, const float* vconv,
                      const float* p_inter_w, const float* p_inter_b, const float* pass_w, const float* pass_b,
                      const float* v_inter_w, const float* v_inter_b, const float* v_misc_w, const float* v_misc_b,
                      const float* prob_w, const float* prob_b, const float* own_w, const float* own_b,
                      float* prob, float* pass, float* own, float* misc) {
    const int maxi = PC > VC ? PC : VC;
    float* pool = (float*)malloc(sizeof(float) * 3 * maxi);
    float* inter = (float*)malloc(sizeof(float) * 3 * maxi);
    so_fc fc;
    memset(&fc, 0, sizeof(fc));
    global_pool(bs, PC, pconv, pool, 0);
    fc.cin = 3 * PC; fc.cout = PC; fc.w = (float*)p_inter_w; fc.b = (float*)p_inter_b;
    fully_connect(&fc, pool, inter, act);
    add_spatial(bs, PC, pconv, inter, NULL, ACT_IDENTITY);
    conv1(bs, PC, prob_ch, pconv, prob_w, prob);
    add_spatial(bs, prob_ch, prob, prob_b, NULL, ACT_IDENTITY);
    fc.cin = PC; fc.cout = pass_outs; fc.w = (float*)pass_w; fc.b = (float*)pass_b;
    fully_connect(&fc, inter, pass, ACT_IDENTITY);
    global_pool(bs, VC, vconv, pool, 1);
    fc.cin = 3 * VC; fc.cout = 3 * VC; fc.w = (float*)v_inter_w; fc.b = (float*)v_inter_b;
    fully_connect(&fc, pool, inter, act);
    conv1(bs, VC, 1, vconv, own_w, own);
    add_spatial(bs, 1, own, own_b, NULL, ACT_IDENTITY);
    fc.cin = 3 * VC; fc.cout = misc_outs; fc.w = (float*)v_misc_w; fc.b = (float*)v_misc_b;
    fully_connect(&fc, inter, misc, ACT_IDENTITY);
    free(pool);
    free(inter);
}

/* ------------------------------------------------------------------ forward
 * blas_forward_pipe.cc:46-563.  Buffers are named by role instead of the reference's
 * swap dance: `x` is the block input (the skip), `y` the block output. */
static int max3(int a, int b, int c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); }

/* Probe for the fixture generator (tests/golden/make_golden.py): max |x| of the residual stream after every block of the
 * NEXT so_forward_raw calls on this thread (entry 0 = after the input convolution); NULL switches it off. */
static __thread float* g_trunk_probe = NULL;
static __thread int g_trunk_probe_cap = 0;
void so_set_trunk_probe(float* absmax, int cap) { g_trunk_probe = absmax; g_trunk_probe_cap = cap; }
static void trunk_probe(int slot, const float* y, int count) {
    if (!g_trunk_probe || slot >= g_trunk_probe_cap) return;
    float m = 0.f;
    for (int i = 0; i < count; ++i) { const float a = fabsf(y[i]); if (a > m) m = a; }
    if (m > g_trunk_probe[slot]) g_trunk_probe[slot] = m;
}

int so_forward_raw(const so_net* n, int bs, const float* planes, float* prob, float* pass,
                   float* misc, float* own) {
    if (!n || bs < 2 || bs > 25) return -1;
    const int S = bs * bs, C = n->channels, act = n->act;
    int peak = C;
    for (int i = 0; i < n->nblocks; ++i) peak = max3(peak, n->tower[i].inner, n->tower[i].ffn);
    float* x = (float*)malloc(sizeof(float) * peak * S);
    float* y = (float*)malloc(sizeof(float) * peak * S);
    float* t0 = (float*)malloc(sizeof(float) * peak * S);
    float* t1 = (float*)malloc(sizeof(float) * peak * S);

    conv3(n, bs, &n->input, planes, y); /* :391-401 */
    add_spatial(bs, C, y, n->input.b, NULL, act);
    trunk_probe(0, y, C * S);

    for (int i = 0; i < n->nblocks; ++i) {
        const so_block* b = &n->tower[i];
        float* tmp = x; x = y; y = tmp; /* x = block input */
        const float* last_skip = b->se ? NULL : x;
        const int last_act = b->se ? ACT_IDENTITY : act;
        if (b->type == BLK_RESIDUAL) { /* :46-87 */
            conv3(n, bs, &b->conv[CV_1], x, t0);
            add_spatial(bs, C, t0, b->conv[CV_1].b, NULL, act);
            conv3(n, bs, &b->conv[CV_2], t0, y);
            add_spatial(bs, C, y, b->conv[CV_2].b, last_skip, last_act);
        } else if (b->type == BLK_BOTTLENECK) { /* :89-161 */
            const int I = b->inner;
            conv1(bs, C, I, x, b->conv[CV_PRE].w, t0);
            add_spatial(bs, I, t0, b->conv[CV_PRE].b, NULL, act);
            conv3(n, bs, &b->conv[CV_1], t0, t1);
            add_spatial(bs, I, t1, b->conv[CV_1].b, NULL, act);
            conv3(n, bs, &b->conv[CV_2], t1, t0);
            add_spatial(bs, I, t0, b->conv[CV_2].b, NULL, act);
            conv1(bs, I, C, t0, b->conv[CV_POST].w, y);
            add_spatial(bs, C, y, b->conv[CV_POST].b, last_skip, last_act);
        } else if (b->type == BLK_NESTED) { /* :163-264 */
            const int I = b->inner;
            float* r1 = (float*)malloc(sizeof(float) * I * S);
            conv1(bs, C, I, x, b->conv[CV_PRE].w, r1);
            add_spatial(bs, I, r1, b->conv[CV_PRE].b, NULL, act);
            conv3(n, bs, &b->conv[CV_1], r1, t0);
            add_spatial(bs, I, t0, b->conv[CV_1].b, NULL, act);
            conv3(n, bs, &b->conv[CV_2], t0, t1);
            add_spatial(bs, I, t1, b->conv[CV_2].b, r1, act); /* t1 = inner residual #2 */
            conv3(n, bs, &b->conv[CV_3], t1, t0);
            add_spatial(bs, I, t0, b->conv[CV_3].b, NULL, act);
            conv3(n, bs, &b->conv[CV_4], t0, r1);
            add_spatial(bs, I, r1, b->conv[CV_4].b, t1, act);
            conv1(bs, I, C, r1, b->conv[CV_POST].w, y);
            add_spatial(bs, C, y, b->conv[CV_POST].b, last_skip, last_act);
            free(r1);
        } else if (b->type == BLK_MIXER) { /* :266-312: x <- act(dw(x)+b) + x, then the ffn */
            const int F = b->ffn;
            conv_depthwise(bs, b->conv[CV_DW].k, C, x, b->conv[CV_DW].w, t0);
            add_spatial_post(bs, C, t0, b->conv[CV_DW].b, act, x);
            memcpy(x, t0, sizeof(float) * C * S); /* the new skip */
            conv1(bs, C, F, x, b->conv[CV_1].w, t0);
            add_spatial(bs, F, t0, b->conv[CV_1].b, NULL, act);
            conv1(bs, F, C, t0, b->conv[CV_2].w, y);
            add_spatial(bs, C, y, b->conv[CV_2].b, last_skip, last_act);
        }
        if (b->se) se_unit(bs, C, b, y, x, act); /* :432-446 */
        trunk_probe(i + 1, y, C * S);
    }

    /* policy head :449-536 */
    const int PC = n->pol_ch, VC = n->val_ch;
    const int maxi = PC > VC ? PC : VC;
    float* pconv = (float*)malloc(sizeof(float) * PC * S);
    float* pool = (float*)malloc(sizeof(float) * 3 * maxi);
    float* inter = (float*)malloc(sizeof(float) * 3 * maxi);
    conv1(bs, C, PC, y, n->p_hd.w, pconv);
    add_spatial(bs, PC, pconv, n->p_hd.b, NULL, act);
    if (n->pol_type == 1) { /* RepLK :469-494 */
        float* buf = (float*)malloc(sizeof(float) * PC * S);
        conv_depthwise(bs, n->p_dw.k, PC, pconv, n->p_dw.w, buf);
        add_spatial(bs, PC, buf, n->p_dw.b, NULL, act);
        conv1(bs, PC, PC, buf, n->p_pt.w, pconv);
        add_spatial(bs, PC, pconv, n->p_pt.b, NULL, act);
        free(buf);
    }
    global_pool(bs, PC, pconv, pool, 0);
    fully_connect(&n->p_inter, pool, inter, act);
    add_spatial(bs, PC, pconv, inter, NULL, ACT_IDENTITY);
    conv1(bs, PC, n->prob_ch, pconv, n->prob.w, prob);
    add_spatial(bs, n->prob_ch, prob, n->prob.b, NULL, ACT_IDENTITY);
    fully_connect(&n->pass_fc, inter, pass, ACT_IDENTITY);

    /* value head :538-580 */
    float* vconv = (float*)malloc(sizeof(float) * VC * S);
    conv1(bs, C, VC, y, n->v_hd.w, vconv);
    add_spatial(bs, VC, vconv, n->v_hd.b, NULL, act);
    global_pool(bs, VC, vconv, pool, 1);
    fully_connect(&n->v_inter, pool, inter, act);
    conv1(bs, VC, n->own_ch, vconv, n->v_own.w, own);
    add_spatial(bs, n->own_ch, own, n->v_own.b, NULL, ACT_IDENTITY);
    fully_connect(&n->v_misc, inter, misc, ACT_IDENTITY);

    free(x); free(y); free(t0); free(t1); free(pconv); free(pool); free(inter); free(vconv);
    return 0;
}

/* blas_forward_pipe.cc:565-619 FillOutputs (encoder version: encoder.h / version<=2 -> 1) */
int so_forward(const so_net* n, int bs, float komi, int offset, const float* planes, float* out) {
    (void)komi;
    if (!n) return -1;
    const int S = bs * bs;
    float* prob = (float*)malloc(sizeof(float) * n->prob_ch * S);
    float* own = (float*)malloc(sizeof(float) * n->own_ch * S);
    float pass[8], misc[16];
    if (so_forward_raw(n, bs, planes, prob, pass, misc, own)) { free(prob); free(own); return -1; }
    float* t = out + 2 * S;
    if (n->version <= 2) {
        memcpy(out, prob, sizeof(float) * S);
        t[0] = pass[0];
        t[5] = misc[4];
        t[6] = 0.f;
        t[7] = 0.f;
        t[8] = 0.f;
    } else {
        if (offset < 0 || offset >= n->prob_ch) { free(prob); free(own); return -1; }
        memcpy(out, prob + (long)offset * S, sizeof(float) * S);
        t[0] = pass[offset];
        t[5] = misc[8];
        t[6] = misc[13];
        t[7] = misc[14];
        t[8] = (float)offset;
    }
    memcpy(out + S, own, sizeof(float) * S);
    t[1] = misc[0]; t[2] = misc[1]; t[3] = misc[2]; t[4] = misc[3];
    free(prob); free(own);
    return 0;
}

/* utils/logits.h:22-39 Softmax (double denominator) */
static void softmax(const float* in, float* out, int n, double temp) {
    float alpha = in[0];
    for (int i = 1; i < n; ++i) if (in[i] > alpha) alpha = in[i];
    double denom = 0.0;
    for (int i = 0; i < n; ++i) {
        const double v = exp((in[i] - alpha) / temp);
        denom += v;
        out[i] = (float)v;
    }
    for (int i = 0; i < n; ++i) out[i] = (float)(out[i] / denom);
}

static float softplus_sq(float x) { /* network.cc:399-404 */
    if (x <= 20.f) x = logf(1.f + expf(x));
    return (x * x) / 4.f;
}

/* network.cc:361-429 TransformResult (identity symmetry) + ActivatePolicy */
int so_postprocess(int bs, float temp, const float* raw, float* post) {
    const int S = bs * bs;
    float* logits = (float*)malloc(sizeof(float) * (S + 1));
    memcpy(logits, raw, sizeof(float) * S);
    logits[S] = raw[2 * S + 0];
    softmax(logits, post, S + 1, temp); /* post[0..S) policy, post[S] pass */
    for (int i = 0; i < S; ++i) post[S + 1 + i] = tanhf(raw[S + i]);
    float* t = post + 2 * S + 1;
    softmax(raw + 2 * S + 1, t, 3, 1.0);
    t[3] = (t[0] - t[2] + 1.f) / 2;                 /* wdl_winrate */
    t[4] = (tanhf(raw[2 * S + 4]) + 1.f) / 2;       /* stm_winrate */
    t[5] = 20 * raw[2 * S + 5];                     /* final_score */
    t[6] = (float)(0.25 * softplus_sq(raw[2 * S + 6]));
    t[7] = (float)(150 * softplus_sq(raw[2 * S + 7]));
    free(logits);
    return 0;
}

/* ------------------------------------------------------------------ weight-file loader
 * loader.cc:67-941 + description.cc */
typedef struct {
    const unsigned char* p;
    const unsigned char* end;
} cursor;

typedef struct { char kind; int d[3]; int nd; } shape_t; /* kind: C conv, D depthwise, B bn, F fc */

typedef struct {
    char err[256];
    int failed;
    int binary;
    cursor cur;
    shape_t* shapes;
    int nshapes, ishape;
} loader;

static void fail(loader* L, const char* fmt, ...) {
    if (L->failed) return;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(L->err, sizeof(L->err), fmt, ap);
    va_end(ap);
    L->failed = 1;
}

static int next_line(cursor* c, char* buf, int cap) { /* std::getline */
    if (c->p >= c->end) return 0;
    int n = 0;
    while (c->p < c->end && *c->p != '\n') {
        if (n + 1 < cap) buf[n++] = (char)*c->p;
        c->p++;
    }
    if (c->p < c->end) c->p++;
    buf[n] = 0;
    return 1;
}

static int split_words(char* line, char** words, int cap) { /* utils/splitter */
    int n = 0;
    char* s = line;
    while (*s) {
        while (*s == ' ' || *s == '\t' || *s == '\r') ++s;
        if (!*s) break;
        if (n < cap) words[n++] = s;
        while (*s && *s != ' ' && *s != '\t' && *s != '\r') ++s;
        if (*s) *s++ = 0;
    }
    return n;
}

/* loader.cc:833-898 GetWeightsFromBuffer */
static float* read_tensor(loader* L, long expect, const char* what) {
    if (L->failed) return NULL;
    float* v = (float*)malloc(sizeof(float) * (expect > 0 ? expect : 1));
    long n = 0;
    cursor* c = &L->cur;
    if (L->binary) {
        for (;;) {
            if (c->p + 4 > c->end) { fail(L, "unexpected end of file in %s", what); break; }
            uint32_t bits = (uint32_t)c->p[0] | ((uint32_t)c->p[1] << 8) | ((uint32_t)c->p[2] << 16) |
                            ((uint32_t)c->p[3] << 24);
            c->p += 4;
            if (bits == 0xffffffffu) break;
            if (n < expect) memcpy(&v[n], &bits, 4);
            ++n;
        }
    } else {
        const unsigned char* s = c->p;
        while (c->p < c->end && *c->p != '\n') c->p++;
        const unsigned char* e = c->p;
        if (c->p < c->end) c->p++;
        char* tmp = (char*)malloc((size_t)(e - s) + 1);
        memcpy(tmp, s, (size_t)(e - s));
        tmp[e - s] = 0;
        char* q = tmp;
        for (;;) {
            char* endp;
            double d = strtod(q, &endp);
            if (endp == q) break;
            if (n < expect) v[n] = (float)d;
            ++n;
            q = endp;
        }
        free(tmp);
    }
    if (!L->failed && n != expect) fail(L, "%s: expect %ld values but got %ld", what, expect, n);
    if (L->failed) { free(v); return NULL; }
    return v;
}

static shape_t* take_shape(loader* L, char kind1, char kind2) {
    if (L->failed) return NULL;
    if (L->ishape >= L->nshapes) { fail(L, "struct list exhausted"); return NULL; }
    shape_t* s = &L->shapes[L->ishape++];
    if (s->kind != kind1 && s->kind != kind2) { fail(L, "unexpected layer kind '%c' in struct", s->kind); return NULL; }
    return s;
}

/* FillConvolutionLayer + FillBatchnormLayer (loader.cc:914-940) and the fold of
 * ProcessWeights (loader.cc:775-793); BN stddev handling description.h:44-54 */
static void load_conv_bn(loader* L, const so_net* n, so_conv* cv, int with_bn) {
    shape_t* s = take_shape(L, 'C', 'D');
    if (!s) return;
    cv->cin = s->d[0]; cv->cout = s->d[1]; cv->k = s->d[2];
    cv->nw = (long)cv->cin * cv->cout * cv->k * cv->k;
    cv->w = read_tensor(L, cv->nw, "conv weights");
    cv->b = read_tensor(L, cv->cout, "conv biases");
    cv->u = NULL;
    if (!with_bn || L->failed) return;
    shape_t* bs = take_shape(L, 'B', 'B');
    if (!bs) return;
    if (bs->d[0] != cv->cout) { fail(L, "batchnorm channels mismatch"); return; }
    float* mean = read_tensor(L, cv->cout, "bn means");
    float* std = read_tensor(L, cv->cout, "bn stddevs");
    if (L->failed) { free(mean); free(std); return; }
    const long stride = cv->nw / cv->cout;
    for (int o = 0; o < cv->cout; ++o) {
        const float scale = n->version == 1 ? 1.0f / sqrtf(std[o] + 1e-5f) : 1.0f / std[o];
        cv->b[o] -= mean[o];
        for (long k = 0; k < stride; ++k) cv->w[stride * o + k] *= scale;
        cv->b[o] *= scale;
    }
    free(mean); free(std);
    if (cv->k == 3) cv->u = wino_transform_f(cv->w, cv->cout, cv->cin);
}

static void load_fc(loader* L, so_fc* fc) { /* loader.cc:900-912 */
    shape_t* s = take_shape(L, 'F', 'F');
    if (!s) return;
    fc->cin = s->d[0]; fc->cout = s->d[1];
    fc->w = read_tensor(L, (long)fc->cin * fc->cout, "fc weights");
    fc->b = read_tensor(L, fc->cout, "fc biases");
}

static const char* info_get(char keys[][48], char vals[][48], int n, const char* key) {
    for (int i = 0; i < n; ++i) if (!strcmp(keys[i], key)) return vals[i];
    return NULL;
}

static void free_conv(so_conv* c) { free(c->w); free(c->b); free(c->u); }
static void free_fc(so_fc* f) { free(f->w); free(f->b); }

void so_free(so_net* n) {
    if (!n) return;
    free_conv(&n->input); free_conv(&n->p_hd); free_conv(&n->p_dw); free_conv(&n->p_pt);
    free_conv(&n->prob); free_conv(&n->v_hd); free_conv(&n->v_own);
    free_fc(&n->p_inter); free_fc(&n->pass_fc); free_fc(&n->v_inter); free_fc(&n->v_misc);
    for (int i = 0; n->tower && i < n->nblocks; ++i) {
        for (int j = 0; j < CV_COUNT; ++j) free_conv(&n->tower[i].conv[j]);
        free_fc(&n->tower[i].squeeze); free_fc(&n->tower[i].excite);
    }
    free(n->tower);
    free(n);
}

so_net* so_load(const char* path, int winograd, char* err, int errlen) {
    loader L;
    memset(&L, 0, sizeof(L));
    so_net* n = NULL;
    unsigned char* data = NULL;
    char (*stack)[48] = NULL;
    int nstack = 0;
    char keys[64][48], vals[64][48];
    int ninfo = 0;
    char line[512];
    char* words[8];

    FILE* f = fopen(path, "rb");
    if (!f) { fail(&L, "couldn't open weights file %s", path); goto done; }
    fseek(f, 0, SEEK_END);
    long size = ftell(f);
    fseek(f, 0, SEEK_SET);
    data = (unsigned char*)malloc((size_t)size + 1);
    if (fread(data, 1, (size_t)size, f) != (size_t)size) { fclose(f); fail(&L, "short read"); goto done; }
    fclose(f);
    L.cur.p = data;
    L.cur.end = data + size;

    /* loader.cc:85-116 */
    if (!next_line(&L.cur, line, sizeof(line))) { fail(&L, "weights file is empty"); goto done; }
    if (split_words(line, words, 8) < 2 || strcmp(words[0], "get") || strcmp(words[1], "main")) {
        fail(&L, "weights file format is not acceptable"); goto done;
    }
    L.shapes = (shape_t*)calloc(4096, sizeof(shape_t));
    stack = (char(*)[48])calloc(1024, 48);
    int in_params = 0;
    while (!in_params && next_line(&L.cur, line, sizeof(line))) {
        int nw = split_words(line, words, 8);
        if (nw < 2 || strcmp(words[0], "get")) continue;
        if (!strcmp(words[1], "info")) { /* :123-134 */
            while (next_line(&L.cur, line, sizeof(line))) {
                nw = split_words(line, words, 8);
                if (nw == 0) continue;
                if (words[0][0] == '#') continue;
                if (!strcmp(words[0], "end")) break;
                if (nw >= 2 && ninfo < 64) {
                    snprintf(keys[ninfo], 48, "%s", words[0]);
                    snprintf(vals[ninfo], 48, "%s", words[1]);
                    ++ninfo;
                }
            }
        } else if (!strcmp(words[1], "stack")) { /* :136-147 */
            while (next_line(&L.cur, line, sizeof(line))) {
                nw = split_words(line, words, 8);
                if (nw == 0) continue;
                if (words[0][0] == '#') continue;
                if (!strcmp(words[0], "end")) break;
                if (nstack < 1024) snprintf(stack[nstack++], 48, "%s", words[0]);
            }
        } else if (!strcmp(words[1], "struct")) { /* :149-188 */
            while (next_line(&L.cur, line, sizeof(line))) {
                nw = split_words(line, words, 8);
                if (nw == 0) continue;
                if (words[0][0] == '#') continue;
                if (!strcmp(words[0], "end")) break;
                shape_t* s = &L.shapes[L.nshapes];
                s->nd = nw - 1;
                for (int i = 1; i < nw && i <= 3; ++i) s->d[i - 1] = atoi(words[i]);
                if (!strcmp(words[0], "FullyConnect") && s->nd == 2) s->kind = 'F';
                else if (!strcmp(words[0], "Convolution") && s->nd == 3) s->kind = 'C';
                else if (!strcmp(words[0], "DepthwiseConvolution") && s->nd == 3) s->kind = 'D';
                else if (!strcmp(words[0], "BatchNorm") && s->nd == 1) s->kind = 'B';
                else { fail(&L, "layer shape is error"); goto done; }
                if (++L.nshapes >= 4096) { fail(&L, "too many layers"); goto done; }
            }
        } else if (!strcmp(words[1], "parameters")) {
            in_params = 1;
        }
    }

    n = (so_net*)calloc(1, sizeof(so_net));
    n->winograd = winograd;
    /* CheckMisc loader.cc:190-316 */
    n->version = 1;
    const char* v;
    if ((v = info_get(keys, vals, ninfo, "FloatType")) && !strcmp(v, "float32bin")) L.binary = 1;
    if ((v = info_get(keys, vals, ninfo, "Version"))) n->version = atoi(v);
    if (n->version >= 6) { fail(&L, "do not support this version"); goto done; }
    if (n->version >= 3) { n->in_ch = 43; n->prob_ch = 5; n->pass_outs = 5; n->own_ch = 1; n->misc_outs = 15; }
    else { n->in_ch = 38; n->prob_ch = 1; n->pass_outs = 1; n->own_ch = 1; n->misc_outs = 5; }
    n->pol_type = 0;
    if ((v = info_get(keys, vals, ninfo, "PolicyHeadType"))) {
        if (!strcasecmp(v, "normal")) n->pol_type = 0;
        else if (!strcasecmp(v, "replk")) n->pol_type = 1;
        else { fail(&L, "unknown policy head type"); goto done; }
    }
    n->act = ACT_RELU;
    if ((v = info_get(keys, vals, ninfo, "ActivationFunction"))) {
        n->act = act_from_string(v);
        if (n->act < 0) { fail(&L, "Unknown activation type."); goto done; }
    }
    if (!(v = info_get(keys, vals, ninfo, "ResidualBlocks"))) { fail(&L, "no ResidualBlocks"); goto done; }
    n->nblocks = atoi(v);
    if (!(v = info_get(keys, vals, ninfo, "ResidualChannels"))) { fail(&L, "no ResidualChannels"); goto done; }
    n->channels = atoi(v);
    const char* pk = n->version >= 5 ? "PolicyHeadChannels" : "PolicyExtract";
    const char* vk = n->version >= 5 ? "ValueHeadChannels" : "ValueExtract";
    if (!(v = info_get(keys, vals, ninfo, pk))) { fail(&L, "no %s", pk); goto done; }
    n->pol_ch = atoi(v);
    if (!(v = info_get(keys, vals, ninfo, vk))) { fail(&L, "no %s", vk); goto done; }
    n->val_ch = atoi(v);
    if (!(v = info_get(keys, vals, ninfo, "InputChannels")) || atoi(v) != n->in_ch) {
        fail(&L, "the number of input channels is wrong"); goto done;
    }
    if (nstack == 0) { /* :267-292 legacy files without a stack: ResidualBlock[-SE] only */
        int inner = 0;
        for (int b = 0; b < n->nblocks; ++b) {
            inner += 4;
            if (inner + 2 < L.nshapes && L.shapes[inner + 2].kind == 'F') {
                snprintf(stack[nstack++], 48, "ResidualBlock-SE");
                inner += 2;
            } else {
                snprintf(stack[nstack++], 48, "ResidualBlock");
            }
        }
        if (L.nshapes != 10 + inner + 2) { fail(&L, "do not support this weights format"); goto done; }
    }
    if (nstack < n->nblocks) { fail(&L, "stack shorter than ResidualBlocks"); goto done; }

    /* FillWeights loader.cc:628-773 */
    load_conv_bn(&L, n, &n->input, 1);
    if (!L.failed && (n->input.cin != n->in_ch || n->input.cout != n->channels || n->input.k != 3))
        fail(&L, "the input layers are wrong");
    n->tower = (so_block*)calloc((size_t)(n->nblocks > 0 ? n->nblocks : 1), sizeof(so_block));
    for (int bi = 0; bi < n->nblocks && !L.failed; ++bi) { /* FillBlock :358-626 */
        so_block* b = &n->tower[bi];
        char name[48];
        snprintf(name, sizeof(name), "%s", stack[bi]);
        int has_se = 0;
        char* tok = strtok(name, "-");
        while (tok) {
            if (!strcmp(tok, "ResidualBlock")) b->type = BLK_RESIDUAL;
            else if (!strcmp(tok, "BottleneckBlock")) b->type = BLK_BOTTLENECK;
            else if (!strcmp(tok, "NestedBottleneckBlock")) b->type = BLK_NESTED;
            else if (!strcmp(tok, "MixerBlock")) b->type = BLK_MIXER;
            else if (!strcmp(tok, "SE")) has_se = 1;
            else if (!strcmp(tok, "FixUp")) {}
            else fail(&L, "do not support this block type [%s]", stack[bi]);
            tok = strtok(NULL, "-");
        }
        b->se = has_se;
        const int C = n->channels;
        if (b->type == BLK_RESIDUAL) {
            load_conv_bn(&L, n, &b->conv[CV_1], 1);
            load_conv_bn(&L, n, &b->conv[CV_2], 1);
            if (!L.failed && (b->conv[CV_1].cin != C || b->conv[CV_1].cout != C || b->conv[CV_2].cin != C ||
                              b->conv[CV_2].cout != C || b->conv[CV_1].k != 3 || b->conv[CV_2].k != 3))
                fail(&L, "the residual block is wrong");
        } else if (b->type == BLK_BOTTLENECK || b->type == BLK_NESTED) {
            load_conv_bn(&L, n, &b->conv[CV_PRE], 1);
            load_conv_bn(&L, n, &b->conv[CV_1], 1);
            load_conv_bn(&L, n, &b->conv[CV_2], 1);
            if (b->type == BLK_NESTED) {
                load_conv_bn(&L, n, &b->conv[CV_3], 1);
                load_conv_bn(&L, n, &b->conv[CV_4], 1);
            }
            load_conv_bn(&L, n, &b->conv[CV_POST], 1);
            b->inner = b->conv[CV_PRE].cout;
            if (!L.failed && (b->conv[CV_PRE].cin != C || b->conv[CV_POST].cout != C || b->conv[CV_PRE].k != 1 ||
                              b->conv[CV_POST].k != 1 || b->conv[CV_1].k != 3 || b->conv[CV_2].k != 3 ||
                              b->conv[CV_1].cin != b->inner || b->conv[CV_1].cout != b->inner ||
                              b->conv[CV_2].cin != b->inner || b->conv[CV_2].cout != b->inner))
                fail(&L, "the bottleneck block is wrong");
        } else if (b->type == BLK_MIXER) {
            load_conv_bn(&L, n, &b->conv[CV_DW], 1);
            load_conv_bn(&L, n, &b->conv[CV_1], 1);
            load_conv_bn(&L, n, &b->conv[CV_2], 1);
            b->ffn = b->conv[CV_1].cout;
            if (!L.failed && (b->conv[CV_DW].cout != C || b->conv[CV_1].cin != C || b->conv[CV_2].cout != C ||
                              b->conv[CV_1].k != 1 || b->conv[CV_2].k != 1))
                fail(&L, "the mixer block is wrong");
        } else {
            fail(&L, "need the ResidualBlock, BottleneckBlock, NestedBottleneckBlock or MixerBlock");
        }
        if (b->se && !L.failed) {
            load_fc(&L, &b->squeeze);
            load_fc(&L, &b->excite);
            b->se_size = b->squeeze.cout;
            if (!L.failed && (b->squeeze.cin != 3 * C || b->excite.cout != 2 * C))
                fail(&L, "the SE module size is wrong");
        }
    }
    load_conv_bn(&L, n, &n->p_hd, 1);
    if (n->pol_type == 1) {
        load_conv_bn(&L, n, &n->p_dw, 1);
        load_conv_bn(&L, n, &n->p_pt, 1);
    }
    load_fc(&L, &n->p_inter);
    load_conv_bn(&L, n, &n->prob, 0);
    load_fc(&L, &n->pass_fc);
    load_conv_bn(&L, n, &n->v_hd, 1);
    load_fc(&L, &n->v_inter);
    load_conv_bn(&L, n, &n->v_own, 0);
    load_fc(&L, &n->v_misc);
    if (!L.failed) {
        if (n->p_hd.k != 1 || n->prob.k != 1) fail(&L, "the policy convolution kernel size is wrong");
        else if (n->prob.cout != n->prob_ch) fail(&L, "the number of policy ouput size is wrong");
        else if (n->p_inter.cout != n->pass_fc.cin || n->p_inter.cin != 3 * n->pol_ch || n->p_inter.cout != n->pol_ch)
            fail(&L, "the number of policy fully connect size is wrong");
        else if (n->pass_fc.cout != n->pass_outs) fail(&L, "the number of pass ouput size is wrong");
        else if (n->v_hd.k != 1 || n->v_own.k != 1) fail(&L, "the value convolution kernel size is wrong");
        else if (n->v_own.cout != n->own_ch) fail(&L, "the number of ownership ouput size is wrong");
        else if (n->v_inter.cout != n->v_misc.cin || n->v_inter.cin != 3 * n->val_ch || n->v_inter.cout != 3 * n->val_ch)
            fail(&L, "the number of value fully connect size is wrong");
        else if (n->v_misc.cout != n->misc_outs) fail(&L, "the misc value layer size is wrong.");
    }
    if (!L.failed) { /* :763-768 */
        if (!next_line(&L.cur, line, sizeof(line)) || split_words(line, words, 8) < 1 || strcmp(words[0], "end"))
            fail(&L, "weights file format is not acceptable");
    }

done:
    free(data);
    free(L.shapes);
    free(stack);
    if (L.failed) {
        if (err && errlen > 0) snprintf(err, (size_t)errlen, "%s", L.err);
        so_free(n);
        return NULL;
    }
    if (err && errlen > 0) err[0] = 0;
    return n;
}

int so_info(const so_net* n, int* info) {
    if (!n) return -1;
    info[0] = n->version; info[1] = n->in_ch; info[2] = n->nblocks; info[3] = n->channels;
    info[4] = n->pol_ch; info[5] = n->val_ch; info[6] = n->prob_ch; info[7] = n->pass_outs;
    info[8] = n->own_ch; info[9] = n->misc_outs; info[10] = n->act; info[11] = n->pol_type;
    return 0;
}

int so_block_info(const so_net* n, int idx, int* binfo) {
    if (!n || idx < 0 || idx >= n->nblocks) return -1;
    const so_block* b = &n->tower[idx];
    binfo[0] = b->type; binfo[1] = b->se; binfo[2] = b->se_size; binfo[3] = b->inner; binfo[4] = b->ffn;
    return 0;
}

long so_get_tensor(const so_net* n, const char* name, float* dst, long cap) {
    if (!n) return -1;
    char path[96];
    snprintf(path, sizeof(path), "%s", name);
    char* dot = strrchr(path, '.');
    if (!dot) return -1;
    *dot = 0;
    const char kind = dot[1];
    const so_conv* cv = NULL;
    const so_fc* fc = NULL;
    const char* lname = path;
    const so_block* b = NULL;
    if (!strncmp(path, "tower.", 6)) {
        char* d2 = strchr(path + 6, '.');
        if (!d2) return -1;
        *d2 = 0;
        const int idx = atoi(path + 6);
        if (idx < 0 || idx >= n->nblocks) return -1;
        b = &n->tower[idx];
        lname = d2 + 1;
        if (!strcmp(lname, "conv1")) cv = &b->conv[CV_1];
        else if (!strcmp(lname, "conv2")) cv = &b->conv[CV_2];
        else if (!strcmp(lname, "conv3")) cv = &b->conv[CV_3];
        else if (!strcmp(lname, "conv4")) cv = &b->conv[CV_4];
        else if (!strcmp(lname, "pre_btl_conv")) cv = &b->conv[CV_PRE];
        else if (!strcmp(lname, "post_btl_conv")) cv = &b->conv[CV_POST];
        else if (!strcmp(lname, "dw_conv")) cv = &b->conv[CV_DW];
        else if (!strcmp(lname, "squeeze")) fc = &b->squeeze;
        else if (!strcmp(lname, "excite")) fc = &b->excite;
    } else {
        if (!strcmp(lname, "input_conv")) cv = &n->input;
        else if (!strcmp(lname, "p_hd_conv")) cv = &n->p_hd;
        else if (!strcmp(lname, "p_dw_conv")) cv = &n->p_dw;
        else if (!strcmp(lname, "p_pt_conv")) cv = &n->p_pt;
        else if (!strcmp(lname, "prob_conv")) cv = &n->prob;
        else if (!strcmp(lname, "v_hd_conv")) cv = &n->v_hd;
        else if (!strcmp(lname, "v_ownership")) cv = &n->v_own;
        else if (!strcmp(lname, "p_inter_fc")) fc = &n->p_inter;
        else if (!strcmp(lname, "pass_fc")) fc = &n->pass_fc;
        else if (!strcmp(lname, "v_inter_fc")) fc = &n->v_inter;
        else if (!strcmp(lname, "v_misc")) fc = &n->v_misc;
    }
    const float* src = NULL;
    long cnt = 0;
    if (cv) {
        if (kind == 'w') { src = cv->w; cnt = cv->nw; }
        else if (kind == 'b') { src = cv->b; cnt = cv->cout; }
        else if (kind == 'u') { src = cv->u; cnt = cv->u ? 36L * cv->cin * cv->cout : 0; }
    } else if (fc) {
        if (kind == 'w') { src = fc->w; cnt = (long)fc->cin * fc->cout; }
        else if (kind == 'b') { src = fc->b; cnt = fc->cout; }
    }
    if (!src) return cv || fc ? 0 : -1;
    if (dst) memcpy(dst, src, sizeof(float) * (size_t)(cnt < cap ? cnt : cap));
    return cnt;
}
