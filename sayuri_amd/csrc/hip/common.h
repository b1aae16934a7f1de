// Shared device helpers for the gfx950 forward-pipe kernels.
//
// Activation storage layout used by every kernel in this directory ("compact NHWC"):
//   act[sample n][pixel p][channel c],  p = y*bs_n + x with the sample's OWN board size,
//   sample stride = slot_pix*cs elements, pixel stride = cs (channel stride, multiple of 32;
//   channels >= the layer's real count are kept at exactly 0).
// A sample smaller than the NN grid therefore has no off-board cells at all: the reference's
// per-layer mask multiply (cuda_kernels.cu:57-59) has nothing to zero, and a masked evaluation
// on the padded grid equals the native small-board evaluation (SURVEY.md 8c probe).
#pragma once
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sayuri {

// Zero bytes every activation buffer carries in front of row 0 (the halo source of conv_board.h's 32-bit-offset DMA).
constexpr int kZeroPrefix = 4096;

enum Act : int { kIdentity = 0, kReLU, kELU, kSELU, kGELU, kMish, kSwish, kHardSwish };

typedef _Float16 f16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }

// All eight activations of the reference (src/neural/activation.h:41-81).  Mish is evaluated
// as x*(1 - 2/(e^2 + 2e + 2)), e = e^x, which equals x*tanh(log(1+e^x)) (tanh(log(1+e)) = n/(n+2) with
// n = e(e+2)) and costs one exp, one rcp and five VALU operations instead of exp+log+tanh.
__device__ __forceinline__ float activate(float x, int act) {
    switch (act) {
    case kReLU: return x > 0.f ? x : 0.f;
    case kELU: return x > 0.f ? x : (__expf(x) - 1.f);
    case kSELU: return x > 0.f ? (1.05070098f * x) : (1.05070098f * 1.67326324f * (__expf(x) - 1.0f));
    case kGELU: {
        const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
        // tanh(u) = 1 - 2/(e^{2u}+1)
        const float t = 1.f - 2.f / (fast_exp(2.f * u) + 1.f);
        return 0.5f * x * (1.0f + t);
    }
    case kMish: {
        // the guard is not needed for the value (e = inf gives rcp = 0 and the result x) but keeps the generated code in
        // one shape for every element: without it hipcc mixes packed and scalar sequences whose last bits differ, and
        // a sample's result then depends on the slot it occupies in the batch (tests/test_gpu_net.py catches that)
        if (x > 20.f) return x;
        const float e = fast_exp(x);
        const float r = __builtin_amdgcn_rcpf(__builtin_fmaf(e, e + 2.f, 2.f));
        return x * __builtin_fmaf(-2.f, r, 1.f);
    }
    case kSwish: return x / (1.0f + fast_exp(-x));
    case kHardSwish: return x >= 3.f ? x : (x <= -3.f ? 0.f : (x * (x + 3.0f) / 6.0f));
    default: return x;
    }
}

template <typename T> struct ElemTraits;
template <> struct ElemTraits<f16> {
    static constexpr int kPieceElems = 8;  // elements per 16-byte piece
};
template <> struct ElemTraits<float> {
    static constexpr int kPieceElems = 4;
};

__device__ __forceinline__ float to_float(f16 v) { return (float)v; }
__device__ __forceinline__ float to_float(float v) { return v; }
template <typename T> __device__ __forceinline__ T from_float(float v);
template <> __device__ __forceinline__ f16 from_float<f16>(float v) { return (f16)v; }
template <> __device__ __forceinline__ float from_float<float>(float v) { return v; }

// Batch geometry shared by all kernels of one forward call.
struct BatchGeom {
    const int* sample_off;  // [n_samples+1] prefix sums of bs_n^2 (device)
    const int* bsz;         // [n_samples] board size per sample (device)
    int n_samples;
    int total_pix;          // sample_off[n_samples]
    int slot_pix;           // pixels reserved per sample slot (= board*board of the NN grid)
};

}  // namespace sayuri
