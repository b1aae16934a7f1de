// MFMA issue-rate micro-benchmark on random (non-zero) fp16 operands, 2 waves per SIMD (512-thread workgroups, one per CU),
// independent accumulators: v_mfma_f32_16x16x32_f16 (48 accumulators of 4 regs) vs v_mfma_f32_32x32x16_f16 (12 of 16 regs).
// Answers: how much faster per flop is the 32x32x16 form under the chip's power limit?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef _Float16 f16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ __launch_bounds__(512, 2) void k16(const f16x8* __restrict__ src, float* __restrict__ out, int iters) {
    const int tid = threadIdx.x;
    f16x8 a[4], b[3];
    for (int i = 0; i < 4; ++i) a[i] = src[(blockIdx.x * 512 + tid) * 8 + i];
    for (int i = 0; i < 3; ++i) b[i] = src[(blockIdx.x * 512 + tid) * 8 + 4 + i];
    f32x4 acc[48];
    for (int i = 0; i < 48; ++i) acc[i] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 12; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i * 12 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j % 3], acc[i * 12 + j], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 48; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * 512 + tid] = s;
}
__global__ __launch_bounds__(512, 2) void k32(const f16x8* __restrict__ src, float* __restrict__ out, int iters) {
    const int tid = threadIdx.x;
    f16x8 a[4], b[3];
    for (int i = 0; i < 4; ++i) a[i] = src[(blockIdx.x * 512 + tid) * 8 + i];
    for (int i = 0; i < 3; ++i) b[i] = src[(blockIdx.x * 512 + tid) * 8 + 4 + i];
    f32x16 acc[12];
    for (int i = 0; i < 12; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0;
    for (int it = 0; it < iters; ++it) {
        // same flops per iteration as k16: 48 x 16x16x32 = 24 x 32x32x16
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 12; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(j + s) & 3], b[j % 3], acc[j], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 12; ++i) s += acc[i][0] + acc[i][15];
    out[blockIdx.x * 512 + tid] = s;
}

// the same 48 x 16x16x32 per iteration with EIGHT different B fragments (k16 re-reads three): operand data changes from one
// MFMA to the next, as in a real K loop -- what do fresh operands cost at the power limit?
__global__ __launch_bounds__(512, 2) void k16r(const f16x8* __restrict__ src, float* __restrict__ out, int iters) {
    const int tid = threadIdx.x;
    f16x8 a[4], b[8];
    for (int i = 0; i < 4; ++i) a[i] = src[(blockIdx.x * 512 + tid) * 8 + i];
    for (int i = 0; i < 8; ++i) b[i] = src[((blockIdx.x * 512 + tid) * 8 + 4 + i * 131) % (256 * 512 * 8)];
    f32x4 acc[48];
    for (int i = 0; i < 48; ++i) acc[i] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 12; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i * 12 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[(j * 4 + i) % 8], acc[i * 12 + j], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 48; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * 512 + tid] = s;
}

template <typename K> double run(K kern, const f16x8* d, float* o, int iters, const char* name) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    kern<<<256, 512>>>(d, o, 10);
    hipDeviceSynchronize();
    hipEventRecord(a);
    kern<<<256, 512>>>(d, o, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double flops = 256.0 * 8 * iters * 48 * (16.0 * 16 * 32 * 2);
    printf("%s: %.3f ms, %.0f TFLOP/s (%.1f %% of 2.5 PF)\n", name, ms, flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 2.5e15 * 100);
    return ms;
}

int main(int argc, char** argv) {
    const size_t n = 256 * 512 * 8;
    std::vector<f16x8> h(n);
    unsigned s = 12345;
    for (auto& v : h)
        for (int e = 0; e < 8; ++e) { s = s * 1664525u + 1013904223u; v[e] = (f16)(((int)(s >> 16) % 2001 - 1000) / 1000.0f); }
    f16x8* d; float* o;
    hipMalloc(&d, n * sizeof(f16x8)); hipMalloc(&o, 256 * 512 * 4);
    hipMemcpy(d, h.data(), n * sizeof(f16x8), hipMemcpyHostToDevice);
    if (argc > 1) {
        // sustained mode: `mfma_rate.so <seconds> [zero]` keeps the 16x16x32 loop running (one line per ~0.25 s burst) so that
        // rocm-smi can be sampled beside it (tools/gpu/r03_evidence.sh)
        const double secs = atof(argv[1]);
        if (argc > 2 && !strcmp(argv[2], "zero")) hipMemset(d, 0, n * sizeof(f16x8));
        double total = 0;
        const char* mode = argc > 2 ? argv[2] : "random";
        if (!strcmp(mode, "k32")) { while (total < secs * 1e3) total += run(k32, d, o, 200000, "sustained 32x32x16 f16, random operands"); return 0; }
        if (!strcmp(mode, "rot")) { while (total < secs * 1e3) total += run(k16r, d, o, 200000, "sustained 16x16x32 f16, random operands, 8 rotating B fragments"); return 0; }
        while (total < secs * 1e3) total += run(k16, d, o, 200000, argc > 2 ? "sustained 16x16x32 f16, zero operands" : "sustained 16x16x32 f16, random operands");
        return 0;
    }
    for (int rep = 0; rep < 2; ++rep) {
        run(k16, d, o, 4000, "16x16x32 f16, 2 waves/SIMD");
        run(k32, d, o, 4000, "32x32x16 f16, 2 waves/SIMD");
    }
    hipMemset(d, 0, n * sizeof(f16x8));
    run(k16, d, o, 4000, "16x16x32 f16, zero operands");
    run(k32, d, o, 4000, "32x32x16 f16, zero operands");
    return 0;
}
