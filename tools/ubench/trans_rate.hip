// Transcendental issue rate on gfx950: how many cycles does a wave64 v_exp_f32 / v_rcp_f32 / v_exp_f16 / v_rcp_f16 take, do plain
// VALU operations run in its shadow, and what do two waves per SIMD get?  (The board kernels' epilogue is Mish: one exp and one rcp
// per value, sixteen of them per 16-byte store.)  One workgroup of 512 threads per CU (2 waves per SIMD) or 256 (1 wave per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define REP8(x) x x x x x x x x
#define HIP_OK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("%s failed: %s\n", #e, hipGetErrorString(_e)); exit(1); } } while (0)

// 8 independent chains, ITER iterations of: 8 x OP  [+ 8 x NPLAIN plain fmas on other registers]
#define KERNEL(name, OPS)                                                                                  \
    __global__ __launch_bounds__(512) void name(float* out, int iters) {                                   \
        float v0 = threadIdx.x * 1e-3f + 1.f, v1 = v0 + .1f, v2 = v0 + .2f, v3 = v0 + .3f, v4 = v0 + .4f, v5 = v0 + .5f, \
              v6 = v0 + .6f, v7 = v0 + .7f;                                                                \
        float p0 = v0, p1 = v1, p2 = v2, p3 = v3, p4 = v4, p5 = v5, p6 = v6, p7 = v7;                     \
        for (int it = 0; it < iters; ++it) {                                                               \
            asm volatile(OPS : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7), "+v"(p0), "+v"(p1), \
                         "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7));                     \
        }                                                                                                  \
        out[blockIdx.x * blockDim.x + threadIdx.x] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + p0 + p1 + p2 + p3 + p4 + p5 + p6 + p7; \
    }
#define T8(op) op " %0, %0\n" op " %1, %1\n" op " %2, %2\n" op " %3, %3\n" op " %4, %4\n" op " %5, %5\n" op " %6, %6\n" op " %7, %7\n"
#define P8 "v_fma_f32 %8, %8, %8, %8\nv_fma_f32 %9, %9, %9, %9\nv_fma_f32 %10, %10, %10, %10\nv_fma_f32 %11, %11, %11, %11\n" \
           "v_fma_f32 %12, %12, %12, %12\nv_fma_f32 %13, %13, %13, %13\nv_fma_f32 %14, %14, %14, %14\nv_fma_f32 %15, %15, %15, %15\n"
// interleaved: one transcendental, one plain fma, ...
#define I8(op) op " %0, %0\nv_fma_f32 %8, %8, %8, %8\n" op " %1, %1\nv_fma_f32 %9, %9, %9, %9\n" op " %2, %2\nv_fma_f32 %10, %10, %10, %10\n" \
               op " %3, %3\nv_fma_f32 %11, %11, %11, %11\n" op " %4, %4\nv_fma_f32 %12, %12, %12, %12\n" op " %5, %5\nv_fma_f32 %13, %13, %13, %13\n" \
               op " %6, %6\nv_fma_f32 %14, %14, %14, %14\n" op " %7, %7\nv_fma_f32 %15, %15, %15, %15\n"
#define I8x3(op) op " %0, %0\nv_fma_f32 %8, %8, %8, %8\nv_fma_f32 %9, %9, %9, %9\nv_fma_f32 %10, %10, %10, %10\n" \
                 op " %1, %1\nv_fma_f32 %11, %11, %11, %11\nv_fma_f32 %12, %12, %12, %12\nv_fma_f32 %13, %13, %13, %13\n" \
                 op " %2, %2\nv_fma_f32 %14, %14, %14, %14\nv_fma_f32 %15, %15, %15, %15\nv_fma_f32 %8, %8, %8, %8\n" \
                 op " %3, %3\nv_fma_f32 %9, %9, %9, %9\nv_fma_f32 %10, %10, %10, %10\nv_fma_f32 %11, %11, %11, %11\n" \
                 op " %4, %4\nv_fma_f32 %12, %12, %12, %12\nv_fma_f32 %13, %13, %13, %13\nv_fma_f32 %14, %14, %14, %14\n" \
                 op " %5, %5\nv_fma_f32 %15, %15, %15, %15\nv_fma_f32 %8, %8, %8, %8\nv_fma_f32 %9, %9, %9, %9\n" \
                 op " %6, %6\nv_fma_f32 %10, %10, %10, %10\nv_fma_f32 %11, %11, %11, %11\nv_fma_f32 %12, %12, %12, %12\n" \
                 op " %7, %7\nv_fma_f32 %13, %13, %13, %13\nv_fma_f32 %14, %14, %14, %14\nv_fma_f32 %15, %15, %15, %15\n"
KERNEL(k_fma, P8)
KERNEL(k_exp32, T8("v_exp_f32"))
KERNEL(k_rcp32, T8("v_rcp_f32"))
KERNEL(k_exp16, T8("v_exp_f16"))
KERNEL(k_rcp16, T8("v_rcp_f16"))
KERNEL(k_exp32_fma, I8("v_exp_f32"))
KERNEL(k_exp32_fma3, I8x3("v_exp_f32"))
KERNEL(k_rcp32_fma3, I8x3("v_rcp_f32"))
KERNEL(k_exp16_fma3, I8x3("v_exp_f16"))

#define PK8 "v_pk_fma_f32 %[a], %[a], %[a], %[a]\nv_pk_fma_f32 %[b], %[b], %[b], %[b]\nv_pk_fma_f32 %[c], %[c], %[c], %[c]\nv_pk_fma_f32 %[d], %[d], %[d], %[d]\n" \
            "v_pk_fma_f32 %[e], %[e], %[e], %[e]\nv_pk_fma_f32 %[f], %[f], %[f], %[f]\nv_pk_fma_f32 %[g], %[g], %[g], %[g]\nv_pk_fma_f32 %[h], %[h], %[h], %[h]\n"
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define KERNEL2(name, OPS)                                                                                 \
    __global__ __launch_bounds__(512) void name(float* out, int iters) {                                   \
        f32x2 a = {threadIdx.x * 1e-3f, 1.f}, b = a + .1f, c = a + .2f, d = a + .3f, e = a + .4f, f = a + .5f, g = a + .6f, h = a + .7f; \
        for (int it = 0; it < iters; ++it)                                                                 \
            asm volatile(OPS : [a] "+v"(a), [b] "+v"(b), [c] "+v"(c), [d] "+v"(d), [e] "+v"(e), [f] "+v"(f), [g] "+v"(g), [h] "+v"(h)); \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a[0] + b[1] + c[0] + d[1] + e[0] + f[1] + g[0] + h[1];  \
    }
KERNEL2(k_pkfma, PK8)
#define PKM8 "v_pk_mul_f32 %[a], %[a], %[a]\nv_pk_mul_f32 %[b], %[b], %[b]\nv_pk_mul_f32 %[c], %[c], %[c]\nv_pk_mul_f32 %[d], %[d], %[d]\n" \
             "v_pk_mul_f32 %[e], %[e], %[e]\nv_pk_mul_f32 %[f], %[f], %[f]\nv_pk_mul_f32 %[g], %[g], %[g]\nv_pk_mul_f32 %[h], %[h], %[h]\n"
KERNEL2(k_pkmul, PKM8)
KERNEL(k_cvt, T8("v_cvt_f32_f16"))
KERNEL(k_mov, T8("v_mov_b32"))
KERNEL(k_swap, "s_nop 1\nv_permlane16_swap_b32 %0, %1\nv_permlane16_swap_b32 %2, %3\nv_permlane16_swap_b32 %4, %5\nv_permlane16_swap_b32 %6, %7\n"
               "s_nop 1\nv_permlane16_swap_b32 %8, %9\nv_permlane16_swap_b32 %10, %11\nv_permlane16_swap_b32 %12, %13\nv_permlane16_swap_b32 %14, %15\n")
KERNEL(k_cvtpk, "v_cvt_pk_f16_f32 %0, %0, %1\nv_cvt_pk_f16_f32 %2, %2, %3\nv_cvt_pk_f16_f32 %4, %4, %5\nv_cvt_pk_f16_f32 %6, %6, %7\n"
                "v_cvt_pk_f16_f32 %8, %8, %9\nv_cvt_pk_f16_f32 %10, %10, %11\nv_cvt_pk_f16_f32 %12, %12, %13\nv_cvt_pk_f16_f32 %14, %14, %15\n")
KERNEL(k_fmamix, "v_fma_mix_f32 %0, %1, 1.0, %0 op_sel_hi:[1,0,0]\nv_fma_mix_f32 %2, %3, 1.0, %2 op_sel_hi:[1,0,0]\nv_fma_mix_f32 %4, %5, 1.0, %4 op_sel_hi:[1,0,0]\nv_fma_mix_f32 %6, %7, 1.0, %6 op_sel_hi:[1,0,0]\n"
                 "v_fma_mix_f32 %8, %9, 1.0, %8 op_sel_hi:[1,0,0]\nv_fma_mix_f32 %10, %11, 1.0, %10 op_sel_hi:[1,0,0]\nv_fma_mix_f32 %12, %13, 1.0, %12 op_sel_hi:[1,0,0]\nv_fma_mix_f32 %14, %15, 1.0, %14 op_sel_hi:[1,0,0]\n")
__global__ __launch_bounds__(512) void k_acc(float* out, int iters) {
    float v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3;
    for (int it = 0; it < iters; ++it)
        asm volatile("v_accvgpr_write_b32 a0, %0\nv_accvgpr_write_b32 a1, %1\nv_accvgpr_write_b32 a2, %2\nv_accvgpr_write_b32 a3, %3\n"
                     "v_accvgpr_read_b32 %0, a0\nv_accvgpr_read_b32 %1, a1\nv_accvgpr_read_b32 %2, a2\nv_accvgpr_read_b32 %3, a3\n"
                     : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : : "a0", "a1", "a2", "a3");
    out[blockIdx.x * blockDim.x + threadIdx.x] = v0 + v1 + v2 + v3;
}

int main() {
    const int grid = 256, iters = 20000;
    float* out;
    HIP_OK(hipMalloc(&out, grid * 512 * sizeof(float)));
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0));
    HIP_OK(hipEventCreate(&e1));
    struct { const char* name; void (*fn)(float*, int); int trans, plain; } ks[] = {
        {"8 x v_fma_f32", k_fma, 0, 8}, {"8 x v_exp_f32", k_exp32, 8, 0}, {"8 x v_rcp_f32", k_rcp32, 8, 0}, {"8 x v_exp_f16", k_exp16, 8, 0},
        {"8 x v_rcp_f16", k_rcp16, 8, 0}, {"8 x (v_exp_f32, v_fma_f32)", k_exp32_fma, 8, 8}, {"8 x (v_exp_f32, 3 v_fma_f32)", k_exp32_fma3, 8, 24},
        {"8 x (v_rcp_f32, 3 v_fma_f32)", k_rcp32_fma3, 8, 24}, {"8 x (v_exp_f16, 3 v_fma_f32)", k_exp16_fma3, 8, 24},
        {"8 x v_pk_fma_f32", k_pkfma, 0, 8}, {"8 x v_pk_mul_f32", k_pkmul, 0, 8}, {"8 x v_cvt_f32_f16", k_cvt, 0, 8}, {"8 x v_mov_b32", k_mov, 0, 8},
        {"8 x v_permlane16_swap (+2 s_nop 1)", k_swap, 0, 8}, {"8 x v_cvt_pk_f16_f32", k_cvtpk, 0, 8}, {"8 x v_fma_mix_f32", k_fmamix, 0, 8},
        {"4 accvgpr_write + 4 accvgpr_read", k_acc, 0, 8}};
    for (int threads : {256, 512}) {
        printf("---- %d threads per workgroup = %d wave(s) per SIMD, one workgroup per CU, %d iterations\n", threads, threads / 256, iters);
        for (auto& k : ks) {
            hipLaunchKernelGGL(k.fn, dim3(grid), dim3(threads), 0, 0, out, 200);
            HIP_OK(hipDeviceSynchronize());
            HIP_OK(hipEventRecord(e0));
            hipLaunchKernelGGL(k.fn, dim3(grid), dim3(threads), 0, 0, out, iters);
            HIP_OK(hipEventRecord(e1));
            HIP_OK(hipEventSynchronize(e1));
            float ms;
            HIP_OK(hipEventElapsedTime(&ms, e0, e1));
            const double per_iter_ns = ms * 1e6 / iters;
            printf("%-32s %8.3f ms  %7.2f ns per iteration per wave-slot -> at 2.0 GHz %6.1f cycles per iteration (%d trans + %d plain per wave)\n",
                   k.name, ms, per_iter_ns, per_iter_ns * 2.0, k.trans, k.plain);
        }
    }
    return 0;
}
