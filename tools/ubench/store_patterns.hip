// Micro-benchmark: how fast can one CU (8 waves) write an NHWC fp16 tile of 384 pixels x 256 channels (512-byte rows)
// with 16-byte stores, by store pattern?  One workgroup per CU, every CU at once (as the convolution epilogue does).
//   pattern 0: per instruction 16 pixels x 64 contiguous bytes  (4 lanes per pixel)      -- conv_board.h today
//   pattern 1: per instruction  8 pixels x 128 contiguous bytes (8 lanes per pixel, full cache lines)
//   pattern 2: per instruction  2 pixels x 512 contiguous bytes (32 lanes per pixel, whole rows)
//   pattern 3: per instruction 64 pixels x 16 bytes (one lane per pixel)                 -- the worst case
// hipcc --offload-arch=gfx950 -O3 tools/ubench/store_patterns.hip -o /tmp/store_patterns && /tmp/store_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int PAT> __global__ __launch_bounds__(512) void store_kernel(unsigned char* out, unsigned long long* cyc, int reps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned char* tile = out + (size_t)blockIdx.x * 384 * 512;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    u32x4 v = {1u, 2u, 3u, (unsigned)lane};
    for (int r = 0; r < reps; ++r) {
        // each wave writes 48 KiB... of its own: wave = (m = wave & 3 -> 128-byte channel slice, n = wave >> 2 -> 192 pixels)
        const int m = wave & 3, n = wave >> 2;
#pragma unroll
        for (int k = 0; k < 24; ++k) {   // 24 instructions x 1 KiB = 192 pixels x 128 bytes
            size_t off;
            if (PAT == 0) {       // 2 instructions per 16 pixels: 64 bytes per pixel each
                const int j = k >> 1, half = k & 1, px = lane & 15, R = lane >> 4;
                off = (size_t)(n * 192 + j * 16 + px) * 512 + m * 128 + half * 64 + R * 16;
            } else if (PAT == 1) { // 8 pixels x 128 bytes
                const int px = k * 8 + (lane >> 3), c = lane & 7;
                off = (size_t)(n * 192 + px) * 512 + m * 128 + c * 16;
            } else if (PAT == 2) { // whole rows: waves split pixels instead of channels: 2 pixels x 512 bytes
                const int px = (wave * 24 + k) * 2 + (lane >> 5), c = lane & 31;
                off = (size_t)px * 512 + c * 16;
            } else {               // one lane per pixel
                const int px = (k % 3) * 64 + lane, c = k / 3;
                off = (size_t)(n * 192 + px) * 512 + m * 128 + c * 16;
            }
            *(u32x4*)(tile + off) = v;
        }
        v[0] += 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int PAT> void run(unsigned char* d, unsigned long long* dc, int wgs, const char* what) {
    const int reps = 4;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    store_kernel<PAT><<<wgs, 512>>>(d, dc, reps);
    hipDeviceSynchronize();
    hipEventRecord(a);
    store_kernel<PAT><<<wgs, 512>>>(d, dc, reps);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> h(wgs * 8);
    hipMemcpy(h.data(), dc, h.size() * 8, hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto c : h) mean += (double)c;
    mean /= h.size();
    const double bytes_per_wg = (double)reps * 8 * 24 * 1024;
    printf("pattern %d (%s), %d workgroups: %.0f cycles per %d passes -> %.1f B/cycle/CU, kernel %.1f us -> %.2f TB/s\n", PAT, what, wgs, mean,
           reps, bytes_per_wg / mean, ms * 1e3, bytes_per_wg * wgs / (ms * 1e-3) / 1e12);
}

int main() {
    unsigned char* d;
    unsigned long long* dc;
    hipMalloc(&d, (size_t)256 * 384 * 512);
    hipMalloc(&dc, 256 * 8 * 8);
    for (int wgs : {256, 32}) {
        run<0>(d, dc, wgs, "16 px x 64 B");
        run<1>(d, dc, wgs, "8 px x 128 B");
        run<2>(d, dc, wgs, "2 px x 512 B");
        run<3>(d, dc, wgs, "64 px x 16 B");
    }
    return 0;
}
