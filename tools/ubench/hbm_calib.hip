// Calibration kernels for the TCC / FETCH_SIZE / WRITE_SIZE counters: stream a known number of bytes through HBM with
// the access width the forward pipe uses (16 bytes per lane), far beyond the 256 MiB Infinity Cache.
//   calib_read_kernel : reads  1 GiB (fresh each launch: 2 buffers alternate), writes 4 bytes per thread at the end
//   calib_write_kernel: writes 1 GiB
// rocprofv3 --pmc <set> --kernel-trace ... -- ./hbm_calib.so ; bytes per count = 2^30 / counter value per dispatch.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__global__ __launch_bounds__(256) void calib_read_kernel(const u32x4* __restrict__ src, unsigned* __restrict__ sink, size_t n16) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned acc = 0;
    for (; i < n16; i += stride) {
        const u32x4 v = src[i];
        acc += v[0] ^ v[1] ^ v[2] ^ v[3];
    }
    if (acc == 0x12345678u) sink[0] = acc;  // keeps the loads alive, practically never writes
}
__global__ __launch_bounds__(256) void calib_write_kernel(u32x4* __restrict__ dst, size_t n16, unsigned seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n16; i += stride) dst[i] = u32x4{seed, (unsigned)i, seed ^ 1u, 7u};
}

// The board convolution's store pattern: a wave-instruction writes 16 B per lane, four lanes cover 64 contiguous bytes of
// one 512-byte NHWC pixel row (256 channels of fp16), the 16 lane groups go to 16 consecutive pixel rows.  Covers the
// buffer exactly once: block = 16 rows, its 8 waves write the 8 64-byte pieces of each row.
__global__ __launch_bounds__(512) void calib_write64_kernel(unsigned char* __restrict__ dst, size_t rows, unsigned seed) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (size_t r0 = (size_t)blockIdx.x * 16; r0 < rows; r0 += (size_t)gridDim.x * 16) {
        const size_t row = r0 + (lane >> 2);
        *(u32x4*)(dst + row * 512 + wave * 64 + (lane & 3) * 16) = u32x4{seed, (unsigned)row, seed ^ 1u, 7u};
    }
}

int main() {
    const size_t bytes = (size_t)1 << 30, n16 = bytes / 16;
    u32x4 *a, *b;
    unsigned* sink;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&sink, 256);
    hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
    hipDeviceSynchronize();
    for (int r = 0; r < 3; ++r) {
        calib_read_kernel<<<4096, 256>>>(r & 1 ? a : b, sink, n16);
        calib_write_kernel<<<4096, 256>>>(r & 1 ? b : a, n16, r);
        calib_write64_kernel<<<4096, 512>>>((unsigned char*)(r & 1 ? a : b), bytes / 512, r);
    }
    hipDeviceSynchronize();
    printf("calibration: 3 x (read 1 GiB, write 1 GiB, write 1 GiB in 64-byte pieces) done\n");
    return 0;
}
